// plan_tail.hip — planner tail of the PPNet inference path on gfx950 (reference: EDaGe-PP/process_map.py).
//
//   collision_segments_kernel : collision_check_circle_edge (process_map.py:383-425), one thread per
//                               segment, float32 like the reference's torch tensors; the any-hit
//                               over a problem's waypoint segments is a host-side reduce of `hit`.
//   extract_paths_kernel      : extract_path's greedy 8-neighbour walk (process_map.py:293-365), one
//                               wave64 per problem: candidates are scored by lanes 0..7, the revisit
//                               test (<= 1.5 px to an earlier point) runs one history point per lane
//                               with a ballot; the 1 s wall-clock timeout becomes the max_wp cap.
#include "ppn_device.h"
#include "ppn_kernels.h"

namespace ppn {

// collision_check_circle_edge for one segment against `count` obstacle rows (ox, oy, size) of type T (float32 arithmetic on
// float32 values: a float64 row is rounded first, as the reference's torch.tensor(..., dtype=float32) does).
template <typename T>
__device__ __forceinline__ uint8_t segment_hits(float s0, float s1, float e0, float e1, const T* obs, int count, float clearance, float bound) {
    // process_map.py:384-387 hard-codes the reference's 224-pixel maps; `bound` carries the map's resolution
    if (s0 < 0.0f || s1 > bound || e0 < 0.0f || e1 > bound) return 1;
    const float sx = s1, sy = s0, ex = e1, ey = e0;                       // swap to (x, y), :388-389
    float dx = ex - sx, dy = ey - sy;
    const float nrm = sqrtf(dx * dx + dy * dy);
    float dirx = dy / nrm, diry = -dx / nrm;                              // :390-391
    const double lim_add = (double)clearance / 2.0;
    for (int k = 0; k < count; ++k) {
        const float ox = (float)obs[k * 3], oy = (float)obs[k * 3 + 1];
        const double size = (double)(float)obs[k * 3 + 2];
        const double lim = size + lim_add;
        // scipy euclidean keeps the float32 of its inputs
        const float ddx = ex - ox, ddy = ey - oy;
        if ((double)sqrtf(ddx * ddx + ddy * ddy) < lim) return 1;         // :397 (tests e twice, never s)
        const float qx = ox - sx, qy = oy - sy;
        float dis = dirx * qx + diry * qy;                                // np.dot, float32
        if (dis > 0.0f) { dirx = -dirx; diry = -diry; }                   // dir mutates across obstacles, :406-407
        dis = fabsf(dis);
        const float px = ox + dis * dirx, py = oy + dis * diry;
        float ax = px - sx, ay = py - sy;
        const float an = sqrtf(ax * ax + ay * ay);
        ax = ax / an; ay = ay / an;
        float bx = px - ex, by = py - ey;
        const float bn = sqrtf(bx * bx + by * by);
        bx = bx / bn; by = by / bn;
        if ((double)dis < lim && (ax * bx + ay * by) < 0.0f) return 1;    // :414
    }
    return 0;
}

__global__ __launch_bounds__(256) void collision_segments_kernel(const float* s_in, const float* e_in,
                                                                 const int32_t* prob, int n_seg, const float* obs,
                                                                 const int32_t* obs_off, float clearance,
                                                                 float bound, uint8_t* hit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seg) return;
    const int p = prob[i];
    hit[i] = segment_hits<float>(s_in[i * 2], s_in[i * 2 + 1], e_in[i * 2], e_in[i * 2 + 1], obs + (size_t)obs_off[p] * 3, obs_off[p + 1] - obs_off[p],
                                 clearance, bound);
}

// The planner tail's glue as two kernels instead of ~30 framework launches per batch (process_map.py:355-359, 491-495):
//   assemble_paths_kernel   [init] + waypoints * rate + [end] into a fixed [n][max_wp + 2][2] polyline, counts = ok ? n_wp + 2 : 0
//   plan_collision_kernel   one thread per consecutive-waypoint segment of every plan; collision[b] = any segment hits any of
//                           the problem's first n_obs[b] obstacle rows (collision must be zeroed by the caller's launch)
__global__ __launch_bounds__(256) void assemble_paths_kernel(const double* __restrict__ wp, const int32_t* __restrict__ wp_n, const uint8_t* __restrict__ ok,
                                                             const double* __restrict__ init, const double* __restrict__ end, double rate, int n, int max_wp,
                                                             double* __restrict__ full, int32_t* __restrict__ counts) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int M = max_wp + 2;
    if (idx >= (long long)n * M) return;
    const int b = (int)(idx / M), s = (int)(idx - (long long)b * M);
    const int e_slot = min(wp_n[b] + 1, max_wp + 1);
    double v0 = 0.0, v1 = 0.0;
    if (s >= 1 && s <= max_wp) { v0 = wp[((size_t)b * max_wp + s - 1) * 2] * rate; v1 = wp[((size_t)b * max_wp + s - 1) * 2 + 1] * rate; }
    if (s == 0) { v0 = init[b * 2]; v1 = init[b * 2 + 1]; }
    if (s == e_slot) { v0 = end[b * 2]; v1 = end[b * 2 + 1]; }
    full[idx * 2] = v0; full[idx * 2 + 1] = v1;
    if (s == 0) counts[b] = ok[b] ? wp_n[b] + 2 : 0;
}

template <typename T>
__global__ __launch_bounds__(256) void plan_collision_kernel(const double* __restrict__ full, const int32_t* __restrict__ counts, const T* __restrict__ obstacles,
                                                             const int32_t* __restrict__ n_obs, int B, int M, int S, float clearance, float bound,
                                                             uint8_t* __restrict__ collision) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * (M - 1)) return;
    const int b = (int)(idx / (M - 1)), i = (int)(idx - (long long)b * (M - 1));
    if (i >= counts[b] - 1) return;
    const double* p = full + ((size_t)b * M + i) * 2;
    const int cnt = min(max(n_obs[b], 0), S);
    if (segment_hits<T>((float)p[0], (float)p[1], (float)p[2], (float)p[3], obstacles + (size_t)b * S * 3, cnt, clearance, bound)) collision[b] = 1;
}

int assemble_paths_launch(const double* wp, const int32_t* wp_n, const uint8_t* ok, const double* init, const double* end, double rate, int n, int max_wp,
                          double* full, int32_t* counts, hipStream_t stream) {
    const long long total = (long long)n * (max_wp + 2);
    hipLaunchKernelGGL(assemble_paths_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, wp, wp_n, ok, init, end, rate, n, max_wp, full, counts);
    return (int)hipGetLastError();
}

int plan_collision_launch(const double* full, const int32_t* counts, const void* obstacles, int obs_f64, const int32_t* n_obs, int B, int M, int S,
                          float clearance, float bound, uint8_t* collision, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(collision, 0, (size_t)B, stream);
    if (e != hipSuccess) return (int)e;
    const long long total = (long long)B * (M - 1);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (obs_f64) hipLaunchKernelGGL((plan_collision_kernel<double>), grid, dim3(256), 0, stream, full, counts, (const double*)obstacles, n_obs, B, M, S, clearance, bound, collision);
    else hipLaunchKernelGGL((plan_collision_kernel<float>), grid, dim3(256), 0, stream, full, counts, (const float*)obstacles, n_obs, B, M, S, clearance, bound, collision);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(64) void extract_paths_kernel(const float* heat, int n, int H, int W, const double* init,
                                                           const double* end, int max_wp, double* wp, int32_t* wp_n,
                                                           uint8_t* ok, int vis_dim, int stage_heat) {
    __shared__ short hist[PPN_MAX_WAYPOINTS][2];                         // offsets from init, exact small ints
    // visited[(r + off) * vis_dim + (c + off)]: one bit per lattice offset, set for the points OLDER than the last two — the
    // revisit rule (process_map.py:327: equal to any earlier point, or within 1.5 px of one that is not among the last two)
    // is then nine bit tests around the candidate plus two comparisons, instead of a scan of the whole history per step
    // (the walk is up to max_wp steps long: the scan made the kernel O(n^2)).  vis_dim = 0: no bitmap (very large maps).
    // stage_heat: the heat map (values k/255 of an 8-bit image) is copied into LDS as its 8-bit codes behind the bitmap — the
    // walk reads 8 candidates per step, one dependent L2 round trip per step otherwise.  Codes order like the values.
    extern __shared__ uint32_t visited[];
    const int p = blockIdx.x, lane = threadIdx.x;
    const float* hm = heat + (size_t)p * H * W;
    uint8_t* hcode = reinterpret_cast<uint8_t*>(visited + (vis_dim * vis_dim + 31) / 32);
    if (stage_heat) {
        bool bad = false;                                                 // a value that is not k/255, k in 0..255: keep the float reads
        for (int q = lane; q < H * W / 4; q += 64) {
            const float4 f = reinterpret_cast<const float4*>(hm)[q];
            const float c0 = rintf(f.x * 255.0f), c1 = rintf(f.y * 255.0f), c2 = rintf(f.z * 255.0f), c3 = rintf(f.w * 255.0f);
            bad = bad || !(c0 / 255.0f == f.x && c1 / 255.0f == f.y && c2 / 255.0f == f.z && c3 / 255.0f == f.w) ||
                  !(c0 >= 0.0f && c0 <= 255.0f && c1 >= 0.0f && c1 <= 255.0f && c2 >= 0.0f && c2 <= 255.0f && c3 >= 0.0f && c3 <= 255.0f);
            reinterpret_cast<uint32_t*>(hcode)[q] = (uint32_t)c0 | ((uint32_t)c1 << 8) | ((uint32_t)c2 << 16) | ((uint32_t)c3 << 24);
        }
        if (__ballot(bad) != 0ull) stage_heat = 0;                        // one wave per problem: uniform
    }
    double* out = wp + (size_t)p * max_wp * 2;
    const double i0 = init[p * 2], i1 = init[p * 2 + 1];
    const double g0 = end[p * 2], g1 = end[p * 2 + 1];
    const int mr[8] = {0, 0, 1, -1, 1, 1, -1, -1};                        // motions, process_map.py:294-297
    const int mc[8] = {1, -1, 0, 0, 1, -1, 1, -1};
    const int voff = vis_dim / 2;
    for (int w = lane; w < (vis_dim * vis_dim + 31) / 32; w += 64) visited[w] = 0u;
    __syncthreads();
    // a waypoint is init + integer offset: keep the offsets exact in int, rebuild doubles on demand
    int cr = 0, cc = 0, cnt = 0, success = 0;
    while (cnt < max_wp) {
        // lanes 0..7 score the candidates
        float v = 0.0f;
        int nr = 0, nc = 0;
        if (lane < 8) {
            nr = cr + mr[lane]; nc = cc + mc[lane];
            const int ri = (int)rint(i0 + (double)nr), ci = (int)rint(i1 + (double)nc);
            if (ri >= 0 && ri < W && ci >= 0 && ci < H)                      // :318 (size[0] bounds c[0])
                v = stage_heat ? (float)hcode[ri * W + ci] : hm[(size_t)ri * W + ci];
        }
        int chosen = -1;
        while (true) {
            // first maximum among lanes 0..7 with v > 0
            float best = 0.0f; int bi = -1;
            for (int k = 0; k < 8; ++k) {
                const float vk = __shfl(v, k, 64);
                if (vk > best) { best = vk; bi = k; }
            }
            if (bi < 0) break;                                            // max(candidate_v) == 0 -> fail
            const int br = __shfl(nr, bi, 64), bcn = __shfl(nc, bi, 64);
            bool rej = false;
            if (vis_dim > 0) {
                bool r = false;
                if (lane < 9) {                                           // the 3 x 3 cells within lattice distance^2 <= 2
                    const int rr = br + lane / 3 - 1 + voff, cq = bcn + lane % 3 - 1 + voff;
                    if (rr >= 0 && rr < vis_dim && cq >= 0 && cq < vis_dim) {
                        const int bit = rr * vis_dim + cq;
                        r = (visited[bit >> 5] >> (bit & 31)) & 1u;
                    }
                } else if (lane < 11) {                                   // equality with the last two points
                    const int q = cnt - 1 - (lane - 9);
                    if (q >= 0) r = hist[q][0] == br && hist[q][1] == bcn;
                }
                rej = __ballot(r) != 0ull;
            } else {
                // revisit test against history[0 .. cnt-3] by scan (process_map.py:327: equal, or <=1.5 px and i < len-2)
                for (int b = 0; b < cnt; b += 64) {
                    const int q = b + lane;
                    bool r = false;
                    if (q < cnt) {
                        const int hr = hist[q][0], hc = hist[q][1];
                        const int dr = hr - br, dc = hc - bcn;
                        const bool same = (dr == 0 && dc == 0);
                        const bool nearp = (dr * dr + dc * dc) <= 2;            // lattice distance <= 1.5
                        r = same || (nearp && q < cnt - 2);
                    }
                    if (__ballot(r) != 0ull) { rej = true; break; }
                }
            }
            if (!rej) { chosen = bi; break; }
            if (lane == bi) v = 0.0f;                                      // candidate_v[candidate_i] = 0
        }
        if (chosen < 0) break;
        cr = __shfl(nr, chosen, 64); cc = __shfl(nc, chosen, 64);
        if (lane == 0) {
            hist[cnt][0] = (short)cr; hist[cnt][1] = (short)cc;
            if (vis_dim > 0 && cnt >= 2) {                                // the point that now drops out of "the last two"
                const int bit = (hist[cnt - 2][0] + voff) * vis_dim + hist[cnt - 2][1] + voff;
                visited[bit >> 5] |= 1u << (bit & 31);
            }
        }
        ++cnt;
        __syncthreads();                                                  // single wave: orders the LDS stores
        const double d0 = (i0 + (double)cr) - g0, d1 = (i1 + (double)cc) - g1;
        if (sqrt(d0 * d0 + d1 * d1) <= 2.5) { success = 1; break; }       // :346
    }
    // offsets -> coordinates (still in down-sampled units; the host scales by the rate)
    __syncthreads();
    for (int q = lane; q < cnt; q += 64) {
        out[q * 2] = i0 + (double)hist[q][0];
        out[q * 2 + 1] = i1 + (double)hist[q][1];
    }
    if (lane == 0) { wp_n[p] = success ? cnt : 0; ok[p] = (uint8_t)success; }
}

// One pass of Pillow's ImagingResample for 8-bit pixels with the bilinear (triangle) filter.
// horizontal != 0: in [n][inH][inW] -> out [n][inH][outW]; else in [n][inH][inW] -> out [n][outH][inW].
// Coefficients follow Pillow's operation order in double (precompute_coeffs + normalize_coeffs_8bpc, PRECISION_BITS = 22).
// They depend only on the output coordinate along the resampled axis, so a thread computes them ONCE (the two double loops with
// a division per tap were the kernel's whole cost: 0.15 ms per 256 heat maps) and then walks RESIZE_REP rows (horizontal: rows of
// the whole batch are independent lines) or RESIZE_REP images (vertical) with them.  Same integer arithmetic per output: bit-exact.
constexpr int RESIZE_REP = PPN_RESIZE_REP, RESIZE_TAPS = 8;
__global__ __launch_bounds__(256) void resize_pass_kernel(const uint8_t* in, int n, int inH, int inW, int outH, int outW,
                                                          int horizontal, uint8_t* out) {
    const int oH = horizontal ? inH : outH, oW = horizontal ? outW : inW;
    // grid.y = 256-pixel pieces of an output row;  grid.x = groups of RESIZE_REP lines of the batch (horizontal: n * inH lines) or
    // (group of RESIZE_REP images, output row) pairs (vertical) — block-uniform, split on the scalar unit
    const int x = (int)(blockIdx.y * 256u + threadIdx.x);
    if (x >= oW) return;
    const int y = horizontal ? 0 : (int)(blockIdx.x % (uint32_t)oH);
    const int inSize = horizontal ? inW : inH, outSize = horizontal ? outW : outH, xx = horizontal ? x : y;
    const double scale = (double)inSize / (double)outSize;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    const double center = 0.0 + ((double)xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > inSize) xmax = inSize;
    xmax -= xmin;
    double ww = 0.0;
    for (int t = 0; t < xmax; ++t) {
        double a = ((double)(t + xmin) - center + 0.5) * ss;
        if (a < 0.0) a = -a;
        ww += a < 1.0 ? 1.0 - a : 0.0;
    }
    const bool cached = xmax <= RESIZE_TAPS;                                  // down-sampling by more than 4 falls back to per-output weights
    int kq[RESIZE_TAPS];
#pragma unroll
    for (int t = 0; t < RESIZE_TAPS; ++t) {
        double a = ((double)(t + xmin) - center + 0.5) * ss;
        if (a < 0.0) a = -a;
        double w = a < 1.0 ? 1.0 - a : 0.0;
        if (ww != 0.0) w = w / ww;
        kq[t] = (t < xmax) ? (w < 0.0 ? (int)(-0.5 + w * 4194304.0) : (int)(0.5 + w * 4194304.0)) : 0;
    }
    const long long lines = horizontal ? (long long)n * inH : (long long)n;
    const long long first = (long long)(horizontal ? blockIdx.x : blockIdx.x / (uint32_t)oH) * RESIZE_REP;
    for (int r = 0; r < RESIZE_REP; ++r) {
        const long long line = first + r;
        if (line >= lines) break;
        // horizontal: `line` is a row of the batch (image line / inH, row line % inH — contiguous either way); vertical: an image
        const uint8_t* src = horizontal ? in + (size_t)line * inW + xmin : in + ((size_t)line * inH + xmin) * inW + x;
        const size_t step = horizontal ? 1 : (size_t)inW;
        int acc = 1 << 21;
        if (cached) {
#pragma unroll
            for (int t = 0; t < RESIZE_TAPS; ++t)
                if (t < xmax) acc += (int)src[(size_t)t * step] * kq[t];
        } else {
            for (int t = 0; t < xmax; ++t) {
                double a = ((double)(t + xmin) - center + 0.5) * ss;
                if (a < 0.0) a = -a;
                double w = a < 1.0 ? 1.0 - a : 0.0;
                if (ww != 0.0) w = w / ww;
                const int k = w < 0.0 ? (int)(-0.5 + w * 4194304.0) : (int)(0.5 + w * 4194304.0);
                acc += (int)src[(size_t)t * step] * k;
            }
        }
        int v = acc >> 22;
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        const size_t o = horizontal ? (size_t)line * oW + x : ((size_t)line * oH + y) * oW + x;
        out[o] = (uint8_t)v;
    }
}

__global__ __launch_bounds__(256) void philox_doubles_kernel(uint64_t seed, uint32_t stream_id, uint64_t instance,
                                                             uint32_t first, int count, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = philox_double(seed, stream_id, instance, first + (uint32_t)i);
}

}  // namespace ppn
