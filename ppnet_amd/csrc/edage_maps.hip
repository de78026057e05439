// edage_maps.hip — stage B of EDaGe-PP on gfx950: one 256-thread workgroup per map instance.
//
// Replaces, per map (paths relative to the reference's EDaGe-PP/):
//   the placement rejection loop + label transforms        MapGenerate.py:57-93
//   Path.boundary_check                                    Path.py:100-111
//   generate_map_randomly (clearance filter)               MapGenerate.py:126-143
//   plot_obstacles  (explicit disc rule)        Path.py:36-49
//   corridor rotate + translate + compose                  MapGenerate.py:102-111
//   add_init_end_single                                    process_map.py:119-145
//
// HBM traffic per map (R=256, K=20): 64 KiB occupancy grid written once with 16-byte stores,
// 16 KiB of label points written once, 8 KiB corridor bit mask + 16 KiB path points read (shared
// by all placements of a target path, so L2-resident after the first).  The corridor mask and
// the 500 odd path points live in LDS; the clearance filter is one wave per obstacle with a
// shuffle min-reduce; the hull-in-bounds test is one lane per hull vertex with a ballot.
#include "ppn_device.h"
#include "ppn_kernels.h"

namespace ppn {

namespace {
constexpr int NT = 256;
constexpr int NW = NT / 64;
constexpr int MAX_OBS = 256 + PPN_MAX_POCKET;     // K <= 256
constexpr double PI = 3.141592653589793;
}

__global__ __launch_bounds__(NT) void edage_maps_kernel(MapsParams prm) {
    extern __shared__ uint32_t space[];           // R*R/32 words: the target path's corridor mask
    __shared__ double podd[PPN_PATH_POINTS / 2][2];
    __shared__ double hull[PPN_MAX_HULL][2];
    __shared__ double obs[MAX_OBS][4];            // cx(col), cy(row), r, r*r
    __shared__ double cand[256][3];               // K candidates: row, col, r
    __shared__ uint8_t acc[256];
    __shared__ double bc[8];
    __shared__ int bci[8];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // XCD-aware block -> map assignment: blocks b and b+8 share an XCD (round-robin dispatch), so
    // give each XCD one contiguous range of maps; placements of one target path then hit the same
    // L2 for that path's points + corridor mask.  Bijective for any n_maps (speed only).
    int m;
    {
        const int n = prm.n_maps, b = blockIdx.x;
        const int q = n / 8, r = n % 8, x = b % 8;
        m = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
    }
    const int R = prm.R, K = prm.K;
    const double Rd = (double)R, half = Rd / 2.0;
    const int pj = m / prm.placements;            // target path index
    const uint64_t mid = prm.first_map_id + (uint64_t)m;
    const ppn_paths_t& P = prm.paths;
    const ppn_maps_t& O = prm.out;
    const int words = R * R / 32;
    uint32_t flags = 0;

    for (int w = tid; w < words; w += NT) space[w] = P.space_bits[(size_t)pj * words + w];
    const int hn = P.hull_n[pj];
    if (tid < PPN_MAX_HULL) {
        hull[tid][0] = P.hull[((size_t)pj * PPN_MAX_HULL + tid) * 2];
        hull[tid][1] = P.hull[((size_t)pj * PPN_MAX_HULL + tid) * 2 + 1];
    }
    __syncthreads();

    // ------------------------------------------------------------------ placement (wave 0)
    if (wv == 0) {
        int attempts = 0, t0 = 0, t1 = 0;
        double angle = 0.0;
        const double* fed = prm.place_draws ? prm.place_draws + (size_t)m * 3 : nullptr;
        while (true) {
            double u0, u1, u2;
            if (fed) { u0 = fed[0]; u1 = fed[1]; u2 = fed[2]; }
            else {
                const uint32_t d = 3u * (uint32_t)attempts;
                u0 = philox_double(prm.seed, STREAM_PLACE, mid, d);
                u1 = philox_double(prm.seed, STREAM_PLACE, mid, d + 1);
                u2 = philox_double(prm.seed, STREAM_PLACE, mid, d + 2);
            }
            ++attempts;
            angle = u0 * 360.0 - 180.0;                                   // MapGenerate.py:63
            t0 = (int)(u1 * Rd - half);                                   // MapGenerate.py:64 (trunc)
            t1 = (int)(u2 * Rd - half);
            // boundary_check(-angle, [t1, t0]), Path.py:100-111, MapOffset = R/2
            const double rad = (-angle) / 180.0 * PI;
            const double c = cos(rad), s = sin(rad);
            bool out = false;
            if (lane < hn) {
                double hx, hy;
                rot2(c, s, hull[lane][0] - half, hull[lane][1] - half, hx, hy);
                hx = hx + (double)t1 + half;
                hy = hy + (double)t0 + half;
                out = (hx < 0.0) || (hx >= Rd) || (hy < 0.0) || (hy >= Rd);
            }
            const bool ok = __ballot(out) == 0ull;
            if (ok) break;
            if (fed || attempts >= PPN_PLACE_TRY_CAP) { flags |= PPN_FLAG_PLACE_CAP; break; }
        }
        if (lane == 0) {
            bc[0] = angle; bci[0] = t0; bci[1] = t1; bci[2] = attempts; bci[3] = (int)flags;
            O.angle[m] = angle;
            O.translation[(size_t)m * 2] = t0; O.translation[(size_t)m * 2 + 1] = t1;
            O.attempts[m] = attempts;
        }
    }
    __syncthreads();
    const double angle = bc[0];
    const int t0 = bci[0], t1 = bci[1];
    flags = (uint32_t)bci[3];
    const double rad = (-angle) / 180.0 * PI;                             // MapGenerate.py:72
    const double c = cos(rad), s = sin(rad);
    const double tr0 = (double)t1, tr1 = (double)t0;                      // [translation[1], translation[0]]

    // ------------------------------------------------------------------ labels
    for (int q = tid; q < PPN_PATH_POINTS; q += NT) {                     // MapGenerate.py:76-80
        const double x = P.pathpoint_image[((size_t)pj * PPN_PATH_POINTS + q) * 2] - half;
        const double y = P.pathpoint_image[((size_t)pj * PPN_PATH_POINTS + q) * 2 + 1] - half;
        double rx, ry;
        rot2(c, s, x, y, rx, ry);
        rx = rx + half + tr0;
        ry = ry + half + tr1;
        if (q & 1) { podd[q >> 1][0] = rx; podd[q >> 1][1] = ry; }
        if (O.pathpoint) {
            O.pathpoint[((size_t)m * PPN_PATH_POINTS + q) * 2] = rx;
            O.pathpoint[((size_t)m * PPN_PATH_POINTS + q) * 2 + 1] = ry;
        }
    }
    if (tid < PPN_SEGS + 1) {                                             // MapGenerate.py:70-74
        const double x = P.segpoint_image[((size_t)pj * 11 + tid) * 2] - half;
        const double y = P.segpoint_image[((size_t)pj * 11 + tid) * 2 + 1] - half;
        double rx, ry;
        rot2(c, s, x, y, rx, ry);
        rx = rx + half + tr0;
        ry = ry + half + tr1;
        O.segpoint[((size_t)m * 11 + tid) * 2] = rx;
        O.segpoint[((size_t)m * 11 + tid) * 2 + 1] = ry;
        if (tid == 0) { bc[2] = rx; bc[3] = ry; }
        if (tid == PPN_SEGS) { bc[4] = rx; bc[5] = ry; }                  // end = segpoint[10]
    }
    // K random obstacle candidates (MapGenerate.py:128-136)
    if (tid < K) {
        double ux, uy, us;
        if (prm.obst_draws) {
            const double* d = prm.obst_draws + (size_t)m * 3 * K;
            ux = d[tid]; uy = d[K + tid]; us = d[2 * K + tid];
        } else {
            ux = philox_double(prm.seed, STREAM_OBST, mid, (uint32_t)tid);
            uy = philox_double(prm.seed, STREAM_OBST, mid, (uint32_t)(K + tid));
            us = philox_double(prm.seed, STREAM_OBST, mid, (uint32_t)(2 * K + tid));
        }
        cand[tid][0] = ux * prm.map_size / prm.map_size * Rd;
        cand[tid][1] = uy * prm.map_size / prm.map_size * Rd;
        cand[tid][2] = us * prm.obstacles_size / prm.map_size * Rd;
    }
    __syncthreads();

    // ------------------------------------------------------------------ clearance filter: one wave per obstacle
    {
        const double c_px = prm.clearance / prm.map_size * Rd;            // MapGenerate.py:142
        for (int k = wv; k < K; k += NW) {
            const double ox = cand[k][0], oy = cand[k][1];
            double mn = 1e300;
            for (int q = lane; q < PPN_PATH_POINTS / 2; q += 64)
                mn = fmin(mn, dist2d(podd[q][0], podd[q][1], ox, oy));
            mn = wave_min(mn);
            if (lane == 0) acc[k] = (mn > cand[k][2] + c_px) ? 1 : 0;
        }
    }
    __syncthreads();
    if (tid == 0) {                                                       // ordered compaction
        int n = 0;
        for (int k = 0; k < K; ++k) {
            if (acc[k]) {
                obs[n][0] = cand[k][1]; obs[n][1] = cand[k][0]; obs[n][2] = cand[k][2];   // [col,row,r]
                ++n;
            }
        }
        bci[4] = n;
    }
    __syncthreads();
    const int n_rand = bci[4];
    const int n_pocket = P.n_obstacles[pj];
    if (tid < n_pocket) {                                                 // MapGenerate.py:83-89
        const double* o = P.obstacles + ((size_t)pj * PPN_MAX_POCKET + tid) * 3;
        double rx, ry;
        rot2(c, s, o[1] - half, o[0] - half, rx, ry);
        rx = rx + half + tr0;
        ry = ry + half + tr1;
        obs[n_rand + tid][0] = ry; obs[n_rand + tid][1] = rx; obs[n_rand + tid][2] = o[2];
    }
    __syncthreads();
    const int n_obs = n_rand + n_pocket;
    for (int n = tid; n < n_obs; n += NT) {
        obs[n][3] = obs[n][2] * obs[n][2];
        double* o = O.obstacles + ((size_t)m * (K + PPN_MAX_POCKET) + n) * 3;
        o[0] = obs[n][0]; o[1] = obs[n][1]; o[2] = obs[n][2];
    }
    if (O.accept && tid < K) O.accept[(size_t)m * K + tid] = acc[tid];
    if (tid == 0) {
        O.n_obstacles[(size_t)m * 2] = n_obs;
        O.n_obstacles[(size_t)m * 2 + 1] = n_rand;
        O.flags[m] = flags | P.flags[pj];
    }
    __syncthreads();

    // ------------------------------------------------------------------ raster: 16 pixels (one 16-byte store) per step
    {
        const double b = (-angle) * PI / 180.0;                           // rotate_nearest(space, -angle)
        const double c3 = cos(b), s3 = sin(b);
        const int r_init = (int)rint(bc[2]), c_init = (int)rint(bc[3]);   // process_map.py:127-135
        const int r_end = (int)rint(bc[4]), c_end = (int)rint(bc[5]);
        const int cpr = R / 16;
        uint8_t* g = O.grid + (size_t)m * R * R;
        for (int ch = tid; ch < R * cpr; ch += NT) {
            const int i = ch / cpr, j0 = (ch - i * cpr) * 16;
            const double yc = (double)i + 0.5;
            uint32_t occ = 0u;
            for (int n = 0; n < n_obs; ++n) {
                const double dy = yc - obs[n][1];
                const double rr = obs[n][3];
                const double dy2 = dy * dy;
                if (dy2 > rr) continue;                                   // row misses the disc
                const double r = obs[n][2], cx = obs[n][0];
                if ((double)j0 + 16.0 < cx - r || (double)j0 > cx + r) continue;   // conservative column cull
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const double dx = ((double)(j0 + k) + 0.5) - cx;
                    if (dx * dx + dy2 <= rr) occ |= 1u << k;
                }
            }
            if (occ) {                                                    // corridor wins over obstacles
                const int i1 = i - t1;                                    // translate: ty = translation[1]
                if (i1 >= 0 && i1 < R) {
                    const double yo = ((double)i1 + 0.5) - half;
                    for (int k = 0; k < 16; ++k) {
                        if (!((occ >> k) & 1u)) continue;
                        const int j1 = j0 + k - t0;
                        if (j1 < 0 || j1 >= R) continue;
                        const double xo = ((double)j1 + 0.5) - half;
                        const double xs = c3 * xo - s3 * yo, ys = s3 * xo + c3 * yo;
                        const int jj = (int)rint(xs + (half - 0.5)), ii = (int)rint(ys + (half - 0.5));
                        if (ii < 0 || ii >= R || jj < 0 || jj >= R) continue;
                        const int bit = ii * R + jj;
                        if ((space[bit >> 5] >> (bit & 31)) & 1u) occ &= ~(1u << k);
                    }
                }
            }
            uint32_t mark = 0u;
            const bool ri = (i >= r_init - 3) && (i <= r_init + 3), re = (i >= r_end - 3) && (i <= r_end + 3);
            if (ri || re) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int j = j0 + k;
                    if ((ri && j >= c_init - 3 && j <= c_init + 3) || (re && j >= c_end - 3 && j <= c_end + 3))
                        mark |= 1u << k;
                }
            }
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t v = 0u;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int bit = q * 4 + k;
                    const uint32_t px = ((mark >> bit) & 1u) ? PPN_GRID_MARK : (((occ >> bit) & 1u) ? PPN_GRID_OBST : PPN_GRID_FREE);
                    v |= px << (8 * k);
                }
                w[q] = v;
            }
            *reinterpret_cast<uint4*>(g + (size_t)i * R + j0) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// Path.boundary_check (Path.py:100-111): one wave per (angle, translation) pair
__global__ __launch_bounds__(NT) void boundary_check_kernel(const double* hull, int hull_n, const double* angle_deg,
                                                            const double* trans_rc, int n, int R, uint8_t* ok) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * NW + (threadIdx.x >> 6);
    if (i >= n) return;
    const double half = (double)R / 2.0, Rd = (double)R;
    const double rad = angle_deg[i] / 180.0 * PI;
    const double c = cos(rad), s = sin(rad);
    bool out = false;
    for (int v = lane; v < hull_n; v += 64) {
        double hx, hy;
        rot2(c, s, hull[v * 2] - half, hull[v * 2 + 1] - half, hx, hy);
        hx = hx + trans_rc[i * 2] + half;
        hy = hy + trans_rc[i * 2 + 1] + half;
        out = out || (hx < 0.0) || (hx >= Rd) || (hy < 0.0) || (hy >= Rd);
    }
    const bool good = __ballot(out) == 0ull;
    if (lane == 0) ok[i] = good ? 1 : 0;
}

// explicit obstacle raster rule (stands in for Path.plot_obstacles, Path.py:36-49)
__global__ __launch_bounds__(NT) void disc_raster_kernel(const double* obstacles, const int32_t* counts, int stride,
                                                         int n_maps, int R, uint8_t* grid) {
    __shared__ double obs[MAX_OBS][4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int n_obs = min(counts[m], MAX_OBS);
    for (int n = tid; n < n_obs; n += NT) {
        const double* o = obstacles + ((size_t)m * stride + n) * 3;
        obs[n][0] = o[0]; obs[n][1] = o[1]; obs[n][2] = o[2]; obs[n][3] = o[2] * o[2];
    }
    __syncthreads();
    const int cpr = R / 16;
    uint8_t* g = grid + (size_t)m * R * R;
    for (int ch = tid; ch < R * cpr; ch += NT) {
        const int i = ch / cpr, j0 = (ch - i * cpr) * 16;
        const double yc = (double)i + 0.5;
        uint32_t occ = 0u;
        for (int n = 0; n < n_obs; ++n) {
            const double dy = yc - obs[n][1], rr = obs[n][3], dy2 = dy * dy;
            if (dy2 > rr) continue;
            const double cx = obs[n][0];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const double dx = ((double)(j0 + k) + 0.5) - cx;
                if (dx * dx + dy2 <= rr) occ |= 1u << k;
            }
        }
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t v = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v |= (((occ >> (q * 4 + k)) & 1u) ? (uint32_t)PPN_GRID_OBST : (uint32_t)PPN_GRID_FREE) << (8 * k);
            w[q] = v;
        }
        *reinterpret_cast<uint4*>(g + (size_t)i * R + j0) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

}  // namespace ppn
