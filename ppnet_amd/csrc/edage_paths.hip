// edage_paths.hip — stage A of EDaGe-PP on gfx950: one 256-thread workgroup per target path.
//
// Replaces, per path (paths relative to the reference's EDaGe-PP/):
//   PathSeg.__init__/random/translation/gradient/length     PathSeg.py:10-58
//   Path.generate/transform/point_transform/plot            Path.py:78-98, 224-233, 253-316
//   Path.draw_boundary                                      Path.py:318-356
//   Path.path_space + free_space_bydirection                Path.py:113-142, 397-404
//   Path.convexhull (Qhull -> exact integer gift wrapping)  Path.py:388-395
//   Path.space_normalization (+ explicit nearest resample)  Path.py:157-193
//   Path.search_isle / Path.set_obstacles                   Path.py:502-537, 463-500
//
// Data layout: everything a path needs lives in LDS for the life of the workgroup — the 2R x 2R
// corridor canvas as a bit mask ((2R)^2/8 bytes: 32 KiB at R=256, 128 KiB at R=512), the 1000
// path points (16 KiB), the integer lattice copy for the hull (8 KiB) — and each result is
// written to HBM exactly once.  Reductions are wave64 shuffles + one LDS slot per wave.
// Arithmetic is unfused double (file is built with -ffp-contract=off), matching oracle/edage_np.py.
#include "ppn_device.h"
#include "ppn_kernels.h"

#ifdef PPN_PHASE_TIMING
__device__ unsigned long long g_paths_phase_cycles[16];
#define PPN_PSTAMP(idx) do { __syncthreads(); if (threadIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        atomicAdd(&g_paths_phase_cycles[idx], t_ - tp_prev_); tp_prev_ = t_; } } while (0)
#define PPN_PSTAMP_INIT unsigned long long tp_prev_ = __builtin_amdgcn_s_memtime()
extern "C" int ppn_debug_paths_phase_cycles(unsigned long long* out_host, int reset) {
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_paths_phase_cycles), sizeof(unsigned long long) * 16) != hipSuccess) return -2;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_paths_phase_cycles), z, sizeof(z)) != hipSuccess) return -2;
    }
    return 0;
}
#else
#define PPN_PSTAMP(idx) do {} while (0)
#define PPN_PSTAMP_INIT do {} while (0)
#endif

namespace ppn {

namespace {

// One path per workgroup.  256 threads: the chain is latency-bound (alone 256/512/1024 threads take the same time), and
// beside stage B a 4-wave workgroup leaves its CU one more stage-B workgroup than an 8-wave one does (registers: 97 x 1
// vs 81 x 2 per SIMD): 0.200 vs 0.210 ms per bench step.
#ifndef PPN_PATHS_WAVES_PER_EU
#define PPN_PATHS_WAVES_PER_EU 6      // register cap (<= 80 VGPRs): what a stage-A wave leaves of its SIMD's file is stage B's
#endif
constexpr int NT = PPN_PATHS_THREADS;
constexpr int NW = NT / 64;

struct SegLds {
    double poly[PPN_SEGS][5];
    double pder[PPN_SEGS][4];
    double endpoint[PPN_SEGS];
    double trans[PPN_SEGS][2];
    double ang[PPN_SEGS];
    double cs[PPN_SEGS], sn[PPN_SEGS];
    double t0[PPN_SEGS][2];        // PathSeg.Translation before chaining: (E, p(E))
    double grad[PPN_SEGS][2];      // GradSt, GradEnd
    double segpoint[PPN_SEGS + 1][2];
    int straight[PPN_SEGS];
};

__device__ __forceinline__ double horner3(const double* p, double x) {
    double y = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) y = y * x + p[k];
    return y;
}

// one boundary sample of segment s at abscissa index j/50 (Path.py:320-333): centre-line point
// after point_transform and the unit normal
__device__ __forceinline__ void boundary_sample(const SegLds& S, int s, int j, double& px, double& py,
                                                double& nx, double& ny) {
    const double x = ((double)j / 50.0) * S.endpoint[s];
    const double y = horner4(S.poly[s], x);
    const double yd = horner3(S.pder[s], x);
    rot2(S.cs[s], S.sn[s], x, y, px, py);
    px = px + S.trans[s][0];
    py = py + S.trans[s][1];
    rot2(S.cs[s], S.sn[s], yd, -1.0, nx, ny);
    const double n = sqrt(nx * nx + ny * ny);
    nx = nx / n;
    ny = ny / n;
}

}  // namespace

__global__ __launch_bounds__(NT, PPN_PATHS_WAVES_PER_EU) void edage_paths_kernel(PathsParams prm) {
    // dynamic LDS: pp [1000][2] f64 path points (world, later image) | lat [1000][2] i32 integer lattice (hull input) |
    // canvas (2R)^2 / 32 words.  They are dynamic rather than static so that the compiler's occupancy estimate is not
    // pinned by 24 KB of static LDS and the register cap below is honoured.
    extern __shared__ uint64_t paths_lds[];
    double (*pp)[2] = reinterpret_cast<double (*)[2]>(paths_lds);
    int (*lat)[2] = reinterpret_cast<int (*)[2]>(reinterpret_cast<unsigned char*>(paths_lds) + PPN_PATH_POINTS * 16);
    uint32_t* canvas = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(paths_lds) + PPN_PATH_POINTS * 24);
    __shared__ SegLds S;
    __shared__ double hull[PPN_MAX_HULL][2];
    __shared__ int hull_i[PPN_MAX_HULL][2];
    __shared__ double fit_part[PPN_SEGS][4];
    __shared__ double red_v[NW];
    __shared__ int red_i[NW];
    __shared__ double bc[16];                     // broadcast scalars
    __shared__ double at2[PPN_SEGS][2];           // atan of each segment's start / end gradient

    const int p = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int R = prm.R;
    const uint64_t pid = prm.first_id + (uint64_t)p;
    const ppn_paths_t& O = prm.out;
    const double* fed = prm.draws ? prm.draws + (size_t)p * PPN_DRAWS_PER_PATH : nullptr;
    const double clearance = prm.clearance;
    const double step_len = 1.0 / (double)R * prm.map_size;      // Path.py:118
    const double step_c2i = prm.map_size / (double)R;            // Path.py:380
    const double Rd = (double)R;
    uint32_t flags = 0;
    PPN_PSTAMP_INIT;

    // zero the canvas
    const int canvas_words = (2 * R) * (2 * R) / 32;
    for (int w = tid; w < canvas_words; w += NT) canvas[w] = 0u;

    // ------------------------------------------------------------------ A1: ten segment fits
    const double d0 = fed ? fed[0] : philox_double(prm.seed, STREAM_PATH, pid, 0);
    const int forced = prm.force_straight ? (int)prm.force_straight[p] : -1;
    const bool path_straight = forced >= 0 ? (forced != 0) : !(d0 > 0.01);    // PathGenerate.py:36 / Path.py:53
    for (int s = wv; s < PPN_SEGS; s += NW) {                    // one wave fits one segment: no cross-wave reduction
        const uint32_t base = 1u + (uint32_t)s * PPN_DRAWS_PER_SEG + 1u;   // first sample draw (even)
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int q = lane; q < 500; q += 64) {
            double u0, u1;
            if (fed) { u0 = fed[base + 2 * q]; u1 = fed[base + 2 * q + 1]; }
            else philox_double2(prm.seed, STREAM_PATH, pid, (base >> 1) + (uint32_t)q, u0, u1);
            const double y0 = u0 * 10.0 - 5.0, y1 = u1 * 10.0 - 5.0;       // PathSeg.py:23
            const int i0 = 2 * q, i1 = 2 * q + 1;
            a0 += prm.W[0 * 1000 + i0] * y0; a0 += prm.W[0 * 1000 + i1] * y1;
            a1 += prm.W[1 * 1000 + i0] * y0; a1 += prm.W[1 * 1000 + i1] * y1;
            a2 += prm.W[2 * 1000 + i0] * y0; a2 += prm.W[2 * 1000 + i1] * y1;
            a3 += prm.W[3 * 1000 + i0] * y0; a3 += prm.W[3 * 1000 + i1] * y1;
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
        if (lane == 0) { fit_part[s][0] = a0; fit_part[s][1] = a1; fit_part[s][2] = a2; fit_part[s][3] = a3; }
    }
    __syncthreads();
    if (tid < PPN_SEGS) {
        const int s = tid;
        const uint32_t b = 1u + (uint32_t)s * PPN_DRAWS_PER_SEG;
        const double uf = fed ? fed[b] : philox_double(prm.seed, STREAM_PATH, pid, b);
        const double ue = fed ? fed[b + 1001] : philox_double(prm.seed, STREAM_PATH, pid, b + 1001);
        const bool st = path_straight || (uf < 0.2);             // PathSeg.py:19
        double poly[5];
        for (int k = 0; k < 4; ++k) poly[k] = fit_part[s][k];
        poly[4] = 0.0;                                           // PathSeg.py:28
        if (st) { poly[0] = 0.0; poly[1] = 0.0; poly[2] = 0.0; } // PathSeg.py:29-31
        const double E = ue * 7.0 + 0.0;                         // PathSeg.py:32
        for (int k = 0; k < 5; ++k) S.poly[s][k] = poly[k];
        S.pder[s][0] = poly[0] * 4.0; S.pder[s][1] = poly[1] * 3.0;
        S.pder[s][2] = poly[2] * 2.0; S.pder[s][3] = poly[3] * 1.0;
        S.endpoint[s] = E;
        S.straight[s] = st ? 1 : 0;
        const double yE = horner4(poly, E);
        S.t0[s][0] = E; S.t0[s][1] = yE;
        S.grad[s][0] = horner3(S.pder[s], 0.0);
        S.grad[s][1] = horner3(S.pder[s], E);
    }
    __syncthreads();
    // PathSeg.length (PathSeg.py:49-58): the 100-sample polyline of every segment.  Its 100 pieces (99 between samples + the one
    // to the end point) are independent — each a Horner pair and a double square root — so all 1 000 are computed in parallel into
    // the (not yet written) path-point area, and only the SUM stays sequential, in the reference's order: as ten 100-iteration
    // chains on ten lanes this loop was a tenth of the kernel's latency.
    {
        double* dl = reinterpret_cast<double*>(paths_lds);           // [PPN_SEGS][100]; slot 0 = the closing piece
        for (int q = tid; q < PPN_SEGS * 100; q += NT) {
            const int s = q / 100, j = q - s * 100;
            const double E = S.endpoint[s];
            const int jp = j ? j - 1 : 99;
            const double xp = ((double)jp / 100.0) * (E - 0.0), yp = horner4(S.poly[s], xp);
            double x = E, y = S.t0[s][1];
            if (j) { x = ((double)j / 100.0) * (E - 0.0); y = horner4(S.poly[s], x); }
            dl[q] = dist2d(x, y, xp, yp);
        }
        __syncthreads();
        if (tid < PPN_SEGS) {
            double len = 0.0;
            for (int j = 1; j < 100; ++j) len = len + dl[tid * 100 + j];
            len = len + dl[tid * 100];
            if (O.seg_length) O.seg_length[(size_t)p * PPN_SEGS + tid] = len;
        }
    }
    __syncthreads();

    PPN_PSTAMP(0);
    // ------------------------------------------------------------------ A2: chaining (serial, tiny)
    // the arctangents, cosines and sines are per segment (ten lanes at once); only the two running sums are serial
    if (tid < PPN_SEGS) { at2[tid][0] = atan(S.grad[tid][0]); at2[tid][1] = atan(S.grad[tid][1]); }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0;
        S.ang[0] = 0.0;
        for (int i = 1; i < PPN_SEGS; ++i) {                     // angle_abs, Path.py:276-289
            a = a + (at2[i - 1][1] - at2[i][0]);
            S.ang[i] = a;
        }
    }
    __syncthreads();
    if (tid < PPN_SEGS) { S.cs[tid] = cos(S.ang[tid]); S.sn[tid] = sin(S.ang[tid]); }
    __syncthreads();
    if (tid == 0) {
        double tx = 0.0, ty = 0.0;                               // translation_seg, Path.py:291-299
        for (int i = 0; i < PPN_SEGS; ++i) {
            S.trans[i][0] = tx; S.trans[i][1] = ty;
            double rx, ry;
            if (i != 0) rot2(S.cs[i], S.sn[i], S.t0[i][0], S.t0[i][1], rx, ry);
            else { rx = S.t0[0][0]; ry = S.t0[0][1]; }
            tx = tx + rx; ty = ty + ry;
        }
        S.segpoint[0][0] = 0.0; S.segpoint[0][1] = 0.0;
        for (int i = 0; i < PPN_SEGS; ++i) {                     // Path.py:86-90
            double x = S.t0[i][0], y = S.t0[i][1], rx, ry;
            if (i != 0) rot2(S.cs[i], S.sn[i], x, y, rx, ry); else { rx = x; ry = y; }
            S.segpoint[i + 1][0] = rx + S.trans[i][0];
            S.segpoint[i + 1][1] = ry + S.trans[i][1];
        }
    }
    __syncthreads();
    if (tid < PPN_SEGS) {
        const size_t o = (size_t)p * PPN_SEGS + tid;
        for (int k = 0; k < 5; ++k) O.seg_poly[o * 5 + k] = S.poly[tid][k];
        O.seg_endpoint[o] = S.endpoint[tid];
        O.seg_rotation[o] = S.ang[tid];
        O.seg_translation[o * 2] = S.trans[tid][0];
        O.seg_translation[o * 2 + 1] = S.trans[tid][1];
        O.seg_straight[o] = S.straight[tid];
        if (O.seg_grad) { O.seg_grad[o * 2] = S.grad[tid][0]; O.seg_grad[o * 2 + 1] = S.grad[tid][1]; }
    }
    if (tid < PPN_SEGS + 1) {
        O.segpoint_world[((size_t)p * 11 + tid) * 2] = S.segpoint[tid][0];
        O.segpoint_world[((size_t)p * 11 + tid) * 2 + 1] = S.segpoint[tid][1];
    }

    PPN_PSTAMP(1);
    // path points (Path.plot, Path.py:256-260): 100 samples per segment
    for (int q = tid; q < PPN_PATH_POINTS; q += NT) {
        const int s = q / 100, j = q - s * 100;
        const double x = ((double)j / 100.0) * S.endpoint[s];
        const double y = horner4(S.poly[s], x);
        double rx, ry;
        if (s != 0) rot2(S.cs[s], S.sn[s], x, y, rx, ry); else { rx = x; ry = y; }
        rx = rx + S.trans[s][0];
        ry = ry + S.trans[s][1];
        pp[q][0] = rx; pp[q][1] = ry;
        O.pathpoint_world[((size_t)p * PPN_PATH_POINTS + q) * 2] = rx;
        O.pathpoint_world[((size_t)p * PPN_PATH_POINTS + q) * 2 + 1] = ry;
        // convexhull input, Path.py:390; clamped to +-16383 so the hull walk's products are exact in 32 bits
        // (a path that far outside the 2R canvas is not a valid instance: flagged below)
        lat[q][0] = max(-16383, min(16383, (int)rint(rx / step_c2i + Rd)));
        lat[q][1] = max(-16383, min(16383, (int)rint(ry / step_c2i + Rd)));
    }
    __syncthreads();
    {
        double part = 0.0;                                       // Path.Length, Path.py:93-94
        for (int q = tid; q < PPN_PATH_POINTS - 1; q += NT)
            part += dist2d(pp[q][0], pp[q][1], pp[q + 1][0], pp[q + 1][1]);
        const double len = block_sum<NW>(part, red_v);
        double mstep = 0.0;
        for (int q = tid; q < PPN_PATH_POINTS - 1; q += NT)
            mstep = fmax(mstep, dist2d(pp[q][0], pp[q][1], pp[q + 1][0], pp[q + 1][1]));
        if (tid == 0) mstep = fmax(mstep, dist2d(pp[PPN_PATH_POINTS - 1][0], pp[PPN_PATH_POINTS - 1][1],
                                                  S.segpoint[PPN_SEGS][0], S.segpoint[PPN_SEGS][1]));
        mstep = block_max<NW>(mstep, red_v);
        if (tid == 0) {
            O.length[p] = len;
            O.straight[p] = path_straight ? 1 : 0;
            O.max_step_px[p] = mstep / step_c2i;
        }
    }
    const double Ex = S.segpoint[PPN_SEGS][0], Ey = S.segpoint[PPN_SEGS][1];

    PPN_PSTAMP(2);
    // ------------------------------------------------------------------ A3 + A4: boundary rays -> canvas
    {
        const int n_steps = (int)rint(0.8 * clearance / step_len);      // Path.py:119,398
        double u0x, u0y, ulx, uly, tnx, tny;
        {   // up.point[0][0] and up.point[9][49] seed the semicircular caps (Path.py:334-343)
            double px, py;
            boundary_sample(S, 0, 0, px, py, tnx, tny);
            u0x = px - 0.5 * clearance * tnx; u0y = py - 0.5 * clearance * tny;
            boundary_sample(S, PPN_SEGS - 1, 49, px, py, tnx, tny);
            ulx = px - 0.5 * clearance * tnx; uly = py - 0.5 * clearance * tny;
        }
        for (int r = tid; r < PPN_BOUNDARY_POINTS; r += NT) {
            double sx, sy, dx, dy;
            int slot;
            if (r < 50) {                                        // init cap
                const double a = 3.141592653589793 / 50.0 * (double)(r + 1);
                rot2(cos(a), sin(a), u0x, u0y, sx, sy);
                const double n = sqrt(sx * sx + sy * sy);
                dx = -step_len * sx / n; dy = -step_len * sy / n;       // Path.py:121
                slot = 49 - r;
            } else if (r < 100) {                                // end cap
                const int i = r - 50;
                const double a = -3.141592653589793 / 50.0 * (double)(i + 1);
                double qx, qy;
                rot2(cos(a), sin(a), ulx - Ex, uly - Ey, qx, qy);
                sx = qx + Ex; sy = qy + Ey;
                const double vx = Ex - sx, vy = Ey - sy;
                const double n = sqrt(vx * vx + vy * vy);
                dx = step_len * vx / n; dy = step_len * vy / n;         // Path.py:124-125
                slot = 550 + i;
            } else {
                const bool up = r < 600;
                const int q = up ? r - 100 : r - 600;
                const int s = q / 50, j = q - s * 50;
                double px, py, nx, ny;
                boundary_sample(S, s, j, px, py, nx, ny);
                if (up) {
                    sx = px - 0.5 * clearance * nx; sy = py - 0.5 * clearance * ny;
                    dx = step_len * nx; dy = step_len * ny;
                    slot = 50 + q;
                } else {
                    sx = px + 0.5 * clearance * nx; sy = py + 0.5 * clearance * ny;
                    dx = step_len * (-1.0 * nx); dy = step_len * (-1.0 * ny);
                    slot = 600 + (499 - q);
                }
            }
            if (O.boundary_world) {
                O.boundary_world[((size_t)p * PPN_BOUNDARY_POINTS + slot) * 2] = sx;
                O.boundary_world[((size_t)p * PPN_BOUNDARY_POINTS + slot) * 2 + 1] = sy;
            }
            for (int k = 0; k < n_steps; ++k) {                  // free_space_bydirection
                const double qx = sx + (double)k * dx, qy = sy + (double)k * dy;
                const int i0 = (int)rint(qx / step_c2i + Rd);
                const int i1 = (int)rint(qy / step_c2i + Rd);
                if (i0 > 0 && i0 < 2 * R && i1 > 0 && i1 < 2 * R) {     // strict, Path.py:400
                    const int bit = i0 * 2 * R + i1;
                    atomicOr(&canvas[bit >> 5], 1u << (bit & 31));
                } else break;
            }
        }
    }
    __syncthreads();
    if (O.canvas_bits)
        for (int w = tid; w < canvas_words; w += NT) O.canvas_bits[(size_t)p * canvas_words + w] = canvas[w];

    PPN_PSTAMP(3);
    // ------------------------------------------------------------------ A5: exact integer hull (gift wrapping)
    // ALL FOUR waves walk the hull together (round 5; rounds 1-4: one wave with 16 points per lane while three idled — 62 k of the
    // kernel's 265 k cycles): a lane keeps 4 lattice points in registers and proposes its best successor; each wave reduces its 64
    // proposals with a butterfly of exchanges under "to the right of cur->cand, or collinear and farther"; the four
    // wave candidates meet in LDS (one slot set per vertex parity: ONE barrier per vertex) and every thread picks the same winner with
    // the same exact integer predicate.  The predicate is a strict total preorder on directions from cur (ties: the farther point;
    // equal points are the same vertex), so the successor — and the hull — is the one the serial walk finds.
    __shared__ int hull_meta[2];                                 // hn, flags
    __shared__ int hull_cand[2][NW][4];                          // [vertex parity][wave]: have, x, y
    {
        constexpr int PPL = (PPN_PATH_POINTS + NT - 1) / NT;     // points per lane
        int px[PPL], py[PPL];
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            const int q = min(tid + NT * k, PPN_PATH_POINTS - 1);    // tail lanes repeat the last point (harmless duplicate)
            px[k] = lat[q][0]; py[k] = lat[q][1];
        }
        // start = lexicographically smallest lattice point
        int sx0 = px[0], sy0 = py[0];
#pragma unroll
        for (int k = 1; k < PPL; ++k) if (px[k] < sx0 || (px[k] == sx0 && py[k] < sy0)) { sx0 = px[k]; sy0 = py[k]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int ox = __shfl_xor(sx0, o, 64), oy = __shfl_xor(sy0, o, 64);
            if (ox < sx0 || (ox == sx0 && oy < sy0)) { sx0 = ox; sy0 = oy; }
        }
        if (lane == 0) { hull_cand[0][wv][1] = sx0; hull_cand[0][wv][2] = sy0; }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int ox = hull_cand[0][w][1], oy = hull_cand[0][w][2];
            if (ox < sx0 || (ox == sx0 && oy < sy0)) { sx0 = ox; sy0 = oy; }
        }
        __syncthreads();                                         // slot set 0 is written again by vertex 0
        // `a` beats `b` as the successor of cur: a is to the right of cur -> b, or collinear with it and farther
        // (branch-free on purpose: with early returns hipcc emitted ~40 instructions and two exec-mask branches per call, 850
        // instructions per hull vertex; 14 calls per vertex)
        auto beats = [&](int ax, int ay, int bx_, int by_, int cx, int cy) -> bool {
            const int ux = ax - cx, uy = ay - cy, vx = bx_ - cx, vy = by_ - cy;
            const int cr = __mul24(vx, uy) - __mul24(vy, ux);
            const int da = __mul24(ux, ux) + __mul24(uy, uy), db = __mul24(vx, vx) + __mul24(vy, vy);
            return (cr < 0) | ((cr == 0) & (da > db));
        };
        int cx = sx0, cy = sy0, n = 0;
        uint32_t hflags = 0;
        while (true) {                                           // block-uniform control flow: every thread sees the same cur / winner
            if (tid == 0) { hull_i[n][0] = cx; hull_i[n][1] = cy; }
            const int par = n & 1;
            ++n;
            // this lane's proposal among its own points
            int bx = 0, by = 0, have = 0;
#pragma unroll
            for (int k = 0; k < PPL; ++k) {
                const int rx = px[k], ry = py[k];
                const bool take = ((rx != cx) | (ry != cy)) & ((have == 0) | beats(rx, ry, bx, by, cx, cy));
                bx = take ? rx : bx; by = take ? ry : by; have = take ? 1 : have;
            }
            // this wave's candidate: a butterfly over the 64 proposals (6 exchanges).  (Rounds 1-4 refined a candidate with ballots,
            // "first lane whose proposal beats it": consecutive lanes hold consecutive path points, whose directions from cur are
            // monotone along the path, so that loop advanced ONE lane per round — up to 64 rounds per vertex, 2 900 cycles of them.)
            int kx = bx, ky = by, kh = have;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const int ox = __shfl_xor(kx, o, 64), oy = __shfl_xor(ky, o, 64), oh = __shfl_xor(kh, o, 64);
                const bool take = (oh != 0) & ((kh == 0) | beats(ox, oy, kx, ky, cx, cy));
                kx = take ? ox : kx; ky = take ? oy : ky; kh = take ? 1 : kh;
            }
            const int hv = kh;
            if (lane == 0) { hull_cand[par][wv][0] = hv ? 1 : 0; hull_cand[par][wv][1] = kx; hull_cand[par][wv][2] = ky; }
            __syncthreads();
            // the four wave candidates -> the successor (the same computation in every thread)
            int wx = 0, wy = 0, whave = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int ch = hull_cand[par][w][0], ox = hull_cand[par][w][1], oy = hull_cand[par][w][2];
                const bool take = (ch != 0) & ((whave == 0) | beats(ox, oy, wx, wy, cx, cy));
                wx = take ? ox : wx; wy = take ? oy : wy; whave = take ? 1 : whave;
            }
            if (!whave) break;                                   // every point coincides with cur
            if (wx == sx0 && wy == sy0) break;
            if (n >= PPN_MAX_HULL) { hflags |= PPN_FLAG_HULL_CAP; break; }
            cx = wx; cy = wy;
        }
        if (tid == 0) { hull_meta[0] = n; hull_meta[1] = (int)hflags; }
    }
    __syncthreads();
    const int hn = hull_meta[0];
    flags |= (uint32_t)hull_meta[1];
    __syncthreads();
    // Reference vertex order (replay mode): Qhull starts the same CCW cycle at an implementation-defined vertex (Path.py:392-393),
    // and search_isle / set_obstacles walk the edges — and consume torch.rand — in that order (Path.py:463-537).  The caller names
    // the start as an index into the canonical cycle; everything downstream (ConvexHull, isles, pocket obstacles) follows.
    {
        const int hs_raw = prm.hull_start ? prm.hull_start[p] : -1;
        const int hs = (hs_raw > 0 && hn > 0) ? hs_raw % hn : 0;                 // block-uniform
        if (hs) {
            int vx = 0, vy = 0;
            if (tid < hn) { const int src = (tid + hs) % hn; vx = hull_i[src][0]; vy = hull_i[src][1]; }
            __syncthreads();
            if (tid < hn) { hull_i[tid][0] = vx; hull_i[tid][1] = vy; }
            __syncthreads();
        }
    }

    PPN_PSTAMP(4);
    // ------------------------------------------------------------------ A6: normalisation
    if (tid == 0) {
        const double rotation = atan(Ey / Ex) / 3.141592653589793 * 180.0 + (-135.0);   // Path.py:159
        const double rad = -rotation / 180.0 * 3.141592653589793;
        const double c = cos(rad), s = sin(rad);
        double mx = 0.0, my = 0.0;
        for (int i = 0; i < hn; ++i) {                           // Path.py:162-164
            double hx, hy;
            rot2(c, s, (double)hull_i[i][0] - Rd, (double)hull_i[i][1] - Rd, hx, hy);
            hull[i][0] = hx + Rd; hull[i][1] = hy + Rd;
            mx += hull[i][0]; my += hull[i][1];
        }
        mx = mx / (double)hn; my = my / (double)hn;              // torch.mean, Path.py:167
        const double t0 = Rd / 2.0 - mx, t1 = Rd / 2.0 - my;      // (t_row, t_col)
        for (int i = 0; i < hn; ++i) { hull[i][0] = hull[i][0] + t0; hull[i][1] = hull[i][1] + t1; }
        bc[0] = rotation; bc[1] = c; bc[2] = s; bc[3] = t0; bc[4] = t1;
        // T.RandomRotation(degrees=(a, a)).get_params returns float(torch.empty(1).uniform_(a, a).item()): the angle reaches
        // F.rotate ROUNDED TO FLOAT32 (Path.py:160-161).  Only the raster rotation: the points rotate with the double angle.
        const double b = (double)(float)rotation * PPN_DEG2RAD;  // rotate_nearest(canvas, -rotation): F.rotate's matrix is
        bc[5] = cos(b); bc[6] = sin(b);                          // [cos b, sin b, 0; -sin b, cos b, 0], b = radians(rotation)
        O.rotation[p] = rotation;
        O.trans_rc[(size_t)p * 2] = t0; O.trans_rc[(size_t)p * 2 + 1] = t1;
        O.hull_n[p] = hn;
    }
    __syncthreads();
    const double nc = bc[1], ns = bc[2], t_row = bc[3], t_col = bc[4];
    if (tid < PPN_MAX_HULL) {
        const bool v = tid < hn;
        O.hull[((size_t)p * PPN_MAX_HULL + tid) * 2] = v ? hull[tid][0] : 0.0;
        O.hull[((size_t)p * PPN_MAX_HULL + tid) * 2 + 1] = v ? hull[tid][1] : 0.0;
        if (O.hull_raw) {
            O.hull_raw[((size_t)p * PPN_MAX_HULL + tid) * 2] = v ? (double)hull_i[tid][0] : 0.0;
            O.hull_raw[((size_t)p * PPN_MAX_HULL + tid) * 2 + 1] = v ? (double)hull_i[tid][1] : 0.0;
        }
    }
    if (tid < PPN_SEGS + 1) {                                    // Path.py:180-182
        double rx, ry;
        rot2(nc, ns, S.segpoint[tid][0], S.segpoint[tid][1], rx, ry);
        O.segpoint_image[((size_t)p * 11 + tid) * 2] = rint(rx / step_c2i + Rd) + t_row;
        O.segpoint_image[((size_t)p * 11 + tid) * 2 + 1] = rint(ry / step_c2i + Rd) + t_col;
    }
    for (int q = tid; q < PPN_PATH_POINTS; q += NT) {            // Path.py:183-185
        double rx, ry;
        rot2(nc, ns, pp[q][0], pp[q][1], rx, ry);
        rx = rint(rx / step_c2i + Rd) + t_row;
        ry = rint(ry / step_c2i + Rd) + t_col;
        pp[q][0] = rx; pp[q][1] = ry;                            // each thread owns its q: no race
        O.pathpoint_image[((size_t)p * PPN_PATH_POINTS + q) * 2] = rx;
        O.pathpoint_image[((size_t)p * PPN_PATH_POINTS + q) * 2 + 1] = ry;
    }

    PPN_PSTAMP(5);
    // corridor mask Path.Space: rotate (nearest, about the canvas centre) then translate + crop, composed per output
    // pixel, 32 pixels (one mask word) per thread iteration.  The corridor covers a few per cent of the canvas, so a
    // coarse map (one bit per 16x16 canvas tile) first rules most words out: the 32 pixels of a word map onto a straight
    // run of canvas pixels between the images of its two ends (the inverse map is affine in the column), so if every
    // tile of that run's bounding box (+1 px for the roundings) is empty the word is zero.  The surviving words are
    // compacted into a list and processed densely (a cull without compaction saves nothing: one live lane holds its wave).
    {
        __shared__ int n_live;
        uint32_t* ctile = reinterpret_cast<uint32_t*>(lat);                  // [128] tile occupancy bits; lat is dead after the hull
        unsigned short* live = reinterpret_cast<unsigned short*>(ctile + 128);   // word indices that may be non-zero (3744 slots; more fall back in place)
        const int tps = 2 * R / 16;                                 // tiles per canvas side
        for (int w = tid; w < 128; w += NT) ctile[w] = 0u;
        if (tid == 0) n_live = 0;
        __syncthreads();
        const int cwpr = 2 * R / 32;
        for (int w = tid; w < canvas_words; w += NT) {
            const uint32_t v = canvas[w];
            if (v) {
                const int row = w / cwpr, ty = row >> 4, tx = (w - row * cwpr) * 2;
                if (v & 0xffffu) { const int t = ty * tps + tx; atomicOr(&ctile[t >> 5], 1u << (t & 31)); }
                if (v >> 16) { const int t = ty * tps + tx + 1; atomicOr(&ctile[t >> 5], 1u << (t & 31)); }
            }
        }
        __syncthreads();
        // torchvision's float32 maps (ppn_device.h tv_axis / tv_src): translate [tx, ty] = [t_col, t_row] on the 2R canvas,
        // then the rotation about its centre
        const int S2 = 2 * R;
        const TvAxis tcol = tv_axis(1.0, 0.0, -t_col, S2), trow = tv_axis(0.0, 1.0, -t_row, S2);
        const TvAxis rcol = tv_axis(bc[5], bc[6], 0.0, S2), rrow = tv_axis(-bc[6], bc[5], 0.0, S2);
        const float ctr = 0.5f - (float)R;                       // pixel k of the canvas has centre coordinate k + ctr
        const int words = R * R / 32, wpr = R / 32;
        const int live_cap = (PPN_PATH_POINTS * 8 - 512) / 2;      // entries the lat region holds behind ctile
        for (int w = tid; w < words; w += NT) {
            const int i = w / wpr, j0 = (w - i * wpr) * 32;
            const int i1 = tv_src(trow, 0.0f, (float)i + ctr);   // translate: row i of the output shows row i1 of the rotated canvas
            bool maybe = false;
            if (i1 >= 0 && i1 < S2) {
                const float yo = (float)i1 + ctr;
                const int j1a = tv_src(tcol, (float)j0 + ctr, 0.0f), j1b = tv_src(tcol, (float)(j0 + 31) + ctr, 0.0f);
                const float xa = (float)j1a + ctr, xb = (float)j1b + ctr;
                const int ja = tv_src(rcol, xa, yo), ia = tv_src(rrow, xa, yo);
                const int jb = tv_src(rcol, xb, yo), ib = tv_src(rrow, xb, yo);
                const int x0 = max(min(ja, jb) - 1, 0), x1 = min(max(ja, jb) + 1, S2 - 1);
                const int y0 = max(min(ia, ib) - 1, 0), y1 = min(max(ia, ib) + 1, S2 - 1);
                for (int ty = y0 >> 4; ty <= (y1 >> 4); ++ty)
                    for (int tx = x0 >> 4; tx <= (x1 >> 4); ++tx) {
                        const int t = ty * tps + tx;
                        maybe = maybe || ((ctile[t >> 5] >> (t & 31)) & 1u);
                    }
            }
            int slot = -1;
            if (maybe) slot = atomicAdd(&n_live, 1);
            if (maybe && slot < live_cap) live[slot] = (unsigned short)w;
            else if (!maybe) O.space_bits[(size_t)p * words + w] = 0u;
            else slot = -2;                                       // list full (cannot happen for R <= 256): do it here
            if (slot == -2) {
                uint32_t m = 0u;
                const float yo = (float)i1 + ctr;
                for (int b = 0; b < 32; ++b) {
                    const int j1 = tv_src(tcol, (float)(j0 + b) + ctr, 0.0f);
                    if (j1 < 0 || j1 >= S2) continue;
                    const float xo = (float)j1 + ctr;
                    const int jj = tv_src(rcol, xo, yo), ii = tv_src(rrow, xo, yo);
                    if (ii < 0 || ii >= S2 || jj < 0 || jj >= S2) continue;
                    const int bit = ii * S2 + jj;
                    m |= ((canvas[bit >> 5] >> (bit & 31)) & 1u) << b;
                }
                O.space_bits[(size_t)p * words + w] = m;
            }
        }
        __syncthreads();
        const int nl = min(n_live, live_cap);
        for (int e = tid; e < nl; e += NT) {
            const int w = live[e];
            const int i = w / wpr, j0 = (w - i * wpr) * 32;
            const int i1 = tv_src(trow, 0.0f, (float)i + ctr);
            const float yo = (float)i1 + ctr;
            uint32_t m = 0u;
            for (int b = 0; b < 32; ++b) {
                const int j1 = tv_src(tcol, (float)(j0 + b) + ctr, 0.0f);
                if (j1 < 0 || j1 >= S2) continue;
                const float xo = (float)j1 + ctr;
                const int jj = tv_src(rcol, xo, yo), ii = tv_src(rrow, xo, yo);
                if (ii < 0 || ii >= S2 || jj < 0 || jj >= S2) continue;
                const int bit = ii * S2 + jj;
                m |= ((canvas[bit >> 5] >> (bit & 31)) & 1u) << b;
            }
            O.space_bits[(size_t)p * words + w] = m;
        }
    }
    __syncthreads();

    PPN_PSTAMP(6);
    // ------------------------------------------------------------------ A7: search_isle
    __shared__ int isles[PPN_MAX_ISLES][2];
    int n_isles = 0;
    if (!path_straight) {
        const int thr = (int)rint(clearance / step_len * 0.2);   // Path.py:533
        for (int i = 0; i < hn; ++i) {
            const int j = (i == hn - 1) ? 0 : i + 1;
            const double ex = hull[j][0] - hull[i][0], ey = hull[j][1] - hull[i][1];
            if (!(sqrt(ex * ex + ey * ey) > 5.0 / step_len)) continue;  // Path.py:512
            MinIdx m0{1e300, 0x7fffffff}, m1{1e300, 0x7fffffff};
            for (int q = tid; q < PPN_PATH_POINTS; q += NT) {
                const double a = dist2d(pp[q][0], pp[q][1], hull[i][0], hull[i][1]);
                const double b = dist2d(pp[q][0], pp[q][1], hull[j][0], hull[j][1]);
                m0 = min_idx(m0, MinIdx{a, q});
                m1 = min_idx(m1, MinIdx{b, q});
            }
            m0 = block_min_idx<NW>(m0, red_v, red_i);
            m1 = block_min_idx<NW>(m1, red_v, red_i);
            const int lo = min(m0.i, m1.i), hi = max(m0.i, m1.i);
            if (hi == lo) { flags |= PPN_FLAG_EMPTY_ISLE; continue; }
            const int len = hi - lo;
            const double vx = pp[lo][0] - pp[hi - 1][0], vy = pp[lo][1] - pp[hi - 1][1];
            const double nrm = sqrt(vx * vx + vy * vy);
            if (nrm == 0.0) continue;
            const double d0x = vx / nrm, d0y = vy / nrm;
            const double dirx = d0y, diry = -d0x;
            int first = 0x7fffffff;
            for (int k = tid; k < len; k += NT) {
                const double dis = fabs((pp[lo + k][0] - pp[lo][0]) * dirx + (pp[lo + k][1] - pp[lo][1]) * diry);
                if (dis > (double)thr + PPN_TIE_EPS) { first = k; break; }
            }
            first = block_min_int<NW>(first, red_i);
            const int k = first == 0x7fffffff ? len - 1 : first;
            const bool isle = (pp[lo + k][0] != pp[hi - 1][0]) || (pp[lo + k][1] != pp[hi - 1][1]);   // Path.py:535
            if (isle) {
                if (n_isles < PPN_MAX_ISLES) {
                    if (tid == 0) { isles[n_isles][0] = lo; isles[n_isles][1] = hi; }
                    ++n_isles;
                } else flags |= PPN_FLAG_ISLE_CAP;
            }
        }
    }
    __syncthreads();
    if (tid < PPN_MAX_ISLES) {
        O.isles[((size_t)p * PPN_MAX_ISLES + tid) * 2] = tid < n_isles ? isles[tid][0] : 0;
        O.isles[((size_t)p * PPN_MAX_ISLES + tid) * 2 + 1] = tid < n_isles ? isles[tid][1] : 0;
    }

    PPN_PSTAMP(7);
    // ------------------------------------------------------------------ A8: set_obstacles
    int n_obs = 0, n_pocket_draws = 0;
    {
        const double c_px = clearance / prm.map_size * Rd;               // Path.py:490
        const float size_clearance_f = (float)(clearance / prm.map_size * Rd * 1.1);   // Path.py:465
        const float c_px_f = (float)c_px;
        uint32_t fdraw = 0;                                              // torch.rand counter
        const float* pfed = prm.pocket ? prm.pocket + (size_t)p * prm.pocket_stride : nullptr;
        auto next_float = [&]() -> float {
            float v;
            if (pfed) {
                if ((int)fdraw < prm.pocket_stride) v = pfed[fdraw];
                else { v = 0.5f; flags |= PPN_FLAG_POCKET_DRAWS; }           // fed draws ran out: say so (the caller re-launches)
            } else v = philox_float(prm.seed, STREAM_POCKET, pid, fdraw);
            ++fdraw;
            return v;
        };
        for (int is = 0; is < n_isles; ++is) {
            const int lo = isles[is][0], hi = isles[is][1], len = hi - lo;
            const double cxm = (pp[lo][0] + pp[hi - 1][0]) / 2.0, cym = (pp[lo][1] + pp[hi - 1][1]) / 2.0;
            const double vx = pp[lo][0] - pp[hi - 1][0], vy = pp[lo][1] - pp[hi - 1][1];
            const double vn = sqrt(vx * vx + vy * vy);
            const double dtx = vx / vn, dty = vy / vn;
            double dnx = dty, dny = -dtx;
            const int mi = lo + len / 2;
            const double mdx = pp[mi][0] - cxm, mdy = pp[mi][1] - cym;
            if (!(mdx * dnx + mdy * dny < 0.0)) { dnx = -dnx; dny = -dny; }     // Path.py:470
            double mymax = -1.0;
            for (int k = tid; k < len; k += NT) {
                const double dis = fabs((pp[lo + k][0] - pp[lo][0]) * dnx + (pp[lo + k][1] - pp[lo][1]) * dny);
                mymax = fmax(mymax, dis);
            }
            const double dmax = block_max<NW>(mymax, red_v);
            int first = 0x7fffffff;
            for (int k = tid; k < len; k += NT) {
                const double dis = fabs((pp[lo + k][0] - pp[lo][0]) * dnx + (pp[lo + k][1] - pp[lo][1]) * dny);
                if (dis >= dmax - PPN_TIE_EPS) { first = k; break; }
            }
            first = block_min_int<NW>(first, red_i);
            const double size_max = dmax * 2.0;
            const float size_max_f = (float)size_max;
            const double pkx = pp[lo + first][0], pky = pp[lo + first][1];
            float obs_sum = 0.0f;
            int cnt = 0, tries = 0;
            float size_pre_f = 0.0f;
            double coordx = 0.0, coordy = 0.0;
            while ((double)obs_sum < size_max) {                         // Path.py:478
                if (tries >= PPN_POCKET_TRY_CAP) { flags |= PPN_FLAG_POCKET_CAP; break; }
                ++tries;
                const float radius = (next_float() * size_max_f) / 2.0f;
                const float rn = cnt ? next_float() : 1.0f;
                float acc = radius + size_pre_f;
                acc = acc + (cnt == 0 ? size_clearance_f : 0.0f);
                float motion = rn * acc;
                const float alt = radius - obs_sum;
                if (alt > motion) motion = alt;                          // python max(motion, alt)
                const double bx = cnt == 0 ? pkx : coordx, by = cnt == 0 ? pky : coordy;
                coordx = bx + (double)motion * dnx;
                coordy = by + (double)motion * dny;
                if (cnt) {
                    const float jit = ((((next_float() - 0.5f) / 0.5f) * radius) / 2.0f);
                    coordx = coordx + (double)jit * dtx;
                    coordy = coordy + (double)jit * dty;
                }
                double mymin = 1e300;
                for (int q = tid; q < PPN_PATH_POINTS / 2; q += NT)         // odd-indexed points, Path.py:487-489
                    mymin = fmin(mymin, dist2d(pp[2 * q + 1][0], pp[2 * q + 1][1], coordx, coordy));
                const double md = block_min<NW>(mymin, red_v);
                double rad_out = (double)radius;
                bool clipped = false;
                if (md < (double)(radius + c_px_f)) { rad_out = md - c_px; clipped = true; }    // Path.py:490-491
                if (rad_out > 0.0) {
                    obs_sum = obs_sum + motion;
                    ++cnt;
                    size_pre_f = clipped ? (float)rad_out : radius;
                    if (n_obs < PPN_MAX_POCKET) {
                        if (tid == 0) {
                            double* o = O.obstacles + ((size_t)p * PPN_MAX_POCKET + n_obs) * 3;
                            o[0] = coordy; o[1] = coordx; o[2] = rad_out;   // [col,row,r], Path.py:495
                        }
                        ++n_obs;
                    } else flags |= PPN_FLAG_POCKET_FULL;
                }
            }
        }
        n_pocket_draws = (int)fdraw;
    }
    PPN_PSTAMP(8);
    if (tid == 0) {
        O.n_isles[p] = n_isles;
        O.n_obstacles[p] = n_obs;
        O.flags[p] = flags;
        if (O.pocket_draws_used) O.pocket_draws_used[p] = n_pocket_draws;
    }
}

}  // namespace ppn
