// edage_maps.hip — stage B of EDaGe-PP on gfx950: one 128-thread workgroup per map instance.
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

#ifdef PPN_PHASE_TIMING
// diagnostic build only (make timing): per-phase s_memtime sums, never read by the kernel itself
__device__ unsigned long long g_phase_cycles[16];
#define PPN_STAMP(idx) do { __syncthreads(); if (threadIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        atomicAdd(&g_phase_cycles[idx], t_ - t_prev_); t_prev_ = t_; } } while (0)
#define PPN_STAMP_INIT unsigned long long t_prev_ = __builtin_amdgcn_s_memtime()
#else
#define PPN_STAMP(idx) do {} while (0)
#define PPN_STAMP_INIT do {} while (0)
#endif

// Waves per map.  The placement is one wave's work, so a four-wave workgroup idles three waves through it, and at the chip's idle
// clocks (a 20-step run) two waves per map measured 3 % faster per bench step (0.204 -> 0.198 ms).  Once the chip has been under
// load for a few hundred milliseconds (bench.py's default window) four waves win by the same 3 % on one box, three alternations:
// 52.6-53.1 M instances/s (maps kernel 0.181-0.183 ms) with two, 54.0-54.8 M (0.174-0.176 ms) with four — the raster and the
// 64 KiB store phase finish sooner per map and fewer maps are in flight per CU.  `make w2` builds the two-wave library for A/B runs.
#ifndef PPN_MAPS_THREADS
#define PPN_MAPS_THREADS 256
#endif
#ifndef PPN_MAPS_WAVES_PER_EU
#define PPN_MAPS_WAVES_PER_EU 6   // register budget of the stage-B kernel: waves per SIMD it must allow (6 -> <= 80 VGPRs, no spills)
#endif

namespace {
constexpr int NT = 256;
constexpr int NW = NT / 64;
constexpr int MAX_OBS = 256 + PPN_MAX_POCKET;     // K <= 256
constexpr double PI = 3.141592653589793;
constexpr int NCOARSE = PPN_PATH_POINTS / 8 + 1;   // coarse points of the clearance filter: odd points 0, 4, .., 496 and 499
}

constexpr int MAPS_SCRATCH = (PPN_MAPS_THREADS / 64) * 256 + 512;        // filter scratch: fmin_w [waves][64] f32 + dref [64] f64
// bytes of the first dynamic-LDS region: the occupancy bit mask (phase 2)
__host__ __device__ constexpr int maps_region_bytes(int phase, int R) {
    return (phase & 2) ? (R * R / 8 > MAPS_SCRATCH ? R * R / 8 : MAPS_SCRATCH) : 0;   // fused: the filter scratch lives here first
}

// bytes of the region that holds the filter's coarse float points, then the raster's row tables (2*(K+64)+1 ints)
__host__ __device__ constexpr int maps_tab_bytes(int K) {
    const int rows = ((2 * (K + PPN_MAX_POCKET) + 1) * 4 + 7) & ~7;
    return rows > 1024 ? rows : 1024;
}

// The corridor-touch margin's rounding budget (see the clearance filter): five nearest-neighbour roundings of <= sqrt(2) / 2 px
// (corridor canvas, path-point lattice, two mask rotations, one fractional mask translation) and the half-pixel offset between
// the label frame and the pixel-centre frame: 6 * 0.7072 = 4.2432 (the float32 arithmetic of torchvision's maps moves a
// source coordinate by < 1e-3 px, inside the remainder).
#define PPN_TOUCH_ROUNDINGS 4.25

// 8 occupancy bits -> 8 grid bytes: bit k set (= occupied) -> byte k 0x00, clear -> 0xFF.  Read-only device table (2 KB, L1 /
// L2 resident): keeping it out of LDS leaves room for one more workgroup per CU.
struct ByteLut {
    uint64_t v[256];
    constexpr ByteLut() : v() {
        for (int e = 0; e < 256; ++e) {
            uint64_t x = 0;
            for (int k = 0; k < 8; ++k) x |= ((e >> k) & 1) ? 0ull : (0xFFull << (8 * k));
            v[e] = x;
        }
    }
};
__device__ const ByteLut g_byte_lut{};

// 4 mask bits -> 4 bytes (0x00 / 0xFF)
__device__ __forceinline__ uint32_t expand4(uint32_t b) {
    return (((b & 1u) | ((b & 2u) << 7) | ((b & 4u) << 14) | ((b & 8u) << 21))) * 255u;
}

// Stage B.  One workgroup per map.  PHASE 3 is the whole of MapGenerate.generate's loop body for a map; PHASE 1 (placement,
// labels, clearance filter: every output but `grid`) and PHASE 2 (obstacle lists -> `grid`, reading what PHASE 1 left in
// global memory) are the same code as two launches (ppn_edage_maps_place / _raster).
template <int PHASE>
__global__ __launch_bounds__(PPN_MAPS_THREADS, PPN_MAPS_WAVES_PER_EU) void edage_maps_kernel_t(MapsParams prm) {
    constexpr int NT = PPN_MAPS_THREADS, NW = NT / 64;                       // shadow the file-level 256 / 4 of the helper kernels
    // dynamic LDS carve (all 8-byte aligned):
    //   occw  [R*R/32] u32  the R*R-bit occupancy mask                                 (PHASE&2)
    //   cand  [K][3] f64    candidates (row, col, r)                                   (PHASE&1)
    //   obs   [K+64][3] f64 kept + pocket obstacles (col, row, r)
    //   poddf [2][128] f32  float copy of the 126 coarse odd points as planes x | y (filter
    //                       pre-pass); afterwards the raster's row tables
    //   (PHASE 1 only) 1.5 KB of filter scratch (fmin_w, dref); fused, the filter borrows the occupancy mask's bytes
    extern __shared__ uint64_t lds_raw[];
    __shared__ double bc[12];
    __shared__ int bci[16];                       // [0..3] placement, [4] kept count, [12] corridor-touch flag

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
    const int regionP_bytes = maps_region_bytes(PHASE, R);
    unsigned char* lds = reinterpret_cast<unsigned char*>(lds_raw);
    // the target path's 1000 image-frame points: read straight from global memory (100 placements share them in L2)
    const double2* pimg = reinterpret_cast<const double2*>(P.pathpoint_image) + (size_t)pj * PPN_PATH_POINTS;
    uint32_t* occw = reinterpret_cast<uint32_t*>(lds);
    const size_t cand_bytes = (PHASE & 1) ? (size_t)K * 24 : 0;
    double (*cand)[3] = reinterpret_cast<double (*)[3]>(lds + regionP_bytes);
    double (*obs)[3] = reinterpret_cast<double (*)[3]>(lds + regionP_bytes + cand_bytes);
    float* poddf = reinterpret_cast<float*>(lds + regionP_bytes + cand_bytes + (size_t)(K + PPN_MAX_POCKET) * 24);
    // [K+64][5] f64 per obstacle: its raster ellipse (cxp, cyp, ex, ey, (ex ey)^2) in the pixel-centre frame   (PHASE&2)
    double (*oell)[5] = reinterpret_cast<double (*)[5]>(reinterpret_cast<unsigned char*>(poddf) + maps_tab_bytes(K) + ((PHASE & 2) ? 0 : MAPS_SCRATCH));
    (void)oell;
    const uint64_t* lut = g_byte_lut.v;
    unsigned char* scratch = (PHASE & 2) ? lds : reinterpret_cast<unsigned char*>(poddf) + maps_tab_bytes(K);   // filter scratch: the occupancy mask's bytes when fused
    // regions that are dead when their second tenant arrives (barriers lie between):
    double (*hullc)[2] = reinterpret_cast<double (*)[2]>(poddf);          // hull - R/2 [64][2]: placement only, before the labels write poddf
    float (*fmin_w)[64] = reinterpret_cast<float (*)[64]>(scratch);       // [NW][64] per-wave partial minima (squared, float) of the filter
    double* dref = reinterpret_cast<double*>(scratch + NW * 256);         // [64] exact minima of the obstacles the coarse filter left undecided
    (void)mid; (void)pimg; (void)occw; (void)cand; (void)hullc; (void)bci; (void)fmin_w; (void)dref; (void)lane; (void)wv; (void)words;
    PPN_STAMP_INIT;

    // the hand-over from the placement half to the raster half
    int n_obs = 0, t0 = 0, t1 = 0;
    bool compose = false;

    if constexpr (PHASE & 1) {
        uint32_t flags = 0;

        const int hn = P.hull_n[pj];
        const int n_pocket = P.n_obstacles[pj];
        const uint32_t path_flags = P.flags[pj];
        // this thread's label points (L2 hits: 100 placements share a path): waves 1-3 have them in flight while the
        // placement runs, wave 0 asks for them once its loop is over (they would cost it 16 VGPRs across the loop)
        constexpr int PER = (PPN_PATH_POINTS + NT - 1) / NT;
        double2 pq_r[PER];
        if (wv != 0) {
#pragma unroll
            for (int k = 0; k < PER; ++k) { const int q = tid + k * NT; pq_r[k] = pimg[q < PPN_PATH_POINTS ? q : 0]; }
        }
        if (wv == 0) {
            // wave 0: hull, then straight into the placement loop (same wave: LDS keeps program order)
            hullc[lane][0] = P.hull[((size_t)pj * PPN_MAX_HULL + lane) * 2] - half;
            hullc[lane][1] = P.hull[((size_t)pj * PPN_MAX_HULL + lane) * 2 + 1] - half;
            if (lane == 0) bci[12] = prm.force_compose;                       // "some obstacle may touch the corridor"
        } else {
            // waves 1..3, concurrently with the placement: stage everything that does not depend on it
            const int t3 = tid - 64;
            // K random obstacle candidates (MapGenerate.py:128-136): draws [0,K) rows, [K,2K) columns, [2K,3K) sizes
            if (prm.obst_draws) {
                const double* d = prm.obst_draws + (size_t)m * 3 * K;
                for (int k = t3; k < K; k += NT - 64) {
                    cand[k][0] = d[k] * prm.map_size / prm.map_size * Rd;
                    cand[k][1] = d[K + k] * prm.map_size / prm.map_size * Rd;
                    cand[k][2] = d[2 * K + k] * prm.obstacles_size / prm.map_size * Rd;
                }
            } else {
                // one Philox block yields draws 2b and 2b+1: a lane per block, not per candidate (a third of the rounds)
                for (int bk = t3; 2 * bk < 3 * K; bk += NT - 64) {
                    double u[2];
                    philox_double2(prm.seed, STREAM_OBST, mid, (uint32_t)bk, u[0], u[1]);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int d = 2 * bk + h;
                        if (d < 3 * K) {
                            const int comp = (d >= K ? 1 : 0) + (d >= 2 * K ? 1 : 0);
                            cand[d - comp * K][comp] = u[h] * (comp == 2 ? prm.obstacles_size : prm.map_size) / prm.map_size * Rd;
                        }
                    }
                }
            }
        }
        PPN_STAMP(0);

        // ------------------------------------------------------------------ placement: wave 0, one attempt per lane
        if (wv == 0) {
            const double* fed = prm.place_draws ? prm.place_draws + (size_t)m * 3 : nullptr;
            int attempts = 0, t0 = 0, t1 = 0;
            double angle = 0.0, ca = 1.0, sa = 0.0;
            for (int round = 0;; ++round) {
                const int a = round * 64 + lane;                              // this lane's attempt index
                double u0, u1, u2;
                if (fed) { u0 = fed[0]; u1 = fed[1]; u2 = fed[2]; }
                else {
                    // draws 3a, 3a+1, 3a+2 live in Philox blocks (3a)>>1 and (3a)>>1 + 1
                    const uint32_t d = 3u * (uint32_t)a;
                    double p0, p1, p2, p3;
                    philox_double2(prm.seed, STREAM_PLACE, mid, d >> 1, p0, p1);
                    philox_double2(prm.seed, STREAM_PLACE, mid, (d >> 1) + 1u, p2, p3);
                    if (d & 1u) { u0 = p1; u1 = p2; u2 = p3; } else { u0 = p0; u1 = p1; u2 = p2; }
                }
                const double ang = u0 * 360.0 - 180.0;                        // MapGenerate.py:63
                const int a0 = (int)(u1 * Rd - half);                         // MapGenerate.py:64 (trunc)
                const int a1 = (int)(u2 * Rd - half);
                // boundary_check(-angle, [t1, t0]), Path.py:100-111, MapOffset = R/2
                const double rad = (-ang) / 180.0 * PI;
                double c, s;
                sincos_small(rad, s, c);
                const double o1 = (double)a1, o0 = (double)a0;
                bool out = false;
                for (int v = 0; v < hn; ++v) {
                    double hx, hy;
                    rot2(c, s, hullc[v][0], hullc[v][1], hx, hy);
                    hx = hx + o1 + half;
                    hy = hy + o0 + half;
                    out = out || (hx < 0.0) || (hx >= Rd) || (hy < 0.0) || (hy >= Rd);
                }
                unsigned long long okm = __ballot(!out);
                if (fed) okm &= 1ull;
                int src;
                if (okm) { src = __ffsll((long long)okm) - 1; attempts = round * 64 + src + 1; }
                else if (fed) { src = 0; attempts = 1; flags |= PPN_FLAG_PLACE_CAP; }
                else if ((round + 1) * 64 >= PPN_PLACE_TRY_CAP) { src = 63; attempts = PPN_PLACE_TRY_CAP; flags |= PPN_FLAG_PLACE_CAP; }
                else continue;
                angle = __shfl(ang, src, 64);
                ca = __shfl(c, src, 64);
                sa = __shfl(s, src, 64);
                t0 = __shfl(a0, src, 64);
                t1 = __shfl(a1, src, 64);
                break;
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) { const int q = tid + k * NT; pq_r[k] = pimg[q < PPN_PATH_POINTS ? q : 0]; }
            if (lane == 0) {
                bc[0] = angle; bc[6] = ca; bc[7] = sa;
                if constexpr (PHASE & 2) {
                    double s3, c3;                                            // rotate_nearest(space, -angle): F.rotate's inverse
                    sincos_small((double)(float)angle * PPN_DEG2RAD, s3, c3);  // matrix [c, s, 0; -s, c, 0] of radians(float32(angle)): RandomRotation's draw, MapGenerate.py:103-104
                    bc[8] = c3; bc[9] = s3;
                }
                bci[0] = t0; bci[1] = t1; bci[2] = attempts; bci[3] = (int)flags;
                O.angle[m] = angle;
                O.translation[(size_t)m * 2] = t0; O.translation[(size_t)m * 2 + 1] = t1;
                if (O.records) {
                    double* rec = O.records + (size_t)m * PPN_RECORD_WIDTH;
                    rec[0] = angle; rec[2] = (double)t0; rec[3] = (double)t1;
                }
                O.attempts[m] = attempts;
            }
        }
        __syncthreads();
        PPN_STAMP(1);
        t0 = bci[0]; t1 = bci[1];
        flags = (uint32_t)bci[3];
        const double c = bc[6], s = bc[7];                                    // cos/sin(-angle/180*pi), MapGenerate.py:72
        const double tr0 = (double)t1, tr1 = (double)t0;                      // [translation[1], translation[0]]

        // ------------------------------------------------------------------ labels
#pragma unroll
        for (int kk = 0; kk < PER; ++kk) {                                    // MapGenerate.py:76-80
            const int q = tid + kk * NT;
            if (q >= PPN_PATH_POINTS) break;
            const double2 pq = pq_r[kk];
            double rx, ry;
            rot2(c, s, pq.x - half, pq.y - half, rx, ry);
            rx = rx + half + tr0;
            ry = ry + half + tr1;
            if (q & 1) {
                const int k = q >> 1;
                if ((k & 3) == 0 || k == PPN_PATH_POINTS / 2 - 1) {                 // the filter's coarse points
                    const int ci = (k & 3) == 0 ? (k >> 2) : NCOARSE - 1;
                    poddf[ci] = (float)rx; poddf[128 + ci] = (float)ry;
                }
            }
#ifdef PPN_NT_STORES
            typedef double f64x2_t __attribute__((ext_vector_type(2)));
            if (O.pathpoint) __builtin_nontemporal_store(f64x2_t{rx, ry}, reinterpret_cast<f64x2_t*>(O.pathpoint + ((size_t)m * PPN_PATH_POINTS + q) * 2));
#else
            if (O.pathpoint) *reinterpret_cast<double2*>(O.pathpoint + ((size_t)m * PPN_PATH_POINTS + q) * 2) = make_double2(rx, ry);
#endif
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
            if (O.records) { O.records[(size_t)m * PPN_RECORD_WIDTH + 4 + 2 * tid] = rx; O.records[(size_t)m * PPN_RECORD_WIDTH + 5 + 2 * tid] = ry; }
            if (tid == 0) { bc[2] = rx; bc[3] = ry; }                         // init = segpoint[0]
            if (tid == PPN_SEGS) { bc[4] = rx; bc[5] = ry; }                  // end = segpoint[10]
        }
        __syncthreads();
        PPN_STAMP(2);

        // ------------------------------------------------------------------ clearance filter: one lane per obstacle,
        // minimum over the 500 odd path points (squared distances, one sqrt: sqrt is monotone and correctly
        // rounded, so sqrt(min d2) == min sqrt(d2) bit for bit), ballot compaction in draw order
        {
            const double c_px = prm.clearance / prm.map_size * Rd;            // MapGenerate.py:142
            // Corridor-touch margin.  Every corridor pixel of the map lies within
            //   0.5*c_px (ray reach from the centre line, Path.py:119-134) + max_step (next odd path point)
            //   + 4.3 px (five nearest-neighbour roundings of <= 0.71 px: canvas, point lattice, two rotations,
            //     one fractional translation; + the half-pixel offset between the label and the disc frames)
            // of an odd path point, and a pixel of obstacle k lies >= md_k - r_k - reach - 0.71 from every odd path point
            // (reach: how far the reference's raster geometry — crop offset, stroke — can carry a disc beyond its radius,
            // ppn_device.h raster_reach_max: a constant of the resolution).  So an obstacle with md_k - r_k > touch_margin
            // cannot meet the corridor; if none can, the compose pass below is a no-op and is skipped
            // (tests/test_gpu_edage.py checks this against a forced run).
            const double touch_margin = 0.5 * c_px + P.max_step_px[pj] + PPN_TOUCH_ROUNDINGS + raster_reach_max(R);
            // Two levels.  Coarse: every 4th odd point (plus the last: NCOARSE = 126 points), in float, one lane per
            // obstacle, each wave a quarter of the 63 point pairs (8-byte same-address LDS reads broadcast; a 16-byte one
            // measured ~64 LDS cycles).  An odd point is at most 4 path steps from a coarse one, and the image-frame
            // points sit on the pixel lattice (each rounded by <= 0.5 per axis, Path.py:378-386), so the true minimum
            // distance lies in [dc - 4*max_step - sqrt(2), dc]: that settles every obstacle but the ~3 % whose coarse
            // distance falls inside that band around the threshold.  Those get the exact double minimum over all 500 odd points,
            // a wave per obstacle, and the decision `mn > thr` is then the reference's (MapGenerate.py:139-143).
            const double slack = 4.0 * P.max_step_px[pj] + 1.4143 + 0.01;      // + float error of dc (< 1e-3 px)
            int n_rand = 0;
            typedef float v2f __attribute__((ext_vector_type(2)));
            const float* cpx = poddf;
            const float* cpy = poddf + 128;
            const int q_lo = wv * ((NCOARSE / 2 + NW - 1) / NW);
            const int q_hi = min(q_lo + (NCOARSE / 2 + NW - 1) / NW, NCOARSE / 2);
            for (int k0 = 0; k0 < K; k0 += 64) {                              // K <= 256: at most 4 groups of 64
                const int k = k0 + lane;
                const bool valid = k < K;
                {
                    const float oxf = valid ? (float)cand[k][0] : 0.0f, oyf = valid ? (float)cand[k][1] : 0.0f;
                    const v2f ox2 = {oxf, oxf}, oy2 = {oyf, oyf};
                    v2f m2 = {3.0e38f, 3.0e38f};
#pragma unroll 4
                    for (int q = q_lo; q < q_hi; ++q) {
                        const v2f dx = *reinterpret_cast<const v2f*>(cpx + 2 * q) - ox2;
                        const v2f dy = *reinterpret_cast<const v2f*>(cpy + 2 * q) - oy2;
                        const v2f d2 = __builtin_elementwise_fma(dx, dx, dy * dy);
                        m2 = __builtin_elementwise_min(m2, d2);
                    }
                    fmin_w[wv][lane] = fminf(m2.x, m2.y);
                }
                __syncthreads();
                double rk = 0.0, thr = 0.0, lb = 0.0;
                bool sure_acc = false;
                if (wv == 0) {
                    rk = valid ? cand[k][2] : 0.0;
                    thr = rk + c_px;
                    float mfw = fmin_w[0][lane];
#pragma unroll
                    for (int w = 1; w < NW; ++w) mfw = fminf(mfw, fmin_w[w][lane]);
                    const double dc = (double)sqrtf(mfw);
                    const bool sure_rej = dc + 0.01 <= thr;                   // true minimum <= coarse minimum
                    lb = dc - slack;                                          // true minimum >= lb
                    sure_acc = lb > thr;
                    const unsigned long long amb = __ballot(valid && !sure_rej && !sure_acc);
                    if (lane == 0) { bci[5] = (int)(uint32_t)amb; bci[6] = (int)(uint32_t)(amb >> 32); }
                }
                __syncthreads();
                const unsigned long long amb = ((unsigned long long)(uint32_t)bci[5]) | ((unsigned long long)(uint32_t)bci[6] << 32);
                {   // exact minima of the undecided obstacles, dealt round-robin to the four waves
                    unsigned long long rest = amb;
                    for (int j = 0; rest; ++j) {
                        const int l = __ffsll((long long)rest) - 1;
                        rest &= rest - 1ull;
                        if ((j & (NW - 1)) != wv) continue;
                        const double ox = cand[k0 + l][0], oy = cand[k0 + l][1];
                        double md = 1e300;
                        for (int q = lane; q < PPN_PATH_POINTS / 2; q += 64) {
                            const double2 pq = pimg[2 * q + 1];
                            double ppx, ppy;
                            rot2(c, s, pq.x - half, pq.y - half, ppx, ppy);
                            ppx = ppx + half + tr0;
                            ppy = ppy + half + tr1;
                            const double dx = ppx - ox, dy = ppy - oy;
                            md = fmin(md, dx * dx + dy * dy);
                        }
                        md = sqrt(wave_min(md));
                        if (lane == 0) dref[l] = md;
                    }
                }
                if (amb) __syncthreads();                                     // uniform: every thread read the same bci words
                if (wv == 0) {
                    const bool refined = (amb >> lane) & 1ull;
                    const double mn = refined ? dref[lane] : lb;              // exact, or a lower bound that settles it
                    const bool acc = valid && (refined ? (mn > thr) : sure_acc);
                    const unsigned long long all = __ballot(acc);
                    // conservative for unrefined obstacles (lower bound): the flag may only be set too often
                    if (__ballot(acc && !(mn - rk > touch_margin)) && lane == 0) bci[12] = 1;
                    if (valid) {
                        if (O.accept) O.accept[(size_t)m * K + k] = acc ? 1 : 0;
                        if (acc) {
                            const int pos = n_rand + __popcll(all & ((1ull << lane) - 1ull));
                            obs[pos][0] = cand[k][1]; obs[pos][1] = cand[k][0]; obs[pos][2] = cand[k][2];   // [col,row,r]
                        }
                    }
                    n_rand += __popcll(all);
                }
                if (k0 + 64 < K) __syncthreads();                             // fmin_w / bci[5..6] / dref are reused by the next group
            }
            if (tid == 0) {
                bci[4] = n_rand;
                // pocket obstacles keep >= c_px from the odd path points by construction (Path.py:490-491)
                if (n_pocket > 0 && !(c_px > touch_margin)) bci[12] = 1;
            }
        }
        __syncthreads();
        PPN_STAMP(3);
        const int n_rand = bci[4];
        if (tid < n_pocket) {                                                 // MapGenerate.py:83-89
            double rx, ry;
            const double* o = P.obstacles + ((size_t)pj * PPN_MAX_POCKET + tid) * 3;   // stage A's (row, col, r), L2-resident
            rot2(c, s, o[1] - half, o[0] - half, rx, ry);
            rx = rx + half + tr0;
            ry = ry + half + tr1;
            obs[n_rand + tid][0] = ry; obs[n_rand + tid][1] = rx; obs[n_rand + tid][2] = o[2];
        }
        __syncthreads();
        n_obs = n_rand + n_pocket;
        for (int n = tid; n < n_obs; n += NT) {
            double* o = O.obstacles + ((size_t)m * (K + PPN_MAX_POCKET) + n) * 3;
            o[0] = obs[n][0]; o[1] = obs[n][1]; o[2] = obs[n][2];
        }
        if (tid == 0) {
            O.n_obstacles[(size_t)m * 2] = n_obs;
            O.n_obstacles[(size_t)m * 2 + 1] = n_rand;
            // PPN_FLAG_CORRIDOR_PASS tells the raster kernel that some obstacle may touch the corridor
            O.flags[m] = flags | path_flags | (bci[12] ? PPN_FLAG_CORRIDOR_PASS : 0u);
            if (O.records) O.records[(size_t)m * PPN_RECORD_WIDTH + 1] = (double)(int32_t)(flags | path_flags | (bci[12] ? PPN_FLAG_CORRIDOR_PASS : 0u));
        }
        compose = bci[12] != 0;                                           // same thread wrote it or a barrier lies between
        PPN_STAMP(4);
    } else {
        n_obs = O.n_obstacles[(size_t)m * 2];
        t0 = O.translation[(size_t)m * 2]; t1 = O.translation[(size_t)m * 2 + 1];
        compose = prm.force_compose || (O.flags[m] & PPN_FLAG_CORRIDOR_PASS);
        for (int n = tid; n < n_obs; n += NT) {
            const double* o = O.obstacles + ((size_t)m * (K + PPN_MAX_POCKET) + n) * 3;
            obs[n][0] = o[0]; obs[n][1] = o[1]; obs[n][2] = o[2];
        }
        if (tid == 0) {
            bc[2] = O.segpoint[(size_t)m * 22];      bc[3] = O.segpoint[(size_t)m * 22 + 1];        // init = segpoint[0]
            bc[4] = O.segpoint[(size_t)m * 22 + 20]; bc[5] = O.segpoint[(size_t)m * 22 + 21];       // end  = segpoint[10]
            double s3v, c3v;
            sincos_small((double)(float)O.angle[m] * PPN_DEG2RAD, s3v, c3v);             // rotate_nearest(space, -angle), as in the placement half
            bc[8] = c3v; bc[9] = s3v;
        }
    }

    if constexpr (PHASE & 2) {
        for (int w = tid; w < words; w += NT) occw[w] = 0u;
        int* row_lo = reinterpret_cast<int*>(poddf);                       // the float points are dead after the filter
        int* row_off = row_lo + (K + PPN_MAX_POCKET);                      // [n_obs + 1] exclusive offsets (2*(K+64)+1 ints <= 641)
        __syncthreads();

        // ------------------------------------------------------------------ raster 1: exact row spans -> LDS bit mask.
        // The pixel rule (ppn_device.h raster_ellipse / disc_pred: the pixel centre lies in the ellipse the reference's stroked
        // circle covers in the pixel-centre frame) is monotone in |j + 0.5 - cxp| under IEEE rounding, so the columns of a row
        // form an interval: estimate it with a float sqrt, then settle both ends with the exact double predicate.
        const int wpr = R / 32;
        const double c3 = bc[8], s3 = bc[9];
        // (obstacle, row) pairs are flattened over the whole workgroup: obstacle n owns rows [row_lo[n], row_lo[n] + cnt),
        // an exclusive scan of the counts (wave 0, shuffles) gives each pair an index, a binary search gives it back.
        for (int n = tid; n < n_obs; n += NT) {
            const RasterEllipse e = raster_ellipse(obs[n][0], obs[n][1], obs[n][2], raster_geom(R));
            oell[n][0] = e.cxp; oell[n][1] = e.cyp; oell[n][2] = e.ex; oell[n][3] = e.ey; oell[n][4] = e.rhs;
            const int lo = max((int)floor(e.cyp - e.ey - 0.5), 0), hi = min((int)ceil(e.cyp + e.ey - 0.5), R - 1);   // rows with |i + 0.5 - cyp| <= ey
            row_lo[n] = lo;
            row_off[n + 1] = max(hi - lo + 1, 0);
        }
        __syncthreads();
        if (wv == 0) {
            int carry = 0;
            for (int b0 = 0; b0 < n_obs; b0 += 64) {
                const int n = b0 + lane;
                int v = n < n_obs ? row_off[n + 1] : 0;
    #pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
                if (n < n_obs) row_off[n + 1] = carry + v;
                carry += __shfl(v, 63, 64);
            }
            if (lane == 0) row_off[0] = 0;
        }
        __syncthreads();
        const int n_pairs = row_off[n_obs];
        // each thread takes a contiguous run of pairs: one binary search for the first, then it walks rows / obstacles
        const int run = (n_pairs + NT - 1) / NT;
        const float amb_eps = 2.0e-6f * (float)R;
        int pr = tid * run;
        const int pr_end = min(pr + run, n_pairs);
        int n = 0;
        if (pr < pr_end) {
            int lo_n = 0, hi_n = n_obs - 1;                                // largest n with row_off[n] <= pr
            while (lo_n < hi_n) { const int mid = (lo_n + hi_n + 1) >> 1; if (row_off[mid] <= pr) lo_n = mid; else hi_n = mid - 1; }
            n = lo_n;
        }
        int nxt = pr < pr_end ? row_off[n + 1] : 0;                        // first pair of the next obstacle
        for (; pr < pr_end; ++pr) {
            while (pr >= nxt) { ++n; nxt = row_off[n + 1]; }               // obstacles with an empty row range are skipped
            const int i = row_lo[n] + (pr - row_off[n]);
            const double cxp = oell[n][0], cyp = oell[n][1], ex = oell[n][2], ey = oell[n][3], rhs = oell[n][4];
            const double b = (((double)i + 0.5) - cyp) * ex;
            const double b2 = b * b;
            if (b2 > rhs) continue;                                       // a*a + b2 >= b2 > rhs for every column
            // real-valued ends of the interval in column units: j + 0.5 in [cxp - w, cxp + w], w = sqrt(rhs - b2) / ey
            const float w = __builtin_amdgcn_sqrtf((float)(rhs - b2)) * __builtin_amdgcn_rcpf((float)ey);   // 1-ulp hardware sqrt / rcp: an estimate is all it is
            const float xl = (float)cxp - w - 0.5f, xr = (float)cxp + w - 0.5f;
            int jl = (int)ceilf(xl), jh = (int)floorf(xr);
            // The float estimate is off by < 1e-6 * R columns (conversion of cxp, 1-ulp sqrt and reciprocal, a product, two
            // subtractions of values < 1.01 R); an end within amb_eps = 2e-6 * R of an integer is settled with the exact double
            // predicate, which is monotone in |dx|: the true end is the estimate's nearest integer or its inward neighbour.
            // (A wide window costs: one lane in the slow path holds up its whole wave.)
            if (fabsf(xl - rintf(xl)) < amb_eps) {
                const int j0 = (int)rintf(xl);
                jl = disc_pred(j0, cxp, ey, b2, rhs) ? j0 : j0 + 1;
            }
            if (fabsf(xr - rintf(xr)) < amb_eps) {
                const int j0 = (int)rintf(xr);
                jh = disc_pred(j0, cxp, ey, b2, rhs) ? j0 : j0 - 1;
            }
            jl = max(jl, 0); jh = min(jh, R - 1);
            if (jl > jh) continue;
            uint32_t* row = occw + (size_t)i * wpr;
            for (int ww = jl >> 5; ww <= (jh >> 5); ++ww) {
                const int lo = max(jl - ww * 32, 0), hi = min(jh - ww * 32, 31);
                const uint32_t msk = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
                atomicOr(&row[ww], msk);
            }
        }
        __syncthreads();
        PPN_STAMP(5);

        // ------------------------------------------------------------------ raster 2: the corridor wins over obstacles
        // (MapGenerate.py:111 saturating sum): inverse-map the occupied pixels into the target path's mask,
        // one lane per pixel, 64 consecutive pixels per wave step.  Skipped when no obstacle can touch it.
        if (compose) {
            const TvAxis rcol = tv_axis(c3, s3, 0.0, R), rrow = tv_axis(-s3, c3, 0.0, R);
            const float ctr = 0.5f - 0.5f * (float)R;
            for (int base = wv * 64; base < R * R; base += NT) {
                const int px = base + lane;
                const int i = px / R, j = px - i * R;
                const uint32_t wbits = occw[px >> 5];
                bool clr = false;
                if ((wbits >> (px & 31)) & 1u) {
                    // torchvision's float32 maps (ppn_device.h tv_src): the integer translate (tx, ty) = (translation[0], [1])
                    // is exactly j - tx, i - ty under that rule (checked for every R and shift in the oracle), then the rotation
                    const int i1 = i - t1, j1 = j - t0;
                    if (i1 >= 0 && i1 < R && j1 >= 0 && j1 < R) {
                        const float yo = (float)i1 + ctr, xo = (float)j1 + ctr;
                        const int jj = tv_src(rcol, xo, yo), ii = tv_src(rrow, xo, yo);
                        if (ii >= 0 && ii < R && jj >= 0 && jj < R) {
                            const int bit = ii * R + jj;
                            clr = (P.space_bits[(size_t)pj * words + (bit >> 5)] >> (bit & 31)) & 1u;   // rare pass: straight from L2
                        }
                    }
                }
                const unsigned long long cm = __ballot(clr);
                if (cm) {
                    if (lane == 0 && (uint32_t)cm) occw[px >> 5] = wbits & ~(uint32_t)cm;
                    if (lane == 32 && (uint32_t)(cm >> 32)) occw[px >> 5] = wbits & ~(uint32_t)(cm >> 32);
                }
            }
            __syncthreads();
        }
        PPN_STAMP(6);

        // ------------------------------------------------------------------ raster 3: bits -> bytes, two 16-byte stores
        // per lane per step (8 bits -> 8 bytes through the LDS table; the two 7x7 marker squares touch <= 28 words)
        {
            const int r_init = (int)rint(bc[2]), c_init = (int)rint(bc[3]);   // process_map.py:127-135
            const int r_end = (int)rint(bc[4]), c_end = (int)rint(bc[5]);
            uint8_t* g = O.grid + (size_t)m * R * R;
            // words of the marker rows: [w_i0, w_i1) and [w_e0, w_e1); the rest never needs its row or column
            const int w_i0 = (r_init - 3) * wpr, w_i1 = (r_init + 4) * wpr, w_e0 = (r_end - 3) * wpr, w_e1 = (r_end + 4) * wpr;
            for (int w = tid; w < words; w += NT) {
                const uint32_t occ = occw[w];
                uint64_t q0 = lut[occ & 0xffu], q1 = lut[(occ >> 8) & 0xffu], q2 = lut[(occ >> 16) & 0xffu], q3 = lut[occ >> 24];
                const bool ri = (w >= w_i0) && (w < w_i1), re = (w >= w_e0) && (w < w_e1);
                if (ri || re) {
                    const int i = w / wpr;
                    const int j0 = (w - i * wpr) * 32;
                    uint32_t mark = 0u;
                    if (ri) {
                        const int lo = max(c_init - 3 - j0, 0), hi = min(c_init + 3 - j0, 31);
                        if (lo <= hi) mark |= (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
                    }
                    if (re) {
                        const int lo = max(c_end - 3 - j0, 0), hi = min(c_end + 3 - j0, 31);
                        if (lo <= hi) mark |= (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
                    }
                    if (mark) {                                           // MARK = 0x80 over whatever is there
                        const uint64_t m0 = ~lut[mark & 0xffu], m1 = ~lut[(mark >> 8) & 0xffu], m2 = ~lut[(mark >> 16) & 0xffu],
                                       m3 = ~lut[mark >> 24];             // 0xFF where marked
                        q0 = (q0 & ~m0) | (m0 & 0x8080808080808080ull); q1 = (q1 & ~m1) | (m1 & 0x8080808080808080ull);
                        q2 = (q2 & ~m2) | (m2 & 0x8080808080808080ull); q3 = (q3 & ~m3) | (m3 & 0x8080808080808080ull);
                    }
                }
#ifdef PPN_NT_STORES
                // write-once output: non-temporal stores (A/B build, tools/r02_nt.sh)
                typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                u32x4_t* dst = reinterpret_cast<u32x4_t*>(g + (size_t)w * 32);
                __builtin_nontemporal_store(u32x4_t{(uint32_t)q0, (uint32_t)(q0 >> 32), (uint32_t)q1, (uint32_t)(q1 >> 32)}, dst);
                __builtin_nontemporal_store(u32x4_t{(uint32_t)q2, (uint32_t)(q2 >> 32), (uint32_t)q3, (uint32_t)(q3 >> 32)}, dst + 1);
#else
                uint4* dst = reinterpret_cast<uint4*>(g + (size_t)w * 32);
                dst[0] = make_uint4((uint32_t)q0, (uint32_t)(q0 >> 32), (uint32_t)q1, (uint32_t)(q1 >> 32));
                dst[1] = make_uint4((uint32_t)q2, (uint32_t)(q2 >> 32), (uint32_t)q3, (uint32_t)(q3 >> 32));
#endif
            }
        }
        PPN_STAMP(7);
    }
}

template <int PHASE>
static int launch_phase(const MapsParams& prm, hipStream_t stream) {
    const int R = prm.R, K = prm.K;
    const size_t lds = (size_t)maps_region_bytes(PHASE, R) + ((PHASE & 1) ? (size_t)K * 24 : 0) + (size_t)(K + PPN_MAX_POCKET) * 24 +
                       maps_tab_bytes(K) + ((PHASE & 2) ? (size_t)(K + PPN_MAX_POCKET) * 40 : MAPS_SCRATCH);
    if (hipFuncSetAttribute((const void*)edage_maps_kernel_t<PHASE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PPN_E_HIP;
    hipLaunchKernelGGL(edage_maps_kernel_t<PHASE>, dim3((unsigned)prm.n_maps), dim3(PPN_MAPS_THREADS), lds, stream, prm);
    return hipGetLastError() == hipSuccess ? PPN_OK : PPN_E_HIP;
}

int edage_maps_launch(int phase, const MapsParams& prm, hipStream_t stream) {
    switch (phase) {
        case 1: return launch_phase<1>(prm, stream);
        case 2: return launch_phase<2>(prm, stream);
        case 3: return launch_phase<3>(prm, stream);
    }
    return PPN_E_INVALID;
}

#ifdef PPN_PHASE_TIMING
extern "C" int ppn_debug_phase_cycles(unsigned long long* out_host, int reset) {
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_phase_cycles), sizeof(unsigned long long) * 16) != hipSuccess) return -2;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)) != hipSuccess) return -2;
    }
    return 0;
}
#endif

// Path.boundary_check (Path.py:100-111): one wave per (angle, translation) pair
__global__ __launch_bounds__(NT) void boundary_check_kernel(const double* hull, int hull_n, const double* angle_deg,
                                                            const double* trans_rc, int n, int R, uint8_t* ok,
                                                            double* hull_out) {
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
        if (hull_out) { hull_out[((size_t)i * hull_n + v) * 2] = hx; hull_out[((size_t)i * hull_n + v) * 2 + 1] = hy; }
        out = out || (hx < 0.0) || (hx >= Rd) || (hy < 0.0) || (hy >= Rd);
    }
    const bool good = __ballot(out) == 0ull;
    if (lane == 0) ok[i] = good ? 1 : 0;
}

// generate_gen_path / generate_seg_space (process_map.py:148-191): one workgroup per map. mask_space is the corridor of
// the compose step (same inverse nearest-neighbour map of the target path's Space bits, staged in LDS), written as
// {0,1} bytes with 16-byte stores; mask_path scatters the 200 every-5th label points as 255.
__global__ __launch_bounds__(NT) void label_masks_kernel(ppn_paths_t P, ppn_maps_t M, int placements, int R, int bound,
                                                         uint8_t* mask_path, uint8_t* mask_space) {
    extern __shared__ uint32_t lm_space[];
    const int m = blockIdx.x, tid = threadIdx.x, pj = m / placements;
    const int words = R * R / 32;
    if (mask_space) {
        for (int w = tid; w < words; w += NT) lm_space[w] = P.space_bits[(size_t)pj * words + w];
        __syncthreads();
        const double angle = M.angle[m];
        const int t0 = M.translation[(size_t)m * 2], t1 = M.translation[(size_t)m * 2 + 1];
        double c3, s3;
        sincos_small((double)(float)angle * PPN_DEG2RAD, s3, c3);             // float32: see the placement half
        const TvAxis rcol = tv_axis(c3, s3, 0.0, R), rrow = tv_axis(-s3, c3, 0.0, R);
        const float ctr = 0.5f - 0.5f * (float)R;
        uint8_t* g = mask_space + (size_t)m * R * R;
        const int cpr = R / 16;
        for (int ch = tid; ch < R * cpr; ch += NT) {
            const int i = ch / cpr, j0 = (ch - i * cpr) * 16;
            const int i1 = i - t1;
            uint32_t bits = 0u;
            if (i1 >= 0 && i1 < R) {
                const float yo = (float)i1 + ctr;
                for (int k = 0; k < 16; ++k) {
                    const int j1 = j0 + k - t0;
                    if (j1 < 0 || j1 >= R) continue;
                    const float xo = (float)j1 + ctr;
                    const int jj = tv_src(rcol, xo, yo), ii = tv_src(rrow, xo, yo);
                    if (ii < 0 || ii >= R || jj < 0 || jj >= R) continue;
                    const int bit = ii * R + jj;
                    bits |= ((lm_space[bit >> 5] >> (bit & 31)) & 1u) << k;
                }
            }
            uint32_t w4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t n4 = (bits >> (4 * q)) & 15u;
                w4[q] = (n4 & 1u) | ((n4 & 2u) << 7) | ((n4 & 4u) << 14) | ((n4 & 8u) << 21);
            }
            *reinterpret_cast<uint4*>(g + (size_t)i * R + j0) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
    }
    if (mask_path) {
        uint8_t* g = mask_path + (size_t)m * R * R;
        for (int w = tid; w < R * R / 16; w += NT) reinterpret_cast<uint4*>(g)[w] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        for (int q = tid; q < PPN_PATH_POINTS / 5; q += NT) {
            const double* pt = M.pathpoint + ((size_t)m * PPN_PATH_POINTS + 5 * q) * 2;
            const int r0 = (int)rint(pt[0]), c0 = (int)rint(pt[1]);
            if (r0 > 0 && r0 < bound && c0 > 0 && c0 < bound && r0 < R && c0 < R) g[(size_t)r0 * R + c0] = 255;
        }
    }
}

// generate_map_randomly's accept loop on its own (MapGenerate.py:128-143): one workgroup per map,
// one wave per candidate, shuffle min-reduce over the 500 odd path points
__global__ __launch_bounds__(NT) void obstacle_filter_kernel(const double* pathpoint, const double* draws, int n, int K,
                                                             int R, double map_size, double obstacles_size,
                                                             double clearance, uint8_t* accept, double* obstacles,
                                                             int32_t* counts) {
    __shared__ double podd[PPN_PATH_POINTS / 2][2];
    __shared__ uint8_t acc[256];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double Rd = (double)R;
    for (int q = tid; q < PPN_PATH_POINTS / 2; q += NT) {
        podd[q][0] = pathpoint[((size_t)m * PPN_PATH_POINTS + 2 * q + 1) * 2];
        podd[q][1] = pathpoint[((size_t)m * PPN_PATH_POINTS + 2 * q + 1) * 2 + 1];
    }
    __syncthreads();
    const double* d = draws + (size_t)m * 3 * K;
    const double c_px = clearance / map_size * Rd;
    for (int k = wv; k < K; k += NW) {
        const double ox = d[k] * map_size / map_size * Rd, oy = d[K + k] * map_size / map_size * Rd;
        const double r = d[2 * K + k] * obstacles_size / map_size * Rd;
        double mn = 1e300;
        for (int q = lane; q < PPN_PATH_POINTS / 2; q += 64) {
            const double dx = podd[q][0] - ox, dy = podd[q][1] - oy;
            mn = fmin(mn, dx * dx + dy * dy);
        }
        mn = sqrt(wave_min(mn));
        if (lane == 0) acc[k] = mn > r + c_px ? 1 : 0;
    }
    __syncthreads();
    if (tid == 0) {
        int cnt = 0;
        for (int k = 0; k < K; ++k) {
            if (accept) accept[(size_t)m * K + k] = acc[k];
            if (acc[k]) {
                double* o = obstacles + ((size_t)m * K + cnt) * 3;
                o[0] = d[K + k] * map_size / map_size * Rd;
                o[1] = d[k] * map_size / map_size * Rd;
                o[2] = d[2 * K + k] * obstacles_size / map_size * Rd;
                ++cnt;
            }
        }
        counts[m] = cnt;
    }
}

// add_init_end_single (process_map.py:119-145): 2 x 49 pixels per grid
__global__ __launch_bounds__(128) void paint_markers_kernel(uint8_t* grid, int n, int R, const double* init, const double* end) {
    const int m = blockIdx.x, t = threadIdx.x;
    if (t >= 98) return;
    const double* p = t < 49 ? init + (size_t)m * 2 : end + (size_t)m * 2;
    const int q = t % 49;
    const int i = (int)rint(p[0]) + q / 7 - 3, j = (int)rint(p[1]) + q % 7 - 3;
    if (i >= 0 && i < R && j >= 0 && j < R) grid[(size_t)m * R * R + (size_t)i * R + j] = PPN_GRID_MARK;
}

// explicit obstacle raster rule (stands in for Path.plot_obstacles, Path.py:36-49; ppn_device.h raster_geom / disc_pred)
__global__ __launch_bounds__(NT) void disc_raster_kernel(const double* obstacles, const int32_t* counts, int stride,
                                                         int n_maps, int R, uint8_t* grid) {
    __shared__ double obs[MAX_OBS][5];                                       // the raster ellipse: cxp, cyp, ex, ey, (ex * ey)^2
    const int m = blockIdx.x, tid = threadIdx.x;
    const int n_obs = min(counts[m], MAX_OBS);
    for (int n = tid; n < n_obs; n += NT) {
        const double* o = obstacles + ((size_t)m * stride + n) * 3;
        const RasterEllipse e = raster_ellipse(o[0], o[1], o[2], raster_geom(R));
        obs[n][0] = e.cxp; obs[n][1] = e.cyp; obs[n][2] = e.ex; obs[n][3] = e.ey; obs[n][4] = e.rhs;
    }
    __syncthreads();
    const int cpr = R / 16;
    uint8_t* g = grid + (size_t)m * R * R;
    for (int ch = tid; ch < R * cpr; ch += NT) {
        const int i = ch / cpr, j0 = (ch - i * cpr) * 16;
        uint32_t occ = 0u;
        for (int n = 0; n < n_obs; ++n) {
            const double b = (((double)i + 0.5) - obs[n][1]) * obs[n][2], rhs = obs[n][4], b2 = b * b;
            if (b2 > rhs) continue;
            const double cxp = obs[n][0], ey = obs[n][3];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (disc_pred(j0 + k, cxp, ey, b2, rhs)) occ |= 1u << k;
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
