// capi.hip — the extern "C" boundary of libppnet_hip.so (declared in include/ppnet_hip.h).
// Argument validation, constant tables, kernel launches.  No torch types, no exceptions.
#include <hip/hip_runtime.h>
#include <math.h>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include <cstdlib>
#include "ppn_kernels.h"

namespace {

thread_local int g_last_hip = 0;

inline int hip_fail(hipError_t e) {
    g_last_hip = (int)e;
    return PPN_E_HIP;
}
#define PPN_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(e_); } while (0)

inline bool bad_R(int R) { return R < 32 || R > 512 || (R % 32) != 0; }

// Least-squares operator for np.polyfit(arange(1000)/100, y, 4): W = pinv(V)[0:4], V[i][k] = x_i^(4-k).
// Built once in extended precision (modified Gram-Schmidt with re-orthogonalisation), so the
// device fit is a constant 4 x 1000 matrix-vector product instead of a per-call LAPACK solve.
void build_polyfit_table(double* out) {
    const int N = PPN_PATH_POINTS, M = 5;
    std::vector<long double> q((size_t)M * N);
    long double Rm[5][5] = {};
    for (int i = 0; i < N; ++i) {
        const long double x = (long double)i / 100.0L;
        long double pw = 1.0L;
        for (int k = M - 1; k >= 0; --k) { q[(size_t)k * N + i] = pw; pw *= x; }
    }
    for (int k = 0; k < M; ++k) {
        long double* qk = &q[(size_t)k * N];
        for (int pass = 0; pass < 2; ++pass)
            for (int j = 0; j < k; ++j) {
                const long double* qj = &q[(size_t)j * N];
                long double r = 0.0L;
                for (int i = 0; i < N; ++i) r += qj[i] * qk[i];
                for (int i = 0; i < N; ++i) qk[i] -= r * qj[i];
                Rm[j][k] += r;
            }
        long double nn = 0.0L;
        for (int i = 0; i < N; ++i) nn += qk[i] * qk[i];
        nn = sqrtl(nn);
        Rm[k][k] = nn;
        for (int i = 0; i < N; ++i) qk[i] /= nn;
    }
    for (int i = 0; i < N; ++i) {                 // W[:, i] = R^-1 (Q^T e_i)
        long double w[5];
        for (int k = M - 1; k >= 0; --k) {
            long double v = q[(size_t)k * N + i];
            for (int j = k + 1; j < M; ++j) v -= Rm[k][j] * w[j];
            w[k] = v / Rm[k][k];
        }
        for (int k = 0; k < 4; ++k) out[(size_t)k * N + i] = (double)w[k];
    }
}

std::mutex g_tab_mu;
double* g_tab_dev[64] = {};

int polyfit_table_device(const double** out) {
    int dev = 0;
    PPN_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return PPN_E_INVALID;
    std::lock_guard<std::mutex> lk(g_tab_mu);
    if (!g_tab_dev[dev]) {
        std::vector<double> host(4 * PPN_PATH_POINTS);
        build_polyfit_table(host.data());
        double* d = nullptr;
        PPN_HIP(hipMalloc((void**)&d, host.size() * sizeof(double)));
        hipError_t e = hipMemcpy(d, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d); return hip_fail(e); }
        g_tab_dev[dev] = d;
    }
    *out = g_tab_dev[dev];
    return PPN_OK;
}

}  // namespace

extern "C" {

int ppn_version(void) { return PPN_ABI_VERSION; }

const char* ppn_error_string(int code) {
    switch (code) {
        case PPN_OK: return "ok";
        case PPN_E_INVALID: return "invalid argument";
        case PPN_E_HIP: return "HIP runtime error";
        case PPN_E_UNSUPPORTED: return "unsupported";
        default: return "unknown error";
    }
}

int ppn_last_hip_error(void) { return g_last_hip; }

int ppn_polyfit_table(double* out) {
    if (!out) return PPN_E_INVALID;
    build_polyfit_table(out);
    return PPN_OK;
}

int ppn_edage_paths(int32_t n_paths, uint64_t first_path_id, int32_t R, double map_size, double clearance,
                    uint64_t seed, const double* draws, const float* pocket_draws, int32_t pocket_stride,
                    const ppn_paths_t* out, void* stream) {
    return ppn_edage_paths_ex(n_paths, first_path_id, R, map_size, clearance, seed, draws, pocket_draws, pocket_stride,
                              nullptr, out, stream);
}

int ppn_edage_paths_ex(int32_t n_paths, uint64_t first_path_id, int32_t R, double map_size, double clearance,
                       uint64_t seed, const double* draws, const float* pocket_draws, int32_t pocket_stride,
                       const int8_t* force_straight, const ppn_paths_t* out, void* stream) {
    return ppn_edage_paths_ex2(n_paths, first_path_id, R, map_size, clearance, seed, draws, pocket_draws, pocket_stride,
                               force_straight, nullptr, out, stream);
}

int ppn_edage_paths_ex2(int32_t n_paths, uint64_t first_path_id, int32_t R, double map_size, double clearance,
                        uint64_t seed, const double* draws, const float* pocket_draws, int32_t pocket_stride,
                        const int8_t* force_straight, const int32_t* hull_start, const ppn_paths_t* out, void* stream) {
    if (n_paths < 0 || bad_R(R) || !out || !(map_size > 0.0) || !(clearance > 0.0)) return PPN_E_INVALID;
    if (pocket_draws && pocket_stride <= 0) return PPN_E_INVALID;
    if (n_paths == 0) return PPN_OK;                           // an empty batch has no buffers to validate
    const ppn_paths_t& o = *out;
    if (!o.seg_poly || !o.seg_endpoint || !o.seg_rotation || !o.seg_translation || !o.seg_straight ||
        !o.segpoint_world || !o.pathpoint_world || !o.hull || !o.hull_n || !o.rotation || !o.trans_rc ||
        !o.segpoint_image || !o.pathpoint_image || !o.space_bits || !o.isles || !o.n_isles || !o.obstacles ||
        !o.n_obstacles || !o.length || !o.straight || !o.flags || !o.max_step_px)
        return PPN_E_INVALID;
    if (n_paths == 0) return PPN_OK;
    ppn::PathsParams prm;
    prm.out = o;
    prm.n_paths = n_paths;
    prm.first_id = first_path_id;
    prm.R = R;
    prm.map_size = map_size;
    prm.clearance = clearance;
    prm.seed = seed;
    prm.draws = draws;
    prm.pocket = pocket_draws;
    prm.pocket_stride = pocket_stride;
    prm.force_straight = force_straight;
    prm.hull_start = hull_start;
    int rc = polyfit_table_device(&prm.W);
    if (rc != PPN_OK) return rc;
    const size_t lds = (size_t)PPN_PATH_POINTS * 24 + (size_t)(2 * R) * (2 * R) / 8;      // path points + lattice + canvas bits
    PPN_HIP(hipFuncSetAttribute((const void*)ppn::edage_paths_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ppn::edage_paths_kernel, dim3(n_paths), dim3(PPN_PATHS_THREADS), lds, (hipStream_t)stream, prm);
    PPN_HIP(hipGetLastError());
    return PPN_OK;
}

// phase: 1 = placement half, 2 = raster half, 3 = both in one kernel
static int maps_launch(int phase, const ppn_paths_t* paths, int32_t n_paths, int32_t placements, uint64_t first_map_id,
                       int32_t R, double map_size, double obstacles_size, int32_t K, double clearance, uint64_t seed,
                       const double* place_draws, const double* obst_draws, const ppn_maps_t* out, void* stream) {
    if (!paths || !out || n_paths < 0 || placements < 0 || bad_R(R) || K < 0 || K > 256) return PPN_E_INVALID;
    if ((phase & 1) && !(map_size > 0.0)) return PPN_E_INVALID;
    if ((long long)n_paths * placements == 0) return PPN_OK;   // an empty batch has no buffers to validate
    const ppn_paths_t& p = *paths;
    const ppn_maps_t& o = *out;
    if ((phase & 1) && (!p.hull || !p.hull_n || !p.segpoint_image || !p.pathpoint_image || !p.obstacles ||
                        !p.n_obstacles || !p.flags || !p.max_step_px))
        return PPN_E_INVALID;
    if (!p.space_bits) return PPN_E_INVALID;
    if (!o.grid || !o.angle || !o.translation || !o.attempts || !o.segpoint || !o.obstacles || !o.n_obstacles || !o.flags)
        return PPN_E_INVALID;
    const long long n_maps = (long long)n_paths * placements;
    if (n_maps > 0x7fffffffLL) return PPN_E_INVALID;
    ppn::MapsParams prm;
    prm.paths = p;
    prm.out = o;
    prm.n_paths = n_paths;
    prm.placements = placements;
    prm.n_maps = (int)n_maps;
    prm.first_map_id = first_map_id;
    prm.R = R;
    prm.map_size = map_size;
    prm.obstacles_size = obstacles_size;
    prm.clearance = clearance;
    prm.K = K;
    prm.seed = seed;
    prm.place_draws = place_draws;
    prm.obst_draws = obst_draws;
    {   // validation knob: always run the corridor compose pass (tests compare it with the proven skip)
        const char* f = getenv("PPN_FORCE_COMPOSE");
        prm.force_compose = (f && f[0] == '1') ? 1 : 0;
    }
    {
        const int rc = ppn::edage_maps_launch(phase, prm, (hipStream_t)stream);
        if (rc == PPN_E_HIP) g_last_hip = (int)hipGetLastError();
        if (rc != PPN_OK) return rc;
    }
    return PPN_OK;
}

int ppn_edage_maps(const ppn_paths_t* paths, int32_t n_paths, int32_t placements, uint64_t first_map_id, int32_t R,
                   double map_size, double obstacles_size, int32_t K, double clearance, uint64_t seed,
                   const double* place_draws, const double* obst_draws, const ppn_maps_t* out, void* stream) {
    return maps_launch(3, paths, n_paths, placements, first_map_id, R, map_size, obstacles_size, K, clearance, seed,
                       place_draws, obst_draws, out, stream);
}

int ppn_edage_maps_place(const ppn_paths_t* paths, int32_t n_paths, int32_t placements, uint64_t first_map_id, int32_t R,
                         double map_size, double obstacles_size, int32_t K, double clearance, uint64_t seed,
                         const double* place_draws, const double* obst_draws, const ppn_maps_t* out, void* stream) {
    return maps_launch(1, paths, n_paths, placements, first_map_id, R, map_size, obstacles_size, K, clearance, seed,
                       place_draws, obst_draws, out, stream);
}

int ppn_edage_maps_raster(const ppn_paths_t* paths, int32_t n_paths, int32_t placements, int32_t R, int32_t K,
                          const ppn_maps_t* out, void* stream) {
    return maps_launch(2, paths, n_paths, placements, 0, R, 0.0, 0.0, K, 0.0, 0, nullptr, nullptr, out, stream);
}

int ppn_label_masks(const ppn_paths_t* paths, const ppn_maps_t* maps, int32_t n_paths, int32_t placements, int32_t R,
                    int32_t bound, uint8_t* mask_path, uint8_t* mask_space, void* stream) {
    if (!paths || !maps || n_paths < 0 || placements < 0 || bad_R(R) || bound <= 0) return PPN_E_INVALID;
    if (mask_space && (!paths->space_bits || !maps->angle || !maps->translation)) return PPN_E_INVALID;
    if (mask_path && !maps->pathpoint) return PPN_E_INVALID;
    const long long n = (long long)n_paths * placements;
    if (n == 0 || (!mask_path && !mask_space)) return PPN_OK;
    if (n > 0x7fffffffLL) return PPN_E_INVALID;
    hipLaunchKernelGGL(ppn::label_masks_kernel, dim3((unsigned)n), dim3(256), (size_t)R * R / 8, (hipStream_t)stream, *paths, *maps,
                       placements, R, bound, mask_path, mask_space);
    PPN_HIP(hipGetLastError());
    return PPN_OK;
}

int ppn_boundary_check(const double* hull, int32_t hull_n, const double* angle_deg, const double* translation_rc,
                       int32_t n, int32_t R, uint8_t* ok, void* stream) {
    return ppn_boundary_check_ex(hull, hull_n, angle_deg, translation_rc, n, R, ok, nullptr, stream);
}

int ppn_boundary_check_ex(const double* hull, int32_t hull_n, const double* angle_deg, const double* translation_rc,
                          int32_t n, int32_t R, uint8_t* ok, double* hull_out, void* stream) {
    if (!hull || hull_n <= 0 || !angle_deg || !translation_rc || n < 0 || R <= 0 || !ok) return PPN_E_INVALID;
    if (n == 0) return PPN_OK;
    hipLaunchKernelGGL(ppn::boundary_check_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, hull, hull_n,
                       angle_deg, translation_rc, n, R, ok, hull_out);
    PPN_HIP(hipGetLastError());
    return PPN_OK;
}

int ppn_obstacle_filter(const double* pathpoint, const double* draws, int32_t n, int32_t K, int32_t R, double map_size,
                        double obstacles_size, double clearance, uint8_t* accept, double* obstacles, int32_t* counts,
                        void* stream) {
    if (!pathpoint || !draws || n < 0 || K <= 0 || K > 256 || R <= 0 || !(map_size > 0.0) || !obstacles || !counts)
        return PPN_E_INVALID;
    if (n == 0) return PPN_OK;
    hipLaunchKernelGGL(ppn::obstacle_filter_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, pathpoint, draws, n, K, R,
                       map_size, obstacles_size, clearance, accept, obstacles, counts);
    PPN_HIP(hipGetLastError());
    return PPN_OK;
}

int ppn_paint_markers(uint8_t* grid, int32_t n, int32_t R, const double* init, const double* end, void* stream) {
    if (!grid || n < 0 || R <= 0 || !init || !end) return PPN_E_INVALID;
    if (n == 0) return PPN_OK;
    hipLaunchKernelGGL(ppn::paint_markers_kernel, dim3(n), dim3(128), 0, (hipStream_t)stream, grid, n, R, init, end);
    PPN_HIP(hipGetLastError());
    return PPN_OK;
}

int ppn_disc_raster(const double* obstacles, const int32_t* counts, int32_t stride, int32_t n_maps, int32_t R,
                    uint8_t* grid, void* stream) {
    if (!obstacles || !counts || stride <= 0 || n_maps < 0 || R <= 0 || (R % 16) != 0 || !grid) return PPN_E_INVALID;
    if (n_maps == 0) return PPN_OK;
    hipLaunchKernelGGL(ppn::disc_raster_kernel, dim3(n_maps), dim3(256), 0, (hipStream_t)stream, obstacles, counts,
                       stride, n_maps, R, grid);
    PPN_HIP(hipGetLastError());
    return PPN_OK;
}

int ppn_collision_segments_bound(const float* s, const float* e, const int32_t* prob, int32_t n_seg, const float* obs,
                                 const int32_t* obs_off, float clearance, float bound, uint8_t* hit, void* stream) {
    if (!s || !e || !prob || n_seg < 0 || !obs || !obs_off || !hit || !(bound > 0.0f)) return PPN_E_INVALID;
    if (n_seg == 0) return PPN_OK;
    hipLaunchKernelGGL(ppn::collision_segments_kernel, dim3((n_seg + 255) / 256), dim3(256), 0, (hipStream_t)stream, s,
                       e, prob, n_seg, obs, obs_off, clearance, bound, hit);
    PPN_HIP(hipGetLastError());
    return PPN_OK;
}

int ppn_collision_segments(const float* s, const float* e, const int32_t* prob, int32_t n_seg, const float* obs,
                           const int32_t* obs_off, float clearance, uint8_t* hit, void* stream) {
    return ppn_collision_segments_bound(s, e, prob, n_seg, obs, obs_off, clearance, 224.0f, hit, stream);
}

int ppn_extract_paths(const float* heat, int32_t n, int32_t H, int32_t W, const double* init, const double* end,
                      int32_t max_wp, double* wp, int32_t* wp_n, uint8_t* ok, void* stream) {
    if (!heat || n < 0 || H <= 0 || W <= 0 || !init || !end || max_wp <= 0 || max_wp > PPN_MAX_WAYPOINTS || !wp || !wp_n || !ok)
        return PPN_E_INVALID;
    // the walk keeps the reference's bounds test (process_map.py:318 checks the row against size[0] = width and the column
    // against the height), which only indexes inside the map when it is square
    if (H != W) return PPN_E_UNSUPPORTED;
    if (n == 0) return PPN_OK;
    // visited bitmap over the lattice offsets (-M .. M per axis, M = max(H, W)); none (history scan) if it would not fit in LDS
    int vis_dim = 2 * (H > W ? H : W) + 4;
    size_t lds = (((size_t)vis_dim * vis_dim + 31) / 32) * 4;
    if (lds > 48 * 1024) { vis_dim = 0; lds = 0; }
    // the heat map as 8-bit codes behind the bitmap when both fit beside the 8 KB history (values must be k/255: the
    // caller's ToTensor of an 8-bit image, process_map.py:302); H*W a multiple of 4 for the 16-byte staging loads
    const int stage_heat = ((size_t)H * W % 4 == 0 && lds + (size_t)H * W <= 52 * 1024 && ((uintptr_t)heat % 16) == 0 && ((size_t)H * W * 4) % 16 == 0) ? 1 : 0;
    if (stage_heat) lds += (size_t)H * W;
    hipLaunchKernelGGL(ppn::extract_paths_kernel, dim3(n), dim3(64), lds, (hipStream_t)stream, heat, n, H, W, init, end,
                       max_wp, wp, wp_n, ok, vis_dim, stage_heat);
    PPN_HIP(hipGetLastError());
    return PPN_OK;
}

int ppn_na2d_fwd(const void* qkv, const float* rpb, void* out, int32_t B, int32_t H, int32_t W, int32_t heads,
                 int32_t dilation, float scale, int32_t dtype, void* stream) {
    return ppn_na2d_fwd_padded(qkv, rpb, out, B, H, W, H, W, heads, dilation, scale, dtype, stream);
}

int64_t ppn_na2d_bwd_workspace(int32_t B, int32_t H, int32_t W, int32_t heads, int32_t dilation) {
    if (B <= 0 || heads <= 0 || heads > 65535 || dilation < 1 || H < 7 * dilation || W < 7 * dilation) return -1;
    if ((long long)B * heads * H * W >= (1LL << 40)) return -1;
    return ppn::na2d_bwd_workspace_floats(B, H, W, heads, dilation);
}

int ppn_na2d_bwd(const void* qkv, const float* rpb, const void* dout, void* dqkv, float* drpb, float* workspace, int64_t workspace_floats,
                 int32_t B, int32_t H, int32_t W, int32_t heads, int32_t dilation, float scale, int32_t dtype, void* stream) {
    if (!qkv || !rpb || !dout || !dqkv || !drpb || !workspace || (dtype != 0 && dtype != 1)) return PPN_E_INVALID;
    const int64_t need = ppn_na2d_bwd_workspace(B, H, W, heads, dilation);
    if (need < 0 || workspace_floats < need || (reinterpret_cast<uintptr_t>(workspace) & 15)) return PPN_E_INVALID;
    const int e = ppn::na2d_bwd_launch(qkv, rpb, dout, dqkv, drpb, workspace, B, H, W, heads, dilation, scale, dtype, (hipStream_t)stream);
    if (e == -1) return PPN_E_INVALID;
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

static int na2d_checked(const void* qkv, const void* pad_kv, const float* rpb, void* out, int32_t B, int32_t H, int32_t W, int32_t Hr,
                        int32_t Wr, int32_t heads, int32_t dilation, float scale, int32_t dtype, void* stream) {
    if (!qkv || !rpb || !out || B <= 0 || heads <= 0 || dilation < 1 || (dtype != 0 && dtype != 1)) return PPN_E_INVALID;
    if (Hr <= 0 || Wr <= 0 || Hr > H || Wr > W) return PPN_E_INVALID;
    if (H < 7 * dilation || W < 7 * dilation) return PPN_E_INVALID;      // caller pads, like NATTEN's module
    {
        const long long hs = (H + dilation - 1) / dilation, ws = (W + dilation - 1) / dilation;
        if (((hs + 15) / 16) * ((ws + 15) / 16) * (long long)B * dilation * dilation > 0x7fffffffLL) return PPN_E_INVALID;
    }
    const int e = ppn::na2d_launch(qkv, pad_kv, rpb, out, B, H, W, Hr, Wr, heads, dilation, scale, dtype, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_na2d_fwd_padded(const void* qkv, const float* rpb, void* out, int32_t B, int32_t H, int32_t W, int32_t Hr, int32_t Wr,
                        int32_t heads, int32_t dilation, float scale, int32_t dtype, void* stream) {
    return na2d_checked(qkv, nullptr, rpb, out, B, H, W, Hr, Wr, heads, dilation, scale, dtype, stream);
}

int ppn_na2d_fwd_vpad(const void* qkv, const void* pad_kv, const float* rpb, void* out, int32_t B, int32_t H, int32_t W, int32_t Hr,
                      int32_t Wr, int32_t heads, int32_t dilation, float scale, int32_t dtype, void* stream) {
    if (!pad_kv) return PPN_E_INVALID;
    return na2d_checked(qkv, pad_kv, rpb, out, B, H, W, Hr, Wr, heads, dilation, scale, dtype, stream);
}

int ppn_residual_layernorm(const void* x, const void* a, const void* gamma, const void* w, const void* b, void* x_out,
                           void* y_out, int64_t rows, int32_t C, float eps, int32_t dtype, void* stream) {
    return ppn_residual_layernorm_padded(x, a, gamma, w, b, x_out, y_out, rows, C, eps, dtype, 0, 0, 0, 0, stream);
}

int ppn_residual_layernorm_padded(const void* x, const void* a, const void* gamma, const void* w, const void* b, void* x_out,
                                  void* y_out, int64_t rows, int32_t C, float eps, int32_t dtype, int32_t Hr, int32_t Wr,
                                  int32_t Hp, int32_t Wp, void* stream) {
    if (Hp != 0 && (Hr <= 0 || Wr <= 0 || Hp < Hr || Wp < Wr || rows % ((int64_t)Hr * Wr) != 0)) return PPN_E_INVALID;
    if (!x || rows < 0 || C <= 0 || (C % 8) != 0 || (dtype != 0 && dtype != 1)) return PPN_E_INVALID;
    if (a && !x_out) return PPN_E_INVALID;
    if (!a && !y_out) return PPN_E_INVALID;
    if (y_out && (!w || !b)) return PPN_E_INVALID;
    if (rows == 0) return PPN_OK;
    const int e = ppn::norm_launch(x, a, gamma, w, b, x_out, y_out, rows, C, eps, dtype, Hr, Wr, Hp, Wp, nullptr, (hipStream_t)stream);
    if (e == -1) return PPN_E_UNSUPPORTED;
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_layernorm_offset(const void* x, const float* xoff, const void* w, const void* b, void* y_out, int64_t rows, int32_t C, float eps,
                         int32_t dtype, void* stream) {
    if (!x || !y_out || !w || !b || rows < 0 || C <= 0 || (C % 8) != 0 || (dtype != 0 && dtype != 1)) return PPN_E_INVALID;
    if (rows == 0) return PPN_OK;
    const int e = ppn::norm_launch(x, nullptr, nullptr, w, b, nullptr, y_out, rows, C, eps, dtype, 0, 0, 0, 0, xoff, (hipStream_t)stream);
    if (e == -1) return PPN_E_UNSUPPORTED;
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_upsample2x_nhwc(const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t relu, int32_t dtype,
                        void* stream) {
    return ppn_upsample2x_nhwc_bias(x, nullptr, y, B, H, W, C, relu, dtype, stream);
}

int ppn_upsample2x_nhwc_bias(const void* x, const void* bias, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t relu,
                             int32_t dtype, void* stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) != 0 || (dtype != 0 && dtype != 1)) return PPN_E_INVALID;
    const int e = ppn::upsample2x_launch(x, bias, nullptr, y, B, H, W, C, relu, dtype, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_upsample2x_add_nhwc(const void* x, const void* add, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
    if (!x || !add || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) != 0 || (dtype != 0 && dtype != 1)) return PPN_E_INVALID;
    const int e = ppn::upsample2x_launch(x, nullptr, add, y, B, H, W, C, 0, dtype, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_resize_concat4_nhwc(const void* x0, const void* x1, const void* x2, const void* x3, const int32_t* hw, void* out, int32_t B, int32_t C,
                            int32_t dtype, void* stream) {
    if (!x0 || !x1 || !x2 || !x3 || !hw || !out || B <= 0 || C <= 0 || (C % 8) != 0 || (dtype != 0 && dtype != 1)) return PPN_E_INVALID;
    const void* x[4] = {x0, x1, x2, x3};
    const int32_t ch[4] = {C, C, C, C};
    return ppn_resize_concat_nhwc(x, hw, ch, 4, out, B, dtype, stream);
}

int ppn_resize_concat_nhwc(const void* const* x, const int32_t* hw, const int32_t* channels, int32_t n, void* out, int32_t B, int32_t dtype, void* stream) {
    if (!x || !hw || !channels || !out || n <= 0 || n > 8 || B <= 0 || (dtype != 0 && dtype != 1)) return PPN_E_INVALID;
    int h[16], ch[8];
    for (int l = 0; l < n; ++l) {
        if (!x[l] || hw[2 * l] <= 0 || hw[2 * l + 1] <= 0 || channels[l] <= 0 || (channels[l] % 8) != 0) return PPN_E_INVALID;
        h[2 * l] = hw[2 * l]; h[2 * l + 1] = hw[2 * l + 1]; ch[l] = channels[l];
    }
    const int e = ppn::resize_concat_launch(x, h, ch, n, out, B, dtype, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_adaptive_pools_nhwc(const void* x, void* const* y, const int32_t* scales, int32_t n, int32_t B, int32_t H, int32_t W, int32_t C, int32_t dtype,
                            void* stream) {
    if (!x || !y || !scales || n <= 0 || n > 4 || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) != 0 || (dtype != 0 && dtype != 1)) return PPN_E_INVALID;
    int sc[4];
    for (int k = 0; k < n; ++k) { if (!y[k] || scales[k] <= 0) return PPN_E_INVALID; sc[k] = scales[k]; }
    const int e = ppn::adaptive_pools_launch(x, y, sc, n, B, H, W, C, dtype, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_grid_to_image(const uint8_t* grid, void* img, int64_t n_pixels, const float* mean3, const float* std3, int32_t dtype, void* stream) {
    if (!grid || !img || !mean3 || !std3 || n_pixels < 0 || (n_pixels % 8) != 0 || (dtype != 0 && dtype != 1)) return PPN_E_INVALID;
    if (n_pixels == 0) return PPN_OK;
    const int e = ppn::grid_image_launch(grid, img, n_pixels, mean3, std3, dtype, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_seg_labels_2class(const void* logits, uint8_t* labels, int32_t B, int32_t h, int32_t w, int32_t Ho, int32_t Wo, int32_t dtype,
                          void* stream) {
    if (!logits || !labels || B <= 0 || h <= 0 || w <= 0 || Ho <= 0 || Wo <= 0 || (dtype != 0 && dtype != 1)) return PPN_E_INVALID;
    const int e = ppn::seg_labels_launch(logits, labels, B, h, w, Ho, Wo, dtype, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_bias_act_nhwc(void* x, const void* bias, int64_t n, int32_t C, float negative_slope, int32_t dtype, void* stream) {
    if (!x || !bias || n < 0 || C <= 0 || (C % 8) != 0 || (n % C) != 0 || (dtype != 0 && dtype != 1)) return PPN_E_INVALID;
    if (n == 0) return PPN_OK;
    const int e = ppn::bias_act_launch(x, bias, n, C, negative_slope, dtype, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_conv3x3_c1_nhwc(const void* x, const float* w, const float* bias, void* y, int32_t B, int32_t H, int32_t W, int32_t Cout,
                        float negative_slope, int32_t dtype, void* stream) {
    if (!x || !w || !bias || !y || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout > 32 || (Cout % 8) != 0 || (dtype != 0 && dtype != 1))
        return PPN_E_INVALID;
    const int e = ppn::conv3x3_c1_launch(x, w, bias, y, B, H, W, Cout, negative_slope, dtype, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_conv3x3_to1_nhwc(const void* x, const float* w, float bias, void* y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t dtype,
                         void* stream) {
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin > 32 || (Cin % 8) != 0 || (dtype != 0 && dtype != 1))
        return PPN_E_INVALID;
    const int e = ppn::conv3x3_to1_launch(x, w, bias, y, B, H, W, Cin, dtype, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_conv3x3_mfma_bf16(const void* x, const void* w, const float* bias, void* y, int32_t B, int32_t H, int32_t W, int32_t Cin,
                          int32_t Cout, int32_t stride, int32_t relu, void* stream) {
    if (!x || !w || !bias || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin % 64) != 0 || Cout <= 0 || (Cout % 8) != 0 ||
        (stride != 1 && stride != 2))
        return PPN_E_INVALID;
    // 32-bit byte offsets inside the kernel: the image and the weights must stay below 4 GiB
    if ((long long)B * H * W * Cin * 2 >= (1LL << 32) || (long long)Cout * 9 * Cin * 2 >= (1LL << 32) || (long long)B * H * W >= (1LL << 31))
        return PPN_E_UNSUPPORTED;
    const int e = ppn::conv3x3_mfma_launch(x, w, bias, y, B, H, W, Cin, Cout, stride, relu, nullptr, nullptr, nullptr, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

// workspace slots: one per 256-column block, none when there is a single block (the kernel then adds straight onto the logits)
int32_t ppn_conv3x3_relu_classify2_slots(int32_t Cout) { return Cout > 0 ? (Cout > 256 ? (Cout + 255) / 256 : 0) : -1; }

int ppn_conv3x3_relu_classify2_bf16(const void* x, const void* w, const float* bias, const float* w2, float* logits, float* partial, int32_t B,
                                    int32_t H, int32_t W, int32_t Cin, int32_t Cout, void* stream) {
    if (!x || !w || !bias || !w2 || !logits || (!partial && Cout > 256) || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin % 64) != 0 || Cout <= 0 || (Cout % 8) != 0)
        return PPN_E_INVALID;
    if ((long long)B * H * W * Cin * 2 >= (1LL << 32) || (long long)Cout * 9 * Cin * 2 >= (1LL << 32) || (long long)B * H * W >= (1LL << 31))
        return PPN_E_UNSUPPORTED;
    const int e = ppn::conv3x3_mfma_launch(x, w, bias, nullptr, B, H, W, Cin, Cout, 1, 1, w2, logits, partial, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_gennet_conv_s2_bf16(const void* x, const void* w, const float* bias, void* y, int32_t B, int32_t H, int32_t W, float negative_slope,
                            int32_t transposed, void* stream) {
    if (!x || !w || !bias || !y || B <= 0 || H <= 0 || W <= 0 || (!transposed && ((H | W) & 1))) return PPN_E_INVALID;
    if ((long long)B * H * W * 4 >= (1LL << 31)) return PPN_E_UNSUPPORTED;          // 32-bit pixel indices inside
    const int e = transposed ? ppn::gennet_dec_conv_launch(x, w, bias, y, B, H, W, negative_slope, (hipStream_t)stream)
                             : ppn::gennet_enc_conv_launch(x, w, bias, y, B, H, W, negative_slope, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_gennet_first_enc_bf16(const void* x1, const void* w1, const float* b1, const void* wk2, const float* bias2, void* y, int32_t B, int32_t H, int32_t W,
                              float slope1, float slope2, void* stream) {
    if (!x1 || !w1 || !b1 || !wk2 || !bias2 || !y || B <= 0 || H <= 0 || W <= 0 || ((H | W) & 1)) return PPN_E_INVALID;
    const int e = ppn::gennet_first_enc_launch(x1, w1, b1, wk2, bias2, y, B, H, W, slope1, slope2, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_gennet_dec_final_bf16(const void* x, const void* w, const float* bias, float negative_slope, const float* w_final, float bias_final, void* y,
                              int32_t B, int32_t H, int32_t W, void* stream) {
    if (!x || !w || !bias || !w_final || !y || B <= 0 || H <= 0 || W <= 0) return PPN_E_INVALID;
    if ((long long)B * H * W * 4 * 24 >= (1LL << 32)) return PPN_E_UNSUPPORTED;     // 32-bit element offsets inside one image batch
    const int e = ppn::gennet_dec_final_launch(x, w, bias, negative_slope, w_final, bias_final, y, B, H, W, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_gennet_trunk_bf16(const void* x, void* y, const float* params, int32_t B, int32_t N, int32_t n_blocks, void* stream) {
    if (!x || !y || !params || B <= 0 || N <= 0 || N > 1024 || (N % 8) != 0 || n_blocks <= 0) return PPN_E_INVALID;
    const int e = ppn::gennet_trunk_launch(x, y, params, B, N, n_blocks, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_assemble_paths(const double* wp, const int32_t* wp_n, const uint8_t* ok, const double* init, const double* end, double rate, int32_t n,
                       int32_t max_wp, double* full, int32_t* counts, void* stream) {
    if (!wp || !wp_n || !ok || !init || !end || !full || !counts || n <= 0 || max_wp <= 0) return PPN_E_INVALID;
    const int e = ppn::assemble_paths_launch(wp, wp_n, ok, init, end, rate, n, max_wp, full, counts, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_plan_collision(const double* waypoints, const int32_t* counts, const void* obstacles, int32_t obstacles_f64, const int32_t* n_obstacles,
                       int32_t B, int32_t M, int32_t S, float clearance, float bound, uint8_t* collision, void* stream) {
    if (!waypoints || !counts || !n_obstacles || !collision || B <= 0 || M < 2 || S < 0 || (S > 0 && !obstacles)) return PPN_E_INVALID;
    const int e = ppn::plan_collision_launch(waypoints, counts, obstacles, obstacles_f64, n_obstacles, B, M, S, clearance, bound, collision,
                                             (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_heatmap_u8(const void* y, uint8_t* out, int32_t B, int32_t n, int32_t dtype, void* stream) {
    if (!y || !out || B <= 0 || n <= 0 || (dtype != 0 && dtype != 1)) return PPN_E_INVALID;
    const int e = ppn::heatmap_u8_launch(y, out, B, n, dtype, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_tokenizer_conv1_codes_bf16(const uint8_t* grid, const void* lut, void* out, int32_t B, int32_t H, int32_t W, void* stream) {
    if (!grid || !lut || !out || B <= 0 || H <= 0 || W <= 0) return PPN_E_INVALID;
    if ((H & 1) || (W % 32)) return PPN_E_UNSUPPORTED;
    const int e = ppn::tokenizer_codes_launch(grid, lut, out, B, H, W, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_tokenizer_codes_bf16(const uint8_t* grid, const void* lut, const void* w2p, const float* vec, void* tokens, int32_t B, int32_t H, int32_t W,
                             float eps, void* stream) {
    if (!grid || !lut || !w2p || !vec || !tokens || B <= 0 || H <= 0 || W <= 0) return PPN_E_INVALID;
    if ((H % 4) || (W % 64)) return PPN_E_UNSUPPORTED;
    const int e = ppn::tokenizer_fused_launch(grid, lut, w2p, vec, tokens, B, H, W, eps, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_nat128_ln_qkv_bf16(const void* s, const float* offset, const void* ln_w, const void* ln_b, const void* w, const void* bias, void* qkv,
                           int64_t tokens, float eps, void* stream) {
    if (!s || !ln_w || !ln_b || !w || !qkv || tokens <= 0) return PPN_E_INVALID;
    if (tokens % 16) return PPN_E_UNSUPPORTED;
    const int e = ppn::nat128_ln_qkv_launch(s, offset, ln_w, ln_b, w, bias, qkv, tokens, eps, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_nat128_ln_mlp_add_bf16(void* s, const float* offset, const void* ln_w, const void* ln_b, const void* w1, const void* b1, const void* w2,
                               const float* final_add, int64_t tokens, float eps, void* stream) {
    if (!s || !ln_w || !ln_b || !w1 || !b1 || !w2 || tokens <= 0) return PPN_E_INVALID;
    if (tokens % 16) return PPN_E_UNSUPPORTED;
    const int e = ppn::nat128_ln_mlp_launch(s, offset, ln_w, ln_b, w1, b1, w2, final_add, tokens, eps, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_nat128_ln_mlp_bf16(void* s, const float* offset, const void* ln_w, const void* ln_b, const void* w1, const void* b1, const void* w2,
                           int64_t tokens, float eps, void* stream) {
    return ppn_nat128_ln_mlp_add_bf16(s, offset, ln_w, ln_b, w1, b1, w2, nullptr, tokens, eps, stream);
}

int ppn_nat128_proj_add_bf16(void* s, const void* a, const void* w, int64_t tokens, void* stream) {
    if (!s || !a || !w || tokens <= 0 || (tokens % 16) != 0) return PPN_E_INVALID;
    const int e = ppn::nat128_proj_add_launch(s, a, w, tokens, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_gemm_bf16(const void* a, const void* w, const float* bias, void* c, int64_t M, int32_t N, int32_t K, int32_t epilogue,
                  int32_t persistent_blocks, void* stream) {
    if (!a || !w || !c || M <= 0 || M >= (1LL << 31) || N <= 0 || (N % 8) != 0 || K < 128 || (K % 64) != 0 || epilogue < 0 || epilogue > 3 ||
        (epilogue != 2 && !bias) || persistent_blocks < 0 || (persistent_blocks % 8) != 0)
        return PPN_E_INVALID;
    if (M * K * 2 >= (1LL << 32) || (long long)N * K * 2 >= (1LL << 32)) return PPN_E_UNSUPPORTED;
    // few rows (small batches): the wave-per-block kernel of gemm_small.hip instead of a handful of 256 x 256 tiles
    static const bool no_small = getenv("PPNET_NO_SMALL_GEMM") != nullptr;       // A/B
    const int e = (!no_small && ppn::gemm_small_wanted(M, N, K))
                      ? ppn::gemm_small_launch(a, w, bias, c, M, N, K, epilogue, (hipStream_t)stream)
                      : ppn::gemm_mfma_launch(a, w, bias, c, M, N, K, epilogue, persistent_blocks, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_nat_gemm_bf16(const void* a, const void* w, const float* bias, const float* colsum, const float* stats_in, int32_t partials_in,
                      float* stats_out, void* c, int64_t M, int32_t N, int32_t K, int32_t mode, float eps, void* stream) {
    if (!a || !w || !c || !bias || M <= 0 || M >= (1LL << 31) || (M % 256) != 0 || N <= 0 || (N % 256) != 0 || K < 64 || (K % 64) != 0 ||
        mode < 0 || mode > 2)
        return PPN_E_INVALID;
    if (mode != 2 && (!colsum || !stats_in || partials_in < 1 || partials_in > 4 || K / 64 < 3)) return PPN_E_INVALID;
    if (mode == 2 && !stats_out) return PPN_E_INVALID;
    if ((long long)N * 2 * 8 >= (1LL << 31) || (long long)K * 2 * 8 >= (1LL << 31)) return PPN_E_UNSUPPORTED;     // per-lane 32-bit offsets
    // mode 2 (round 5): the 256 x 256 core of mfma_gemm.h with the old c read in its epilogue — faster than both round-4 forms on all six
    // shapes (profiles/r05_natgemm_timing.txt); PPNET_NAT_ACC=old selects those (A/B)
    static const bool acc_old = [] { const char* v = getenv("PPNET_NAT_ACC"); return v && v[0] == 'o'; }();
    if (mode == 2 && !acc_old && K >= 128) {
        const int n_cu = ppn::device_cu_count();
        if (!n_cu) return hip_fail(hipErrorInvalidDevice);
        const int e2 = ppn::gemm_acc_stats_launch(a, w, bias, stats_out, c, M, N, K, ppn::nat_gemm128_partials(N) ? 1 : 0, n_cu, (hipStream_t)stream);
        if (e2 != 0) return hip_fail((hipError_t)e2);
        return PPN_OK;
    }
    static const bool ln_old = [] { const char* v = getenv("PPNET_NAT_LN"); return v && v[0] == 'o'; }();
    if (mode != 2 && !ln_old && K >= 128) {
        const int n_cu = ppn::device_cu_count();
        if (!n_cu) return hip_fail(hipErrorInvalidDevice);
        const int e2 = ppn::gemm_ln_launch(a, w, bias, colsum, stats_in, partials_in, c, M, N, K, mode == 1, eps, n_cu, (hipStream_t)stream);
        if (e2 != 0) return hip_fail((hipError_t)e2);
        return PPN_OK;
    }
    // the HBM-bound levels (stream width <= 512) run on the small-tile kernel of nat_gemm128.hip, the rest on nat_gemm.hip's persistent one
    const int e = ppn::nat_gemm128_wanted(N, K, mode)
                      ? ppn::nat_gemm128_launch(a, w, bias, colsum, stats_in, partials_in, stats_out, c, M, N, K, mode, eps, (hipStream_t)stream)
                      : ppn::nat_gemm_launch(a, w, bias, colsum, stats_in, partials_in, stats_out, c, M, N, K, mode, eps, (hipStream_t)stream);
    if (e == -2) return hip_fail(hipErrorInvalidDevice);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int32_t ppn_nat_gemm_partials(int32_t C) {
    if (C <= 0 || (C % 256) != 0) return -1;
    return ppn::nat_gemm128_partials(C) ? C / 128 : C / 256;
}

int32_t ppn_nat_mlp_supported(int64_t M, int32_t C, int32_t HID) { return ppn::nat_mlp_supported(M, C, HID) ? 1 : 0; }

int ppn_nat_mlp_pack_bf16(const void* w1, const void* w2, void* wpk, int32_t C, int32_t HID, void* stream) {
    if (!w1 || !w2 || !wpk || HID <= 0 || (HID % 32) != 0) return PPN_E_INVALID;
    const int e = ppn::nat_mlp_pack_launch(w1, w2, wpk, C, HID, (hipStream_t)stream);
    if (e == -1) return PPN_E_UNSUPPORTED;
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_nat_mlp_bf16(void* s, const void* wpk, const float* hb, const float* b2, float* stats_out, int64_t M, int32_t C, int32_t HID,
                     float eps, void* stream) {
    if (!s || !wpk || !hb || !b2 || M <= 0 || M >= (1LL << 31)) return PPN_E_INVALID;
    if (!ppn::nat_mlp_supported(M, C, HID)) return PPN_E_UNSUPPORTED;
    const int e = ppn::nat_mlp_launch(s, wpk, hb, b2, stats_out, M, C, HID, eps, (hipStream_t)stream);
    if (e == -1) return PPN_E_UNSUPPORTED;
    if (e == -2) return hip_fail(hipErrorInvalidDevice);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_row_stats_bf16(const void* x, int64_t rows, int32_t C, float* stats, void* stream) {
    if (!x || !stats || rows < 0 || C < 64 || C > 1024 || (C % 8) != 0) return PPN_E_INVALID;
    if (rows == 0) return PPN_OK;
    const int e = ppn::row_stats_launch(x, rows, C, stats, (hipStream_t)stream);
    if (e != 0) return hip_fail((hipError_t)e);
    return PPN_OK;
}

int ppn_resize_bilinear_u8(const uint8_t* in, int32_t n, int32_t H, int32_t W, int32_t outH, int32_t outW, uint8_t* tmp,
                           uint8_t* out, void* stream) {
    if (!in || n < 0 || H <= 0 || W <= 0 || outH <= 0 || outW <= 0 || !tmp || !out) return PPN_E_INVALID;
    if (n == 0) return PPN_OK;
    // a thread computes its output coordinate's filter weights once and walks PPN_RESIZE_REP lines (pass 1) / images (pass 2)
    const long long g1 = ((long long)n * H + ppn::PPN_RESIZE_REP - 1) / ppn::PPN_RESIZE_REP;
    const long long g2 = (((long long)n + ppn::PPN_RESIZE_REP - 1) / ppn::PPN_RESIZE_REP) * outH;
    if (g1 >= (1LL << 31) || g2 >= (1LL << 31) || (outW + 255) / 256 > 65535) return PPN_E_UNSUPPORTED;
    hipLaunchKernelGGL(ppn::resize_pass_kernel, dim3((unsigned)g1, (unsigned)((outW + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, n, H, W,
                       H, outW, 1, tmp);
    hipLaunchKernelGGL(ppn::resize_pass_kernel, dim3((unsigned)g2, (unsigned)((outW + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)tmp, n, H, outW, outH, outW, 0, out);
    PPN_HIP(hipGetLastError());
    return PPN_OK;
}

int ppn_philox_doubles(uint64_t seed, uint32_t stream_id, uint64_t instance, uint32_t first, int32_t count, double* out,
                       void* stream) {
    if (count < 0 || !out) return PPN_E_INVALID;
    if (count == 0) return PPN_OK;
    hipLaunchKernelGGL(ppn::philox_doubles_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed,
                       stream_id, instance, first, count, out);
    PPN_HIP(hipGetLastError());
    return PPN_OK;
}

}  // extern "C"
