// ppn_kernels.h — kernel parameter blocks and launch prototypes (internal to libppnet_hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/ppnet_hip.h"

#ifndef PPN_PATHS_THREADS
#define PPN_PATHS_THREADS 256
#endif

namespace ppn {

// ---- per-device launcher state.  Function attributes, device-resident constants and CU counts belong to the device that is
// current at the call; a process that drives several GPUs gets one slot per device ordinal (nothing here is shared between
// devices, and nothing is keyed by "first caller").
constexpr int PPN_MAX_DEVICES = 64;
inline int current_device() {
    int d = 0;
    return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < PPN_MAX_DEVICES) ? d : -1;
}
struct DeviceOnce { std::atomic<int> done[PPN_MAX_DEVICES]; };       // static storage: zero-initialised
struct DeviceBuffer { std::atomic<void*> p[PPN_MAX_DEVICES]; };
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): 0 or the HIP error
inline int dynamic_lds_once(DeviceOnce& o, const void* fn, int bytes) {
    const int d = current_device();
    if (d < 0) return (int)hipErrorInvalidDevice;
    if (o.done[d].load(std::memory_order_acquire)) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    o.done[d].store(1, std::memory_order_release);
    return 0;
}
// Region size (16 / 8 / 4 queries a side) of the matrix-core neighbourhood-attention kernels for a dilation group of hq x wq queries:
// the cheapest cover, priced per covered query cell.  Measured on 11 x 11 groups (DiNAT-B's dilation-3 layers at 32 x 32 tokens,
// batch 256: tools/na_timing_512.py): 16 x 16 regions 0.52 ms (47 % of the cells used), 8 x 8 0.97 ms, 4 x 4 1.56 ms (84 % used; the
// v_dot2 kernel 2.23 ms) — per cell 1 : 1.9 : 5.3.  Rounds 2-3 chose by lane utilisation alone and took the 4 x 4 form here.
inline int na_region_size(int hq, int wq) {
    auto cells = [&](int t) { return (double)((hq + t - 1) / t * t) * (double)((wq + t - 1) / t * t); };
    const double c16 = cells(16), c8 = 1.9 * cells(8), c4 = 5.3 * cells(4);
    return (c16 <= c8 && c16 <= c4) ? 16 : (c8 <= c4 ? 8 : 4);
}

// compute units of the current device (cached per device; 0 when the query fails)
inline int device_cu_count() {
    static std::atomic<int> cus[PPN_MAX_DEVICES];
    const int d = current_device();
    if (d < 0) return 0;
    int v = cus[d].load(std::memory_order_relaxed);
    if (!v) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) return 0;
        cus[d].store(v, std::memory_order_relaxed);
    }
    return v;
}

struct PathsParams {
    ppn_paths_t out;
    int n_paths;
    uint64_t first_id;
    int R;
    double map_size, clearance;
    uint64_t seed;
    const double* draws;       // [n][PPN_DRAWS_PER_PATH] or null
    const float* pocket;       // [n][pocket_stride] or null
    int pocket_stride;
    const double* W;           // [4][1000] least-squares operator (device)
    const int8_t* force_straight;  // [n] or null
    const int32_t* hull_start;     // [n] or null: first hull vertex as an index into the canonical cycle (-1 / null = canonical)
};

struct MapsParams {
    ppn_paths_t paths;
    ppn_maps_t out;
    int n_paths, placements, n_maps;
    uint64_t first_map_id;
    int R;
    double map_size, obstacles_size, clearance;
    int K;
    uint64_t seed;
    const double* place_draws; // [n_maps][3] or null
    const double* obst_draws;  // [n_maps][3K] or null
    int force_compose;         // validation: never skip the corridor compose
};

__global__ void edage_paths_kernel(PathsParams prm);
// phase: 1 = placement / labels / filter, 2 = obstacle lists -> grid, 3 = both in one kernel
int edage_maps_launch(int phase, const MapsParams& prm, hipStream_t stream);
__global__ void boundary_check_kernel(const double* hull, int hull_n, const double* angle_deg,
                                      const double* trans_rc, int n, int R, uint8_t* ok, double* hull_out);
__global__ void obstacle_filter_kernel(const double* pathpoint, const double* draws, int n, int K, int R,
                                       double map_size, double obstacles_size, double clearance, uint8_t* accept,
                                       double* obstacles, int32_t* counts);
__global__ void paint_markers_kernel(uint8_t* grid, int n, int R, const double* init, const double* end);
__global__ void disc_raster_kernel(const double* obstacles, const int32_t* counts, int stride,
                                   int n_maps, int R, uint8_t* grid);
__global__ void collision_segments_kernel(const float* s, const float* e, const int32_t* prob, int n_seg,
                                          const float* obs, const int32_t* obs_off, float clearance,
                                          float bound, uint8_t* hit);
__global__ void extract_paths_kernel(const float* heat, int n, int H, int W, const double* init,
                                     const double* end, int max_wp, double* wp, int32_t* wp_n, uint8_t* ok, int vis_dim, int stage_heat);

constexpr int PPN_RESIZE_REP = 16;      // lines / images one thread of resize_pass_kernel walks with its filter weights
__global__ void resize_pass_kernel(const uint8_t* in, int n, int inH, int inW, int outH, int outW, int horizontal,
                                   uint8_t* out);
__global__ void philox_doubles_kernel(uint64_t seed, uint32_t stream_id, uint64_t instance, uint32_t first, int count,
                                      double* out);

int na2d_launch(const void* qkv, const void* pad_kv, const float* rpb, void* out, int B, int H, int W, int Hr, int Wr, int heads, int dil,
                float scale, int dtype, hipStream_t stream);

const void* zero_line();
int na2d_mfma_launch(const void* qkv, const void* pad_kv, const float* rpb, void* out, int B, int H, int W, int Hr, int Wr, int heads, int dil,
                     float scale, hipStream_t stream);
int na2d_halo16_launch(const void* qkv, const void* pad_kv, const float* rpb, void* out, int B, int H, int W, int Hr, int Wr, int heads, int dil,
                       float scale, const void* zero, hipStream_t stream);
long long na2d_bwd_workspace_floats(int B, int H, int W, int heads, int dil);
int na2d_bwd_launch(const void* qkv, const float* rpb, const void* dout, void* dqkv, float* drpb, float* workspace, int B, int H, int W, int heads,
                    int dil, float scale, int dtype, hipStream_t stream);

int norm_launch(const void* x, const void* a, const void* gamma, const void* w, const void* b, void* x_out, void* y_out,
                long long rows, int C, float eps, int dtype, int Hr, int Wr, int Hp, int Wp, const void* xoff, hipStream_t stream);

int resize_concat_launch(const void* const* x, const int* hw, const int* ch, int n, void* out, int B, int dtype, hipStream_t stream);
int adaptive_pools_launch(const void* x, void* const* y, const int* scales, int n, int B, int H, int W, int C, int dtype, hipStream_t stream);
int upsample2x_launch(const void* x, const void* bias, const void* add, void* y, int B, int H, int W, int C, int relu, int dtype, hipStream_t stream);
int conv3x3_c1_launch(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int Cout, float slope, int dtype,
                      hipStream_t stream);
int conv3x3_to1_launch(const void* x, const float* w, float bias, void* y, int B, int H, int W, int Cin, int dtype, hipStream_t stream);
int seg_labels_launch(const void* lo, uint8_t* labels, int B, int h, int w, int Ho, int Wo, int dtype, hipStream_t stream);
int grid_image_launch(const uint8_t* grid, void* img, long long n, const float* mean, const float* stdv, int dtype, hipStream_t stream);
int bias_act_launch(void* x, const void* bias, long long n, int C, float slope, int dtype, hipStream_t stream);

int conv3x3_mfma_launch(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout, int stride,
                        int relu, const float* w2, float* logits, float* partial, hipStream_t stream);
int gennet_enc_conv_launch(const void* x, const void* wk, const float* bias, void* y, int B, int H, int W, float slope, hipStream_t stream);
int gennet_dec_conv_launch(const void* x, const void* wt, const float* bias, void* y, int B, int H, int W, float slope, hipStream_t stream);
int assemble_paths_launch(const double* wp, const int32_t* wp_n, const uint8_t* ok, const double* init, const double* end, double rate, int n, int max_wp,
                          double* full, int32_t* counts, hipStream_t stream);
int plan_collision_launch(const double* full, const int32_t* counts, const void* obstacles, int obs_f64, const int32_t* n_obs, int B, int M, int S,
                          float clearance, float bound, uint8_t* collision, hipStream_t stream);
int na2d_dense7_launch(const void* qkv, const void* pad_kv, const float* rpb, void* out, int B, int H, int W, int Hr, int Wr, int heads, int dil,
                       float scale, hipStream_t stream);
int gennet_dec_final_launch(const void* x, const void* wt, const float* bias, float slope, const float* w1, float bias1, void* y, int B, int H, int W,
                            hipStream_t stream);
int gennet_first_enc_launch(const void* x1, const void* w1, const float* b1, const void* wk2, const float* bias2, void* y, int B, int H, int W, float slope1,
                            float slope2, hipStream_t stream);
int heatmap_u8_launch(const void* y, uint8_t* out, int B, int n, int dtype, hipStream_t stream);
int tokenizer_fused_launch(const uint8_t* grid, const void* lut, const void* w2p, const float* vec, void* out, int B, int H, int W, float eps,
                           hipStream_t stream);
int tokenizer_codes_launch(const uint8_t* grid, const void* lut, void* out, int B, int H, int W, hipStream_t stream);
int nat128_ln_qkv_launch(const void* s, const float* off, const void* lnw, const void* lnb, const void* w, const void* bias, void* qkv, long long tokens,
                         float eps, hipStream_t stream);
int nat128_ln_mlp_launch(void* s, const float* off, const void* lnw, const void* lnb, const void* w1, const void* b1, const void* w2, const float* add,
                         long long tokens, float eps, hipStream_t stream);
int nat128_proj_add_launch(void* s, const void* a, const void* w, long long tokens, hipStream_t stream);
int gennet_trunk_launch(const void* x, void* y, const float* params, int B, int N, int n_blocks, hipStream_t stream);
bool gemm_small_wanted(long long M, int N, int K);
int gemm_small_launch(const void* a, const void* w, const float* bias, void* c, long long M, int N, int K, int epilogue, hipStream_t stream);
bool nat_gemm128_wanted(int N, int K, int mode);
bool nat_gemm128_partials(int C);
int nat_gemm128_launch(const void* a, const void* w, const float* bias, const float* colsum, const float* stats_in, int p_in, float* stats_out,
                       void* c, long long M, int N, int K, int mode, float eps, hipStream_t stream);
int nat_gemm_launch(const void* a, const void* w, const float* bias, const float* colsum, const float* stats_in, int p_in,
                    float* stats_out, void* c, long long M, int N, int K, int mode, float eps, hipStream_t stream);
int row_stats_launch(const void* x, long long rows, int C, float* stats, hipStream_t stream);
// nat_mlp.hip: the fused MLP of a NAT layer (LN -> fc1 -> GELU -> fc2 -> residual, hidden activation never leaves the CU)
bool nat_mlp_supported(long long M, int C, int HID);
int nat_mlp_pack_launch(const void* w1, const void* w2, void* wpk, int C, int HID, hipStream_t stream);
int nat_mlp_launch(void* s, const void* wpk, const float* hb, const float* b2, float* stats_out, long long M, int C, int HID, float eps, hipStream_t stream);
int gemm_acc_stats_launch(const void* a, const void* w, const float* bias, float* stats, void* c, long long M, int N, int K, int p128,
                          int n_cu, hipStream_t stream);
int gemm_ln_launch(const void* a, const void* w, const float* bias, const float* colsum, const float* stats, int parts, void* c, long long M,
                   int N, int K, int gelu, float eps, int n_cu, hipStream_t stream);
int gemm_mfma_launch(const void* a, const void* w, const float* bias, void* c, long long M, int N, int K, int epi, int persistent,
                     hipStream_t stream);

__global__ void label_masks_kernel(ppn_paths_t paths, ppn_maps_t maps, int placements, int R, int bound, uint8_t* mask_path,
                                   uint8_t* mask_space);

}  // namespace ppn
