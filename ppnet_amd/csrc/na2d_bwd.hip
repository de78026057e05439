// na2d_bwd.hip — backward of the 2-D neighbourhood attention (the first brick of the training step,
// reference GenNet/train.py:93-147 / SegNet/mmseg/apis/train.py:67-167; NATTEN's natten2dqkrpb / natten2dav backward kernels behind
// SegNet/nat.py:14,111-120).  Same semantics as the forward kernel (oracle/na_np.py): kernel 7, head dim 32, dilation d, window
// start clamp(u - 3, 0, n - 7) inside the query's dilation group, relative position bias rpb[h][ki + ws - u + 6][kj + ws' - v + 6].
//
// Correctness first (two passes, global memory only, float32 arithmetic):
//   pass 1, one thread per (query, head): recompute the 49 logits and probabilities p, dP_n = dout . v_n,
//           dS_n = p_n (dP_n - sum_m p_m dP_m);  dq = scale * sum_n dS_n k_n;  p and dS are written to [B][heads][H][W][49]
//           workspaces (what NATTEN materialises as `attn` and its gradient);  drpb: 169 bins per head summed in LDS, then
//           one float atomic per bin per workgroup;
//   pass 2, one thread per (key, head): gather over the queries whose window contains the key — a contiguous range per axis,
//           the inverse of the clamped-window map — dk = scale * sum dS q,  dv = sum p dout.  No atomics on dk / dv.
#include <hip/hip_bf16.h>
#include "ppn_device.h"
#include "ppn_kernels.h"

namespace ppn {

namespace {
constexpr int BK = 7, BN = 3, BHD = 32;

template <typename T> __device__ __forceinline__ float ldv(const T* p);
template <> __device__ __forceinline__ float ldv<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldv<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }
template <typename T> __device__ __forceinline__ void stv(T* p, float v);
template <> __device__ __forceinline__ void stv<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stv<__hip_bfloat16>(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }

__device__ __forceinline__ int clampw(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
}  // namespace

template <typename T>
__global__ __launch_bounds__(128) void na2d_bwd_query_kernel(const T* __restrict__ qkv, const float* __restrict__ rpb, const T* __restrict__ dout,
                                                             T* __restrict__ dqkv, float* __restrict__ drpb, float* __restrict__ attn_p,
                                                             float* __restrict__ attn_ds, int B, int H, int W, int heads, int dil, float scale) {
    __shared__ float bins[169];
    const int h = blockIdx.y;
    for (int t = threadIdx.x; t < 169; t += blockDim.x) bins[t] = 0.f;
    __syncthreads();
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * H * W;
    if (idx < total) {
        const int x = (int)(idx % W), y = (int)((idx / W) % H), b = (int)(idx / ((long long)W * H));
        const int gi = y % dil, gj = x % dil, u = y / dil, v = x / dil;
        const int hs = (H - gi + dil - 1) / dil, ws = (W - gj + dil - 1) / dil;
        const int wi = clampw(u - BN, 0, hs - BK), wj = clampw(v - BN, 0, ws - BK);
        const size_t tok = (size_t)3 * heads * BHD;
        const T* qrow = qkv + ((size_t)(b * H + y) * W + x) * tok + (size_t)h * BHD;
        float q[BHD], g[BHD];
#pragma unroll
        for (int c = 0; c < BHD; ++c) { q[c] = ldv<T>(qrow + c) * scale; g[c] = ldv<T>(dout + ((size_t)(b * H + y) * W + x) * ((size_t)heads * BHD) + (size_t)h * BHD + c); }
        float p[BK * BK], dp[BK * BK];
        float mx = -3.0e38f;
        for (int ki = 0; ki < BK; ++ki)
            for (int kj = 0; kj < BK; ++kj) {
                const int yy = gi + (wi + ki) * dil, xx = gj + (wj + kj) * dil;
                const T* krow = qkv + ((size_t)(b * H + yy) * W + xx) * tok + ((size_t)heads + h) * BHD;
                const T* vrow = qkv + ((size_t)(b * H + yy) * W + xx) * tok + ((size_t)2 * heads + h) * BHD;
                float s = 0.f, d = 0.f;
#pragma unroll
                for (int c = 0; c < BHD; ++c) { s = fmaf(q[c], ldv<T>(krow + c), s); d = fmaf(g[c], ldv<T>(vrow + c), d); }
                s += rpb[(size_t)h * 169 + (wi + ki - u + BK - 1) * 13 + (wj + kj - v + BK - 1)];
                p[ki * BK + kj] = s; dp[ki * BK + kj] = d;
                mx = fmaxf(mx, s);
            }
        float sum = 0.f;
        for (int t = 0; t < BK * BK; ++t) { p[t] = expf(p[t] - mx); sum += p[t]; }
        const float inv = 1.0f / sum;
        float dot = 0.f;
        for (int t = 0; t < BK * BK; ++t) { p[t] *= inv; dot = fmaf(p[t], dp[t], dot); }
        float dq[BHD];
#pragma unroll
        for (int c = 0; c < BHD; ++c) dq[c] = 0.f;
        float* prow = attn_p + (((size_t)b * heads + h) * H * W + (size_t)y * W + x) * (BK * BK);
        float* drow = attn_ds + (((size_t)b * heads + h) * H * W + (size_t)y * W + x) * (BK * BK);
        for (int ki = 0; ki < BK; ++ki)
            for (int kj = 0; kj < BK; ++kj) {
                const int t = ki * BK + kj;
                const float ds = p[t] * (dp[t] - dot);
                prow[t] = p[t]; drow[t] = ds;
                atomicAdd(&bins[(wi + ki - u + BK - 1) * 13 + (wj + kj - v + BK - 1)], ds);
                const int yy = gi + (wi + ki) * dil, xx = gj + (wj + kj) * dil;
                const T* krow = qkv + ((size_t)(b * H + yy) * W + xx) * tok + ((size_t)heads + h) * BHD;
#pragma unroll
                for (int c = 0; c < BHD; ++c) dq[c] = fmaf(ds, ldv<T>(krow + c), dq[c]);
            }
        T* dqrow = dqkv + ((size_t)(b * H + y) * W + x) * tok + (size_t)h * BHD;
#pragma unroll
        for (int c = 0; c < BHD; ++c) stv<T>(dqrow + c, dq[c] * scale);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 169; t += blockDim.x)
        if (bins[t] != 0.f) atomicAdd(drpb + (size_t)h * 169 + t, bins[t]);
}

template <typename T>
__global__ __launch_bounds__(128) void na2d_bwd_key_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, T* __restrict__ dqkv,
                                                           const float* __restrict__ attn_p, const float* __restrict__ attn_ds, int B, int H, int W,
                                                           int heads, int dil, float scale) {
    const int h = blockIdx.y;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * H * W) return;
    const int x = (int)(idx % W), y = (int)((idx / W) % H), b = (int)(idx / ((long long)W * H));
    const int gi = y % dil, gj = x % dil, i = y / dil, j = x / dil;              // key (i, j) of its dilation group
    const int hs = (H - gi + dil - 1) / dil, ws = (W - gj + dil - 1) / dil;
    // queries u with clamp(u - 3, 0, hs - 7) <= i <= clamp(u - 3, 0, hs - 7) + 6: a contiguous range
    const int ulo = (i <= BK - 1) ? 0 : i - BN, uhi = (i >= hs - BK) ? hs - 1 : i + BN;
    const int vlo = (j <= BK - 1) ? 0 : j - BN, vhi = (j >= ws - BK) ? ws - 1 : j + BN;
    const size_t tok = (size_t)3 * heads * BHD;
    float dk[BHD], dv[BHD];
#pragma unroll
    for (int c = 0; c < BHD; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
    for (int u = ulo; u <= uhi; ++u) {
        const int wi = clampw(u - BN, 0, hs - BK);
        if (i < wi || i > wi + BK - 1) continue;
        for (int v = vlo; v <= vhi; ++v) {
            const int wj = clampw(v - BN, 0, ws - BK);
            if (j < wj || j > wj + BK - 1) continue;
            const int qy = gi + u * dil, qx = gj + v * dil;
            const size_t a = (((size_t)b * heads + h) * H * W + (size_t)qy * W + qx) * (BK * BK) + (size_t)(i - wi) * BK + (j - wj);
            const float pp = attn_p[a], ds = attn_ds[a];
            const T* qrow = qkv + ((size_t)(b * H + qy) * W + qx) * tok + (size_t)h * BHD;
            const T* grow = dout + ((size_t)(b * H + qy) * W + qx) * ((size_t)heads * BHD) + (size_t)h * BHD;
#pragma unroll
            for (int c = 0; c < BHD; ++c) { dk[c] = fmaf(ds, ldv<T>(qrow + c), dk[c]); dv[c] = fmaf(pp, ldv<T>(grow + c), dv[c]); }
        }
    }
    T* dkrow = dqkv + ((size_t)(b * H + y) * W + x) * tok + ((size_t)heads + h) * BHD;
    T* dvrow = dqkv + ((size_t)(b * H + y) * W + x) * tok + ((size_t)2 * heads + h) * BHD;
#pragma unroll
    for (int c = 0; c < BHD; ++c) { stv<T>(dkrow + c, dk[c] * scale); stv<T>(dvrow + c, dv[c]); }
}

template <typename T>
static int bwd_typed(const void* qkv, const float* rpb, const void* dout, void* dqkv, float* drpb, float* attn_p, float* attn_ds, int B, int H,
                     int W, int heads, int dil, float scale, hipStream_t stream) {
    const long long total = (long long)B * H * W;
    const dim3 grid((unsigned)((total + 127) / 128), heads);
    hipLaunchKernelGGL((na2d_bwd_query_kernel<T>), grid, dim3(128), 0, stream, (const T*)qkv, rpb, (const T*)dout, (T*)dqkv, drpb, attn_p, attn_ds, B, H,
                       W, heads, dil, scale);
    hipLaunchKernelGGL((na2d_bwd_key_kernel<T>), grid, dim3(128), 0, stream, (const T*)qkv, (const T*)dout, (T*)dqkv, attn_p, attn_ds, B, H, W, heads, dil,
                       scale);
    return (int)hipGetLastError();
}

int na2d_bwd_launch(const void* qkv, const float* rpb, const void* dout, void* dqkv, float* drpb, float* attn_p, float* attn_ds, int B, int H, int W,
                    int heads, int dil, float scale, int dtype, hipStream_t stream) {
    return dtype == 0 ? bwd_typed<float>(qkv, rpb, dout, dqkv, drpb, attn_p, attn_ds, B, H, W, heads, dil, scale, stream)
                      : bwd_typed<__hip_bfloat16>(qkv, rpb, dout, dqkv, drpb, attn_p, attn_ds, B, H, W, heads, dil, scale, stream);
}

}  // namespace ppn
