// fused_norm.hip — HBM-bound fusions around the NAT layer's dense ops on gfx950 (reference SegNet/nat.py:140-153):
//
//   layernorm_kernel      y = LN(x)                                   (norm1 of the first layer of a level, output norms)
//   residual_ln_kernel    x' = x + gamma * a ;  y = LN(x')            (residual + LayerScale + the NEXT sub-layer's norm)
//
// The reference runs these as separate torch ops (mul, add, layer_norm = 7 tensor passes per sub-layer); fused they are
// 4 (read x, read a, write x', write y).  A row of C channels (128..1024, every NAT/DiNAT level) is owned by C/8 lanes
// (at most one wave), 8 elements per lane per pass = one 16-byte access for bf16; statistics in float32.
#include <hip/hip_bf16.h>
#include "ppn_device.h"
#include "ppn_kernels.h"

namespace ppn {

namespace {

template <typename T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct Vec8<__hip_bfloat16> {
    static __device__ __forceinline__ void load(const __hip_bfloat16* p, float (&v)[8]) {
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
        v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) { return pack_bf16x2(lo, hi); }
    static __device__ __forceinline__ void store(__hip_bfloat16* p, const float (&v)[8]) {
        *reinterpret_cast<uint4*>(p) = make_uint4(pack(v[0], v[1]), pack(v[2], v[3]), pack(v[4], v[5]), pack(v[6], v[7]));
    }
};

// sum over the `lpr` lanes that share a row (lpr a power of two <= 64, groups aligned)
__device__ __forceinline__ float group_sum(float v, int lpr) {
    for (int o = lpr >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace

constexpr int NORM_ROW_ITERS = 8;     // row groups per wave: the per-channel vectors (w, b, offset: 5x the bytes of a bf16 row piece) are loaded once

// PASSES = C / (8 * lpr): 1 for C <= 512, 2 for C = 1024.  RESID: fuse x' = x + gamma * a (gamma may be null = 1).
template <typename T, int PASSES, bool RESID>
__global__ __launch_bounds__(256) void norm_kernel(const T* __restrict__ x, const T* __restrict__ a, const T* __restrict__ gamma,
                                                   const T* __restrict__ w, const T* __restrict__ b, T* __restrict__ x_out,
                                                   T* __restrict__ y_out, long long rows, int C, int lpr, float eps,
                                                   int Hr, int Wr, int Hp, int Wp, const float* __restrict__ xoff) {
    const int lane = threadIdx.x & 63;
    const int rows_per_wave = 64 / lpr;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int li = lane % lpr;
    // per-channel vectors are the same for every row: loaded once per wave, NORM_ROW_ITERS row groups reuse them
    float ov[PASSES][8], wv[PASSES][8], bv[PASSES][8], gv[PASSES][8];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int c0 = (p * lpr + li) * 8;
        if (xoff) Vec8<float>::load(xoff + c0, ov[p]);
        if (RESID && gamma) Vec8<T>::load(gamma + c0, gv[p]);
        if (y_out) { Vec8<T>::load(w + c0, wv[p]); Vec8<T>::load(b + c0, bv[p]); }
    }
    for (int it = 0; it < NORM_ROW_ITERS; ++it) {
        const long long row = (wave * NORM_ROW_ITERS + it) * rows_per_wave + lane / lpr;
        if (__builtin_amdgcn_readfirstlane((int)((wave * NORM_ROW_ITERS + it) * rows_per_wave >= rows))) break;   // wave-uniform: nothing left
        const bool live = row < rows;                                        // dead lanes still join the shuffles
        float v[PASSES][8];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int c0 = (p * lpr + li) * 8;
            if (live) {
                Vec8<T>::load(x + row * C + c0, v[p]);
                if (xoff) {                                                  // y = LN(x + xoff): a per-channel float32 offset carried outside x
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[p][k] += ov[p][k];
                }
                if (RESID) {
                    float av[8];
                    Vec8<T>::load(a + row * C + c0, av);
                    if (gamma) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[p][k] = fmaf(gv[p][k], av[k], v[p][k]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[p][k] += av[k];
                    }
                    Vec8<T>::store(x_out + row * C + c0, v[p]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[p][k] = 0.0f;
            }
        }
        if (!y_out) continue;                                                // residual only (last sub-layer of a level)
        if (RESID && sizeof(T) == 2) {
            // LN sees what the next op would read back: the rounded residual stream
#pragma unroll
            for (int p = 0; p < PASSES; ++p)
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    const uint32_t u = Vec8<__hip_bfloat16>::pack(v[p][k], v[p][k + 1]);
                    v[p][k] = __uint_as_float(u << 16); v[p][k + 1] = __uint_as_float(u & 0xffff0000u);
                }
        }
        float s = 0.0f;
#pragma unroll
        for (int p = 0; p < PASSES; ++p)
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[p][k];
        const float mean = group_sum(s, lpr) / (float)C;
        float q = 0.0f;
#pragma unroll
        for (int p = 0; p < PASSES; ++p)
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float d = v[p][k] - mean; q = fmaf(d, d, q); }
        const float rstd = rsqrtf(group_sum(q, lpr) / (float)C + eps);
        if (!live) continue;
        long long yrow = row;                                                // optional scatter into a zero-padded token grid
        if (Hp) {
            const long long hw = (long long)Hr * Wr, bi = row / hw, rem = row - bi * hw;
            yrow = (bi * Hp + rem / Wr) * Wp + rem % Wr;
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int c0 = (p * lpr + li) * 8;
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = fmaf((v[p][k] - mean) * rstd, wv[p][k], bv[p][k]);
            Vec8<T>::store(y_out + yrow * C + c0, o);
        }
    }
}

// Bilinear x2 up-sampling of an NHWC tensor (F.interpolate(scale_factor=2, mode="bilinear", align_corners=False), the
// SETR-UP head's Upsample, mmseg/ops/wrappers.py:30-51) with the preceding ReLU folded into the loads: one thread per
// 8 channels of one output pixel, four 16-byte loads, one 16-byte store.
template <typename T> __device__ __forceinline__ float through(float v);
template <> __device__ __forceinline__ float through<float>(float v) { return v; }
template <> __device__ __forceinline__ float through<__hip_bfloat16>(float v) { return __bfloat162float(__float2bfloat16(v)); }

template <typename T, bool RELU>
__global__ __launch_bounds__(256) void upsample2x_kernel(const T* __restrict__ x, const T* __restrict__ bias, const T* add, T* y, int B, int H, int W,
                                                         int C) {
    // One thread per 8 channels of a 2 x 2 OUTPUT block {2i+1, 2i+2} x {2j+1, 2j+2}, i = -1 .. H-1, j = -1 .. W-1: its four pixels
    // interpolate the same 2 x 2 input block (rows i, i+1, columns j, j+1, clamped to the image), so a thread makes four 16-byte
    // loads for up to four 16-byte stores — one load per output instead of four (the pixel-per-thread form was bound by its
    // 4x re-reads through L1 / L2: 0.45 ms for the 1 GB of the 32 x 32 -> 64 x 64 stage).  Each output still uses the weights of
    // PyTorch's formula for ITS coordinate (a clamped border row enters with the weight the formula gives it), so the result is
    // bit-identical to the per-pixel form.
    // grid: x = (image, block row) = b * (H + 1) + i + 1, y = 256-thread pieces of a block row ((W + 1) blocks x C/8 groups).
    const uint32_t cg = (uint32_t)C >> 3;
    const uint32_t e = blockIdx.y * 256u + threadIdx.x;
    if (e >= ((uint32_t)W + 1u) * cg) return;
    const uint32_t jb = e / cg;
    const int c0 = (int)(e - jb * cg) * 8, j = (int)jb - 1;
    const int b = (int)(blockIdx.x / ((uint32_t)H + 1u)), i = (int)(blockIdx.x - (uint32_t)b * ((uint32_t)H + 1u)) - 1;
    const int r0 = max(i, 0), r1 = min(i + 1, H - 1), q0 = max(j, 0), q1 = min(j + 1, W - 1);
    const T* base = x + (size_t)b * H * W * C + c0;
    float v[2][2][8];
    Vec8<T>::load(base + ((size_t)r0 * W + q0) * C, v[0][0]);
    Vec8<T>::load(base + ((size_t)r0 * W + q1) * C, v[0][1]);
    Vec8<T>::load(base + ((size_t)r1 * W + q0) * C, v[1][0]);
    Vec8<T>::load(base + ((size_t)r1 * W + q1) * C, v[1][1]);
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                  // the convolution's (BN-folded) bias rides along
    if (bias) Vec8<T>::load(bias + c0, bv);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float t = v[a][d][k] + bv[k]; v[a][d][k] = RELU ? fmaxf(t, 0.0f) : t; }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int oy = 2 * i + 1 + dy;
        if (oy < 0 || oy >= 2 * H) continue;
        // source coordinate (dst + 0.5) / 2 - 0.5, clamped below at 0 (PyTorch's area_pixel_compute_source_index)
        const float sy = fmaxf(((float)oy + 0.5f) * 0.5f - 0.5f, 0.0f);
        const int y0 = (int)sy;
        const float ly = sy - (float)y0, hy = 1.0f - ly;
        // the formula's rows (y0, y1) are this block's (r0, r1) — on output row 0 it names (0, 1) with weights (1, 0) and the
        // block holds (0, 0): the zero-weight term is a finite value times 0 either way; the last row clamps to (H-1, H-1) in both
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int ox = 2 * j + 1 + dx;
            if (ox < 0 || ox >= 2 * W) continue;
            const float sx = fmaxf(((float)ox + 0.5f) * 0.5f - 0.5f, 0.0f);
            const int x0 = (int)sx;
            const float lx = sx - (float)x0, hx = 1.0f - lx;
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = hy * (hx * v[0][0][k] + lx * v[0][1][k]) + ly * (hx * v[1][0][k] + lx * v[1][1][k]);
            const size_t at = (((size_t)b * 2 * H + oy) * 2 * W + ox) * C + c0;
            if (add) {                                                      // y = add + resize(x) (the FPN's top-down step, uper_head.py:103-108;
                float ad[8];                                                // add may be y): the resized value rounds to T before the sum, as the
                Vec8<T>::load(add + at, ad);                                // framework's two kernels leave it
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = ad[k] + through<T>(o[k]);
            }
            Vec8<T>::store(y + at, o);
        }
    }
}

int upsample2x_launch(const void* x, const void* bias, const void* add, void* y, int B, int H, int W, int C, int relu, int dtype, hipStream_t stream) {
    const long long per_row = ((long long)W + 1) * (C / 8), rows = (long long)B * (H + 1);
    if ((per_row + 255) / 256 > 65535 || rows >= (1LL << 31)) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)rows, (unsigned)((per_row + 255) / 256));
#define PPN_UP(T, R) hipLaunchKernelGGL((upsample2x_kernel<T, R>), grid, dim3(256), 0, stream, (const T*)x, (const T*)bias, (const T*)add, (T*)y, B, H, W, C)
    if (dtype == 0) { if (relu) PPN_UP(float, true); else PPN_UP(float, false); }
    else { if (relu) PPN_UP(__hip_bfloat16, true); else PPN_UP(__hip_bfloat16, false); }
#undef PPN_UP
    return (int)hipGetLastError();
}

// Resize + channel concatenation of up to 8 NHWC tensors in ONE pass: out[b][i][j][off_l + c] = resize(x_l)[b][i][j][c], level 0 (and any
// level of its size) copied.  Two places of UPerHead use it: the FPN output assembly (uper_head.py:117-127: every FPN level resized
// to the finest one's size, then concatenated: 4 levels of `channels`) and the pyramid pooling module's output (psp_head.py:48-60 +
// uper_head.py:76-84: the input and its 1 / 2 / 3 / 6-bin pooled, projected copies resized back: 1024 + 4 x `channels`).  One thread
// per 8 channels of one (pixel, level), the level = blockIdx.y (uniform: its geometry stays in scalars, a wave stores runs of whole
// 16-byte pieces); each output uses PyTorch's formula for its coordinate (source index scale * (dst + 0.5) - 0.5 clamped at 0 with
// scale = in / out as a float, the four taps weighted in float32: upsample_bilinear2d, align_corners False).  Replaces, for the FPN,
// three interpolate launches, the concatenation and the channels_last copy behind it.
struct ConcatParams {
    const void* x[8];
    int H[8], W[8], C[8], off[8];
    int Ctot;
};
template <typename T>
__global__ __launch_bounds__(256) void resize_concat_kernel(ConcatParams p, T* __restrict__ out, int B) {
    const int l = (int)blockIdx.y;
    const int cg = p.C[l] >> 3;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int H0 = p.H[0], W0 = p.W[0];
    if (idx >= (long long)B * H0 * W0 * cg) return;
    const int g = (int)(idx % cg);
    const long long pix = idx / cg;
    const int j = (int)(pix % W0), i = (int)((pix / W0) % H0), b = (int)(pix / ((long long)W0 * H0));
    const int Hl = p.H[l], Wl = p.W[l], C = p.C[l];
    const T* src = reinterpret_cast<const T*>(p.x[l]) + (size_t)b * Hl * Wl * C + g * 8;
    T* dst = out + (size_t)pix * p.Ctot + p.off[l] + g * 8;
    float o[8];
    if (Hl == H0 && Wl == W0) {
        Vec8<T>::load(src + ((size_t)i * Wl + j) * C, o);
    } else {
        const float sh = (float)Hl / (float)H0, sw = (float)Wl / (float)W0;
        const float sy = fmaxf(sh * ((float)i + 0.5f) - 0.5f, 0.0f), sx = fmaxf(sw * ((float)j + 0.5f) - 0.5f, 0.0f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < Hl - 1 ? 1 : 0), x1 = x0 + (x0 < Wl - 1 ? 1 : 0);
        const float ly = sy - (float)y0, hy = 1.0f - ly, lx = sx - (float)x0, hx = 1.0f - lx;
        float a[8], bq[8], c[8], d[8];
        Vec8<T>::load(src + ((size_t)y0 * Wl + x0) * C, a);
        Vec8<T>::load(src + ((size_t)y0 * Wl + x1) * C, bq);
        Vec8<T>::load(src + ((size_t)y1 * Wl + x0) * C, c);
        Vec8<T>::load(src + ((size_t)y1 * Wl + x1) * C, d);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = hy * (hx * a[k] + lx * bq[k]) + ly * (hx * c[k] + lx * d[k]);
    }
    Vec8<T>::store(dst, o);
}

int resize_concat_launch(const void* const* x, const int* hw, const int* ch, int n, void* out, int B, int dtype, hipStream_t stream) {
    ConcatParams p;
    int off = 0, cmax = 0;
    for (int l = 0; l < 8; ++l) {
        const int k = l < n ? l : 0;
        p.x[l] = x[k]; p.H[l] = hw[2 * k]; p.W[l] = hw[2 * k + 1]; p.C[l] = ch[k]; p.off[l] = off;
        if (l < n) { off += ch[l]; cmax = ch[l] > cmax ? ch[l] : cmax; }
    }
    p.Ctot = off;
    const long long total = (long long)B * p.H[0] * p.W[0] * (cmax / 8);
    if ((total + 255) / 256 >= (1LL << 31)) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)((total + 255) / 256), (unsigned)n);
    if (dtype == 0) hipLaunchKernelGGL(resize_concat_kernel<float>, grid, dim3(256), 0, stream, p, (float*)out, B);
    else hipLaunchKernelGGL(resize_concat_kernel<__hip_bfloat16>, grid, dim3(256), 0, stream, p, (__hip_bfloat16*)out, B);
    return (int)hipGetLastError();
}

// The pyramid pooling module's adaptive average pools (psp_head.py:33-38: nn.AdaptiveAvgPool2d(s) for s in pool_scales) of one NHWC
// tensor in ONE launch: y_k [B][s_k][s_k][C], bin (i, j) = the mean over rows floor(i H / s) .. ceil((i + 1) H / s) - 1 and the
// columns alike (PyTorch's bins), summed in float32.  One thread per 8 channels of one bin; blockIdx.y = the scale.
struct PoolParams {
    void* y[4];
    int s[4];
};
template <typename T>
__global__ __launch_bounds__(256) void adaptive_pools_kernel(const T* __restrict__ x, PoolParams p, int B, int H, int W, int C) {
    const int k = (int)blockIdx.y, s = p.s[k], cg = C >> 3;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * s * s * cg) return;
    const int g = (int)(idx % cg);
    const long long bin = idx / cg;
    const int j = (int)(bin % s), i = (int)((bin / s) % s), b = (int)(bin / ((long long)s * s));
    const int r0 = (i * H) / s, r1 = ((i + 1) * H + s - 1) / s, c0 = (j * W) / s, c1 = ((j + 1) * W + s - 1) / s;
    const T* src = x + (size_t)b * H * W * C + g * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; ++r)
        for (int c = c0; c < c1; ++c) {
            float v[8];
            Vec8<T>::load(src + ((size_t)r * W + c) * C, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    const float inv = (float)((r1 - r0) * (c1 - c0));
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = acc[e] / inv;
    Vec8<T>::store(reinterpret_cast<T*>(p.y[k]) + (size_t)bin * C + g * 8, acc);
}

int adaptive_pools_launch(const void* x, void* const* y, const int* scales, int n, int B, int H, int W, int C, int dtype, hipStream_t stream) {
    PoolParams p;
    int smax = 0;
    for (int k = 0; k < 4; ++k) { p.y[k] = y[k < n ? k : 0]; p.s[k] = scales[k < n ? k : 0]; if (k < n && scales[k] > smax) smax = scales[k]; }
    const long long total = (long long)B * smax * smax * (C / 8);
    if ((total + 255) / 256 >= (1LL << 31)) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)((total + 255) / 256), (unsigned)n);
    if (dtype == 0) hipLaunchKernelGGL(adaptive_pools_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, p, B, H, W, C);
    else hipLaunchKernelGGL(adaptive_pools_kernel<__hip_bfloat16>, grid, dim3(256), 0, stream, (const __hip_bfloat16*)x, p, B, H, W, C);
    return (int)hipGetLastError();
}

// LayerNorm (+ optional residual, as norm_kernel) for narrow rows, one THREAD per row: GenNet's ViT has C = 24 — three
// 16-byte pieces — where a library LayerNorm spends 0.26 ms on 12 MB.  C a multiple of 8, C <= 64.
template <typename T, int NV>
__global__ __launch_bounds__(256) void norm_rows_kernel(const T* __restrict__ x, const T* __restrict__ a, const T* __restrict__ gamma,
                                                        const T* __restrict__ w, const T* __restrict__ b, T* __restrict__ x_out,
                                                        T* __restrict__ y_out, long long rows, float eps, const float* __restrict__ xoff) {
    constexpr int C = NV * 8;
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float v[NV][8];
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        Vec8<T>::load(x + row * C + p * 8, v[p]);
        if (xoff) {
            float ov[8];
            Vec8<float>::load(xoff + p * 8, ov);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[p][k] += ov[k];
        }
        if (a) {
            float av[8];
            Vec8<T>::load(a + row * C + p * 8, av);
            if (gamma) {
                float gv[8];
                Vec8<T>::load(gamma + p * 8, gv);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[p][k] = fmaf(gv[k], av[k], v[p][k]);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[p][k] += av[k];
            }
            Vec8<T>::store(x_out + row * C + p * 8, v[p]);
            if (sizeof(T) == 2) {                                            // LN sees the rounded residual stream
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    const uint32_t u = Vec8<__hip_bfloat16>::pack(v[p][k], v[p][k + 1]);
                    v[p][k] = __uint_as_float(u << 16); v[p][k + 1] = __uint_as_float(u & 0xffff0000u);
                }
            }
        }
    }
    if (!y_out) return;
    float s = 0.0f;
#pragma unroll
    for (int p = 0; p < NV; ++p)
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[p][k];
    const float mean = s / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int p = 0; p < NV; ++p)
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = v[p][k] - mean; q = fmaf(d, d, q); }
    const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        float wv[8], bv[8], o[8];
        Vec8<T>::load(w + p * 8, wv);
        Vec8<T>::load(b + p * 8, bv);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = fmaf((v[p][k] - mean) * rstd, wv[k], bv[k]);
        Vec8<T>::store(y_out + row * C + p * 8, o);
    }
}

// In place y = leaky_relu(x + bias[c], slope) on an NHWC tensor (slope 0 = ReLU, 1 = bias only): the library convolutions run
// without bias and this one pass replaces the separate add_ and activation kernels.  C a multiple of 8.
template <typename T>
__global__ __launch_bounds__(256) void bias_act_kernel(T* __restrict__ x, const T* __restrict__ bias, long long n8, int C, float slope) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    // channel group of element i: the block's first element is reduced on the scalar unit, the lane offset in 32 bits
    const uint32_t cg = (uint32_t)C >> 3, blk0 = (uint32_t)(((unsigned long long)blockIdx.x * 256ull) % cg);
    const int c0 = (int)(((blk0 + threadIdx.x) % cg) * 8u);
    float v[8], bv[8];
    Vec8<T>::load(x + i * 8, v);
    Vec8<T>::load(bias + c0, bv);
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float t = v[k] + bv[k]; v[k] = t > 0.0f ? t : t * slope; }
    Vec8<T>::store(x + i * 8, v);
}

int bias_act_launch(void* x, const void* bias, long long n, int C, float slope, int dtype, hipStream_t stream) {
    const long long n8 = n / 8;
    const dim3 grid((unsigned)((n8 + 255) / 256));
    if (dtype == 0) hipLaunchKernelGGL((bias_act_kernel<float>), grid, dim3(256), 0, stream, (float*)x, (const float*)bias, n8, C, slope);
    else hipLaunchKernelGGL((bias_act_kernel<__hip_bfloat16>), grid, dim3(256), 0, stream, (__hip_bfloat16*)x, (const __hip_bfloat16*)bias, n8, C, slope);
    return (int)hipGetLastError();
}

// SegNet's input from stage B's occupancy codes in one kernel: the reference renders the map as an RGB JPEG (free white,
// obstacle black, start / goal red, process_map.py:120,128) and normalises it (x - mean) / std per channel
// (planning_seg.py:12-41).  Only six values exist (three channels x {0, 255}); eight pixels per thread, NHWC output.
template <typename T>
__global__ __launch_bounds__(256) void grid_image_kernel(const uint8_t* __restrict__ grid, T* __restrict__ img, long long n8, float lo0,
                                                         float lo1, float lo2, float hi0, float hi1, float hi2) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const uint2 g = *reinterpret_cast<const uint2*>(grid + i * 8);
    float v[24];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t code = ((k < 4 ? g.x : g.y) >> (8 * (k & 3))) & 0xffu;
        const bool free_ = code == PPN_GRID_FREE, mark = code == PPN_GRID_MARK;
        v[3 * k] = (free_ || mark) ? hi0 : lo0;
        v[3 * k + 1] = free_ ? hi1 : lo1;
        v[3 * k + 2] = free_ ? hi2 : lo2;
    }
    T* out = img + i * 24;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = v[q * 8 + k];
        Vec8<T>::store(out + q * 8, t);
    }
}

int grid_image_launch(const uint8_t* grid, void* img, long long n, const float* mean, const float* stdv, int dtype, hipStream_t stream) {
    const long long n8 = n / 8;
    const dim3 g((unsigned)((n8 + 255) / 256));
    const float lo0 = (0.0f - mean[0]) / stdv[0], lo1 = (0.0f - mean[1]) / stdv[1], lo2 = (0.0f - mean[2]) / stdv[2];
    const float hi0 = (255.0f - mean[0]) / stdv[0], hi1 = (255.0f - mean[1]) / stdv[1], hi2 = (255.0f - mean[2]) / stdv[2];
    if (dtype == 0) hipLaunchKernelGGL((grid_image_kernel<float>), g, dim3(256), 0, stream, grid, (float*)img, n8, lo0, lo1, lo2, hi0, hi1, hi2);
    else hipLaunchKernelGGL((grid_image_kernel<__hip_bfloat16>), g, dim3(256), 0, stream, grid, (__hip_bfloat16*)img, n8, lo0, lo1, lo2, hi0, hi1, hi2);
    return (int)hipGetLastError();
}

// SegNet's output tail for two classes in one kernel (setr_up_head.py:78-80 Upsample x2 of the logits, encoder_decoder.py:76-79
// resize to the input size, :242,257 softmax + argmax): lo [B][2][h][w] -> bilinear to [2h][2w] (rounded to T, as the
// materialised tensor would be) -> bilinear to [Ho][Wo] (rounded to T) -> float32 softmax over the two classes -> label.
// One thread per output pixel; the 16 low-resolution taps per class come from L1/L2 (the logits are a few MB).
template <typename T>
__device__ __forceinline__ float round_to(float v) {
    if constexpr (sizeof(T) == 2) { const uint32_t u = Vec8<__hip_bfloat16>::pack(v, 0.0f); return __uint_as_float(u << 16); }
    return v;
}
template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }

__device__ __forceinline__ void bil_src(int dst, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
    const float s = fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.0f);        // area_pixel_compute_source_index, align_corners=False
    i0 = (int)s;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - (float)i0; l0 = 1.0f - l1;
}

template <typename T>
__global__ __launch_bounds__(256) void seg_labels_kernel(const T* __restrict__ lo, uint8_t* __restrict__ labels, int B, int h, int w, int Ho,
                                                         int Wo) {
    // grid: x = image, y = output row, z = 256-pixel pieces of the row — no per-thread division of a flat 64-bit index
    const int ox = (int)(blockIdx.z * 256u + threadIdx.x), oy = (int)blockIdx.y, b = (int)blockIdx.x;
    if (ox >= Wo) return;
    const size_t idx = ((size_t)b * Ho + oy) * Wo + ox;
    const int hm = 2 * h, wm = 2 * w;                                       // the head's x2 stage
    int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
    bil_src(oy, (float)hm / (float)Ho, hm, y0, y1, ly0, ly1);
    bil_src(ox, (float)wm / (float)Wo, wm, x0, x1, lx0, lx1);
    float logit[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const T* plane = lo + ((size_t)b * 2 + c) * h * w;
        auto mid = [&](int my, int mx) {                                    // one pixel of the x2 stage, rounded to T
            int a0, a1, c0, c1; float la0, la1, lc0, lc1;
            bil_src(my, 0.5f, h, a0, a1, la0, la1);
            bil_src(mx, 0.5f, w, c0, c1, lc0, lc1);
            const float v = la0 * (lc0 * ldf<T>(plane + a0 * w + c0) + lc1 * ldf<T>(plane + a0 * w + c1)) +
                            la1 * (lc0 * ldf<T>(plane + a1 * w + c0) + lc1 * ldf<T>(plane + a1 * w + c1));
            return round_to<T>(v);
        };
        const float v = ly0 * (lx0 * mid(y0, x0) + lx1 * mid(y0, x1)) + ly1 * (lx0 * mid(y1, x0) + lx1 * mid(y1, x1));
        logit[c] = round_to<T>(v);
    }
    const float m = fmaxf(logit[0], logit[1]);                              // F.softmax(logits.float(), dim=1).argmax(dim=1)
    const float e0 = expf(logit[0] - m), e1 = expf(logit[1] - m);
    const float sum = e0 + e1;
    labels[idx] = (e1 / sum > e0 / sum) ? 1 : 0;                            // argmax keeps the first maximum
}

int seg_labels_launch(const void* lo, uint8_t* labels, int B, int h, int w, int Ho, int Wo, int dtype, hipStream_t stream) {
    if (Ho > 65535 || (Wo + 255) / 256 > 65535) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)B, (unsigned)Ho, (unsigned)((Wo + 255) / 256));
    if (dtype == 0) hipLaunchKernelGGL((seg_labels_kernel<float>), grid, dim3(256), 0, stream, (const float*)lo, labels, B, h, w, Ho, Wo);
    else hipLaunchKernelGGL((seg_labels_kernel<__hip_bfloat16>), grid, dim3(256), 0, stream, (const __hip_bfloat16*)lo, labels, B, h, w, Ho, Wo);
    return (int)hipGetLastError();
}

template <typename T>
static int launch_norm(const void* x, const void* a, const void* gamma, const void* w, const void* b, void* x_out, void* y_out,
                       long long rows, int C, float eps, int Hr, int Wr, int Hp, int Wp, const void* xoff, hipStream_t stream) {
    if (C <= 64 && C % 8 == 0 && !Hp) {                                      // narrow rows: one thread per row
        const dim3 g((unsigned)((rows + 255) / 256));
#define PPN_ROWS(NV) hipLaunchKernelGGL((norm_rows_kernel<T, NV>), g, dim3(256), 0, stream, (const T*)x, (const T*)a, (const T*)gamma, \
    (const T*)w, (const T*)b, (T*)x_out, (T*)y_out, rows, eps, (const float*)xoff)
        switch (C / 8) {
            case 1: PPN_ROWS(1); break; case 2: PPN_ROWS(2); break; case 3: PPN_ROWS(3); break; case 4: PPN_ROWS(4); break;
            case 5: PPN_ROWS(5); break; case 6: PPN_ROWS(6); break; case 7: PPN_ROWS(7); break; default: PPN_ROWS(8); break;
        }
#undef PPN_ROWS
        return (int)hipGetLastError();
    }
    int lpr = C / 8, passes = 1;
    if (lpr > 64) { passes = lpr / 64; lpr = 64; }
    if (passes > 2 || (lpr & (lpr - 1)) != 0 || lpr * 8 * passes != C) return -1;
    const long long rows_per_block = 4LL * (64 / lpr) * NORM_ROW_ITERS;
    const dim3 grid((unsigned)((rows + rows_per_block - 1) / rows_per_block));
    const bool resid = a != nullptr;
#define PPN_NORM_LAUNCH(P, R) hipLaunchKernelGGL((norm_kernel<T, P, R>), grid, dim3(256), 0, stream, (const T*)x, (const T*)a, \
    (const T*)gamma, (const T*)w, (const T*)b, (T*)x_out, (T*)y_out, rows, C, lpr, eps, Hr, Wr, Hp, Wp, (const float*)xoff)
    if (passes == 1) { if (resid) PPN_NORM_LAUNCH(1, true); else PPN_NORM_LAUNCH(1, false); }
    else { if (resid) PPN_NORM_LAUNCH(2, true); else PPN_NORM_LAUNCH(2, false); }
#undef PPN_NORM_LAUNCH
    return (int)hipGetLastError();
}

int norm_launch(const void* x, const void* a, const void* gamma, const void* w, const void* b, void* x_out, void* y_out,
                long long rows, int C, float eps, int dtype, int Hr, int Wr, int Hp, int Wp, const void* xoff, hipStream_t stream) {
    return dtype == 0 ? launch_norm<float>(x, a, gamma, w, b, x_out, y_out, rows, C, eps, Hr, Wr, Hp, Wp, xoff, stream)
                      : launch_norm<__hip_bfloat16>(x, a, gamma, w, b, x_out, y_out, rows, C, eps, Hr, Wr, Hp, Wp, xoff, stream);
}

}  // namespace ppn
