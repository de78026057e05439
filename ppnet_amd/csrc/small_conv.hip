// small_conv.hip — the two single-channel 3x3 convolutions at the ends of GenNet's AE-ViT (GenNet/networks/ae_vit.py:17-20,58:
// conv_first 1 -> dim, conv_final dim -> 1, both stride 1, padding 1, at the full R x R resolution) as direct kernels.
// With one input or one output channel there is no GEMM to speak of — 216 FMAs per pixel against 48 bytes moved — and the
// library's implicit-GEMM kernels take 0.54 + 0.72 ms per 256-problem batch where the tensors' HBM time is 0.16 ms each.
// NHWC (channels_last) on the multi-channel side, float32 accumulation, bias (+ LeakyReLU for conv_first) fused.
#include <hip/hip_bf16.h>
#include <cstdlib>
#include "ppn_device.h"
#include "ppn_kernels.h"

namespace ppn {

namespace {
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const __hip_bfloat16* p) { return __bfloat162float(*p); }
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) { return pack_bf16x2(lo, hi); }
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<__hip_bfloat16>(__hip_bfloat16* p, const float (&v)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
}
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<__hip_bfloat16>(const __hip_bfloat16* p, float (&v)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
}  // namespace

// y[b][i][j][co] = leaky(bias[co] + sum_{di,dj} w[co][0][di][dj] * x[b][i+di-1][j+dj-1]), zero padding.  One thread per
// output pixel, CG = Cout / 8 groups of 8 channels; the 9 * Cout weights sit in LDS as [tap][co] (broadcast reads).
template <typename T, int CG>
__global__ __launch_bounds__(256) void conv3x3_c1_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         T* __restrict__ y, int B, int H, int W, float slope) {
    constexpr int C = CG * 8;
    __shared__ float ws[9][C];
    __shared__ float bs[C];
    for (int t = threadIdx.x; t < 9 * C; t += 256) { const int co = t / 9, tap = t - co * 9; ws[tap][co] = w[t]; }
    for (int t = threadIdx.x; t < C; t += 256) bs[t] = bias[t];
    __syncthreads();
    const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * H * W;
    if (px >= total) return;
    const int j = (int)(px % W);
    const long long r = px / W;
    const int i = (int)(r % H);
    const T* xb = x + (r - i) * W;                                           // image base
    float in[9];
#pragma unroll
    for (int di = 0; di < 3; ++di)
#pragma unroll
        for (int dj = 0; dj < 3; ++dj) {
            const int ii = i + di - 1, jj = j + dj - 1;
            in[di * 3 + dj] = (ii >= 0 && ii < H && jj >= 0 && jj < W) ? ld1(xb + (long long)ii * W + jj) : 0.0f;
        }
    T* out = y + px * C;
#pragma unroll
    for (int g = 0; g < CG; ++g) {
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = bs[g * 8 + k];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = fmaf(ws[tap][g * 8 + k], in[tap], acc[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = acc[k] > 0.0f ? acc[k] : acc[k] * slope;
        store8<T>(out + g * 8, acc);
    }
}

// y[b][i][j] = bias + sum_{di,dj,ci} w[0][ci][di][dj] * x[b][i+di-1][j+dj-1][ci], zero padding.  A 16 x 16 tile of outputs
// per workgroup, its 18 x 18 x Cin halo staged in LDS as float (entry pitch Cin + 4 floats: 16-byte aligned, and the 16
// lanes of a tile row then start their ds_read_b128 on 16 different banks), weights in LDS as [tap][ci].
template <typename T, int CG>
__global__ __launch_bounds__(256) void conv3x3_to1_kernel(const T* __restrict__ x, const float* __restrict__ w, float bias,
                                                          T* __restrict__ y, int B, int H, int W) {
    constexpr int C = CG * 8, PITCH = C + 4, HT = 18;
    __shared__ __attribute__((aligned(16))) float tile[HT * HT * PITCH];
    __shared__ __attribute__((aligned(16))) float ws[9][C];
    for (int t = threadIdx.x; t < 9 * C; t += 256) { const int ci = t / 9, tap = t - ci * 9; ws[tap][ci] = w[t]; }
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
    const int b = blockIdx.x / (tiles_x * tiles_y), tt = blockIdx.x - b * tiles_x * tiles_y;
    const int i0 = (tt / tiles_x) * 16, j0 = (tt % tiles_x) * 16;
    const T* xb = x + (long long)b * H * W * C;
    for (int p = threadIdx.x; p < HT * HT * CG; p += 256) {
        const int e = p / CG, g = p - e * CG;
        const int hi = e / HT, hj = e - hi * HT;
        const int ii = i0 + hi - 1, jj = j0 + hj - 1;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ii >= 0 && ii < H && jj >= 0 && jj < W) load8<T>(xb + ((long long)ii * W + jj) * C + g * 8, v);
        *reinterpret_cast<float4*>(tile + e * PITCH + g * 8) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(tile + e * PITCH + g * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    __syncthreads();
    const int ti = threadIdx.x / 16, tj = threadIdx.x % 16;
    const int i = i0 + ti, j = j0 + tj;
    if (i >= H || j >= W) return;
    float acc = bias;
#pragma unroll
    for (int di = 0; di < 3; ++di)
#pragma unroll
        for (int dj = 0; dj < 3; ++dj) {
            const float4* e = reinterpret_cast<const float4*>(tile + ((ti + di) * HT + tj + dj) * PITCH);
            const float4* wv = reinterpret_cast<const float4*>(ws[di * 3 + dj]);
#pragma unroll
            for (int c4 = 0; c4 < C / 4; ++c4) {
                const float4 a = e[c4], b4 = wv[c4];
                acc = fmaf(b4.x, a.x, acc); acc = fmaf(b4.y, a.y, acc); acc = fmaf(b4.z, a.z, acc); acc = fmaf(b4.w, a.w, acc);
            }
        }
    T* out = y + ((long long)b * H + i) * W + j;
    if constexpr (sizeof(T) == 2) *out = __float2bfloat16(acc); else *out = acc;
}

// The same convolution for GenNet's case (24 input channels, bfloat16) on the matrix cores, as TAP RESPONSES: with one output
// channel the nine taps can be the MFMA's rows — T[tap][p] = sum_c w[tap][c] x[p][c] for every pixel p of the tile's halo is ONE
// v_mfma_f32_16x16x32_bf16 per 16 pixels (rows = taps, 9 of 16 used; k = channels, 24 of 32 used; the float32 weights enter as
// hi + lo bfloat16 fragments, two MFMAs, so the products are the float32 kernel's to 2^-17) — and the output is nine shifted reads,
// y[i][j] = bias + sum_{di,dj} T[3 di + dj][i + di - 1][j + dj - 1].  Every input pixel is loaded from HBM / L2 exactly once per
// tile, 16 bytes per lane straight into the B operand (pixel = column, 8 channels per lane quarter) — no LDS staging of the
// image, 1/5 of the VALU form's LDS reads (it spends 108 ds_read_b128 per pixel on 216 FMAs and runs at 2 TB/s of its 805 MB).
// Tile: 32 x 16 outputs per 256-thread workgroup, halo 34 x 18 = 612 pixels = 39 groups of 16 over the 4 waves, T in LDS as
// [pixel][12 floats] (a lane's 4 taps = one 16-byte write), 29 KB.
namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 t1_bf16x8;
typedef __attribute__((ext_vector_type(4))) float t1_f32x4;
constexpr int T1_W = 32, T1_H = 16, T1_HW = T1_W + 2, T1_HH = T1_H + 2, T1_PIX = T1_HW * T1_HH, T1_GROUPS = (T1_PIX + 15) / 16;
constexpr int T1_GPW = (T1_GROUPS + 3) / 4;                                // groups per wave
}  // namespace

__global__ __launch_bounds__(256) void conv3x3_to1_mfma_kernel(const __bf16* __restrict__ x, const float* __restrict__ w, float bias,
                                                               __bf16* __restrict__ y, int B, int H, int W, const __bf16* __restrict__ zero) {
    __shared__ __attribute__((aligned(16))) float Tl[T1_GROUPS * 16 * 12];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
    const int tiles_x = (W + T1_W - 1) / T1_W, tiles_y = (H + T1_H - 1) / T1_H;
    const int b = blockIdx.x / (tiles_x * tiles_y), tt = blockIdx.x - b * tiles_x * tiles_y;
    const int i0 = (tt / tiles_x) * T1_H, j0 = (tt % tiles_x) * T1_W;
    // A operand: row = tap (lane & 15), k = channel 8g .. 8g+7 (g = 3 and taps 9 .. 15: zero).  w is [24][9] float32 (ci * 9 + tap).
    t1_bf16x8 ahi, alo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float wv = (j < 9 && g < 3) ? w[(8 * g + e) * 9 + j] : 0.0f;
        const __bf16 h = (__bf16)wv;
        ahi[e] = h; alo[e] = (__bf16)(wv - (float)h);
    }
    const __bf16* xb = x + (size_t)b * H * W * 24;
    // all of this wave's pixel groups in flight before the first MFMA (unconditional addresses: out-of-image pixels and the
    // unused lane quarter read a line of zeros, which is also the convolution's zero padding)
    t1_bf16x8 fb[T1_GPW];
#pragma unroll
    for (int q = 0; q < T1_GPW; ++q) {
        const int grp = wave * T1_GPW + q;
        const int e = min(grp * 16 + j, T1_PIX - 1);
        const int hy = e / T1_HW, hx = e - hy * T1_HW;
        const int ii = i0 + hy - 1, jj = j0 + hx - 1;
        const bool in = grp < T1_GROUPS && g < 3 && ii >= 0 && ii < H && jj >= 0 && jj < W;
        const __bf16* src = in ? xb + ((size_t)ii * W + jj) * 24 + 8 * g : zero;
        fb[q] = *reinterpret_cast<const t1_bf16x8*>(src);
    }
#pragma unroll
    for (int q = 0; q < T1_GPW; ++q) {
        const int grp = wave * T1_GPW + q;
        t1_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, fb[q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, fb[q], acc, 0, 0, 0);
        // lane (j, g) holds taps 4g .. 4g+3 of pixel 16 grp + j
        if (grp < T1_GROUPS && g < 3) *reinterpret_cast<float4*>(Tl + (grp * 16 + j) * 12 + 4 * g) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty0 = threadIdx.x >> 5;               // two output rows per thread: ty0 and ty0 + 8
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int ty = ty0 + 8 * r;
        const int i = i0 + ty, jx = j0 + tx;
        float acc = bias;
#pragma unroll
        for (int di = 0; di < 3; ++di)
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) acc += Tl[((ty + di) * T1_HW + tx + dj) * 12 + di * 3 + dj];
        if (i < H && jx < W) y[((size_t)b * H + i) * W + jx] = (__bf16)acc;
    }
}

int conv3x3_c1_launch(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int Cout, float slope, int dtype,
                      hipStream_t stream) {
    const long long total = (long long)B * H * W;
    const dim3 grid((unsigned)((total + 255) / 256));
#define PPN_C1(T, CG) hipLaunchKernelGGL((conv3x3_c1_kernel<T, CG>), grid, dim3(256), 0, stream, (const T*)x, w, bias, (T*)y, B, H, W, slope)
    const int cg = Cout / 8;
    if (dtype == 0) { if (cg == 1) PPN_C1(float, 1); else if (cg == 2) PPN_C1(float, 2); else if (cg == 3) PPN_C1(float, 3); else PPN_C1(float, 4); }
    else { if (cg == 1) PPN_C1(__hip_bfloat16, 1); else if (cg == 2) PPN_C1(__hip_bfloat16, 2); else if (cg == 3) PPN_C1(__hip_bfloat16, 3); else PPN_C1(__hip_bfloat16, 4); }
#undef PPN_C1
    return (int)hipGetLastError();
}

// Per-sample min-max normalisation of GenNet's output to an 8-bit heat map (reference GenNet/predict.py:95-102:
// (y - min) / (max - min), then ToPILImage's mul(255).byte()).  One workgroup per sample: pass 1 reduces min / max over the
// sample's pixels (float32 of the stored values), pass 2 re-reads them (128 KB at 256 x 256 bfloat16: L2-resident) and writes
// u8.  The arithmetic is the reference's, in its order and in float32 — (f - lo) / (hi - lo), * 255, truncate — so the result
// is bit-identical to the torch composition it replaces (seven elementwise / reduction launches).
template <typename T>
__global__ __launch_bounds__(1024) void heatmap_u8_kernel(const T* __restrict__ y, uint8_t* __restrict__ out, int n) {
    __shared__ float red[2][16];
    const T* src = y + (size_t)blockIdx.x * n;
    float lo = 3.0e38f, hi = -3.0e38f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float v = ld1(src + i);
        lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lo; red[1][threadIdx.x >> 6] = hi; }
    __syncthreads();
    lo = red[0][0]; hi = red[1][0];
    for (int w = 1; w < 16; ++w) { lo = fminf(lo, red[0][w]); hi = fmaxf(hi, red[1][w]); }
    const float range = hi - lo;
    uint8_t* dst = out + (size_t)blockIdx.x * n;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float nrm = (ld1(src + i) - lo) / range;
        dst[i] = (uint8_t)(int)(nrm * 255.0f);                              // float -> uint8 as torch's .to(torch.uint8): truncation
    }
}

int heatmap_u8_launch(const void* y, uint8_t* out, int B, int n, int dtype, hipStream_t stream) {
    if (dtype == 0) hipLaunchKernelGGL((heatmap_u8_kernel<float>), dim3(B), dim3(1024), 0, stream, (const float*)y, out, n);
    else hipLaunchKernelGGL((heatmap_u8_kernel<__hip_bfloat16>), dim3(B), dim3(1024), 0, stream, (const __hip_bfloat16*)y, out, n);
    return (int)hipGetLastError();
}

int conv3x3_to1_launch(const void* x, const float* w, float bias, void* y, int B, int H, int W, int Cin, int dtype, hipStream_t stream) {
    // GenNet's shape on the matrix cores (PPNET_TO1_VALU=1 keeps the direct form for A/B runs)
    static const bool valu = getenv("PPNET_TO1_VALU") != nullptr;
    if (dtype == 1 && Cin == 24 && !valu) {
        const __bf16* zero = (const __bf16*)zero_line();
        if (!zero) return (int)hipErrorOutOfMemory;
        const long long tiles = (long long)B * ((H + T1_H - 1) / T1_H) * ((W + T1_W - 1) / T1_W);
        if (tiles >= (1LL << 31)) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(conv3x3_to1_mfma_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, (const __bf16*)x, w, bias, (__bf16*)y, B, H, W, zero);
        return (int)hipGetLastError();
    }
    const dim3 grid((unsigned)((long long)B * ((H + 15) / 16) * ((W + 15) / 16)));
#define PPN_TO1(T, CG) hipLaunchKernelGGL((conv3x3_to1_kernel<T, CG>), grid, dim3(256), 0, stream, (const T*)x, w, bias, (T*)y, B, H, W)
    const int cg = Cin / 8;
    if (dtype == 0) { if (cg == 1) PPN_TO1(float, 1); else if (cg == 2) PPN_TO1(float, 2); else if (cg == 3) PPN_TO1(float, 3); else PPN_TO1(float, 4); }
    else { if (cg == 1) PPN_TO1(__hip_bfloat16, 1); else if (cg == 2) PPN_TO1(__hip_bfloat16, 2); else if (cg == 3) PPN_TO1(__hip_bfloat16, 3); else PPN_TO1(__hip_bfloat16, 4); }
#undef PPN_TO1
    return (int)hipGetLastError();
}

}  // namespace ppn
