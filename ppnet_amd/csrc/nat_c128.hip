// nat_c128.hip — the dense half of a DiNAT level-0 layer (C = 128 channels, 1 M tokens per batch of 256 images) as two
// token-streaming kernels whose weights live in LDS (reference SegNet/nat.py:101-153: norm1 -> qkv, and
// norm2 -> fc1 -> GELU -> fc2 -> residual).  At C = 128 these projections are HBM-bound (K = 128 or 256: ~100 flop/byte) and a
// GEMM library pays one tensor pass per op; here
//
//   nat128_ln_qkv_kernel   qkv = LN(s + off) Wqkv^T + b        reads s once, writes qkv once            (2 passes -> 1 + 3)
//   nat128_ln_mlp_kernel   s  += GELU(LN(s + off) W1^T + b1) W2^T     reads s, writes s; the 256-wide hidden row never
//                                                                     leaves the registers             (7 passes -> 2)
//
// One wave owns 16 tokens at a time.  All products are computed TRANSPOSED, D^T[out][token] = W[out][:] . Y^T[:][token], so
// that the MFMA result layout (lane = token column, 4 consecutive rows per register group) is already the B-operand layout
// of the next product: the LayerNorm output feeds fc1, and GELU(fc1) feeds fc2, without a shuffle or an LDS round trip.
// Weights are staged once per workgroup as 16-byte A-operand pieces, XOR-swizzled by row so the 16 rows a fragment read
// touches fall in 16 different bank groups.  Memory instructions cover 64 CONTIGUOUS bytes per token: the four lanes of a token
// (g = 0..3) read / write adjacent 16-byte pieces in one instruction, so a wave-instruction touches 16 half-lines instead of 64
// separate 16-byte pieces — the k-slot -> channel map of the operands and the (tile, row) -> output channel map are chosen for
// that (any permutation works as long as the staged weights use the same one): k-slot (ks, g, e) is channel 32 ks + 8 g + e, and
// of every four output tiles a lane holds channels 8 g .. 8 g + 7 (tiles 0, 1) and 32 + 8 g .. 32 + 8 g + 7 (tiles 2, 3).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "ppn_device.h"
#include "ppn_kernels.h"
// The library is built with -ffp-contract=off because the generator kernels promise unfused IEEE double arithmetic (the parity
// contract with the oracle).  This file holds network arithmetic checked against float32 / float64 references to a tolerance:
// here a * b + c is one v_fma (otherwise every multiply-add of the LayerNorm, the GELU polynomial and the per-token linear
// phases is two instructions — these kernels are VALU-bound).
#pragma clang fp contract(fast)

namespace ppn {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int C128 = 128, NAT128_THREADS = 1024, NAT128_WAVES = 16;

// A&S 7.1.26 erf (|error| <= 1.5e-7): the same form as the GEMM core's GELU epilogue (mfma_gemm.h)
__device__ __forceinline__ float gelu_erf128(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.0f - poly * __expf(-z * z);
    return 0.5f * x + 0.5f * fabsf(x) * e;
}

// The same function without a transcendental: Phi(x) - 1/2 as an odd polynomial on [-4.25, 4.25] (x clamped, the end point pinned
// to 1/2 so that gelu(x) is x or 0 beyond), 8 coefficients in x^2 from a minimax fit of x Phi(x): |error| <= 9.2e-5 absolute in
// float32 — under one bfloat16 ulp of every value above 0.012 and far under the rounding of the hidden activation to bfloat16.
// Two values per instruction (v_pk_mul_f32 / v_pk_fma_f32): 10 packed operations + 2 v_med3 per pair = 24 cycles per 64 values
// against 80 for the erf form above (v_exp and v_rcp issue at a quarter of the rate) — these kernels are VALU-bound and the 256
// GELUs per token are most of their instructions.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) {
    const f32x2 xc = {__builtin_amdgcn_fmed3f(x.x, -4.25f, 4.25f), __builtin_amdgcn_fmed3f(x.y, -4.25f, 4.25f)};
    const f32x2 u = xc * xc;
    constexpr float K[8] = {3.984200563e-01f, -6.545352466e-02f, 9.257837137e-03f, -9.404510850e-04f, 6.552754503e-05f, -2.938833893e-06f,
                            7.570944350e-08f, -8.460567657e-10f};
    f32x2 q = {K[7], K[7]};
#pragma unroll
    for (int i = 6; i >= 0; --i) q = __builtin_elementwise_fma(q, u, f32x2{K[i], K[i]});
    return x * __builtin_elementwise_fma(xc, q, f32x2{0.5f, 0.5f});
}
// Diagnostic builds only (make c128abl; never shipped): nat128_ln_mlp_kernel with one part removed — 1 no row loads, 2 no stores,
// 4 no GELU, 8 no fc1 products, 16 no fc2 products, 32 no LayerNorm arithmetic.
#ifndef C128_ABL
#define C128_ABL 0
#endif
#ifndef PPN_GELU128_FORM
#define PPN_GELU128_FORM 4      // 0: erf form (A&S), 1 / 3: logistic fit (v_exp + v_rcp) packed / scalar, 2 / 4: polynomial packed / scalar — 4 ships
#endif
__device__ __forceinline__ f32x2 gelu128_pair(f32x2 x) {
#if PPN_GELU128_FORM == 0
    return f32x2{gelu_erf128(x.x), gelu_erf128(x.y)};
#elif PPN_GELU128_FORM == 1
    f32x2 x2 = x * x;
    x2 = f32x2{fminf(x2.x, 64.0f), fminf(x2.y, 64.0f)};
    const f32x2 t = x * __builtin_elementwise_fma(x2, __builtin_elementwise_fma(x2, f32x2{-1.0350827e-3f, -1.0350827e-3f}, f32x2{0.10690469f, 0.10690469f}),
                                                  f32x2{2.3009787f, 2.3009787f});
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(-t.x), __builtin_amdgcn_exp2f(-t.y)} + f32x2{1.0f, 1.0f};
    return x * f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
#elif PPN_GELU128_FORM == 3                      // logistic fit, one value per instruction
    f32x2 r;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float v = x[i], v2 = fminf(v * v, 64.0f);
        const float t = v * fmaf(v2, fmaf(v2, -1.0350827e-3f, 0.10690469f), 2.3009787f);
        r[i] = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-t));
    }
    return r;
#elif PPN_GELU128_FORM == 4                      // the polynomial, one value per instruction
    constexpr float K[8] = {3.984200563e-01f, -6.545352466e-02f, 9.257837137e-03f, -9.404510850e-04f, 6.552754503e-05f, -2.938833893e-06f,
                            7.570944350e-08f, -8.460567657e-10f};
    f32x2 r;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float v = x[i], vc = __builtin_amdgcn_fmed3f(v, -4.25f, 4.25f), u = vc * vc;
        float q = K[7];
#pragma unroll
        for (int k = 6; k >= 0; --k) q = fmaf(q, u, K[k]);
        r[i] = v * fmaf(vc, q, 0.5f);
    }
    return r;
#else
    return gelu_poly2(x);
#endif
}

// output channel of (tile t, row i) — see the header: a lane (token j, group g) holds rows 4g..4g+3 of every tile
__device__ __forceinline__ int out_channel(int t, int i) { return (t >> 2) * 64 + 32 * ((t >> 1) & 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3); }

__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// LayerNorm of 16 tokens in B-operand form.  Lane (j = lane & 15, g = lane >> 4) loads channels [32k + 8g, 32k + 8g + 8), k = 0..3,
// of token j (load k: 64 contiguous bytes per token over its four lanes); MFMA k-slot (ks, g, e) is channel 32ks + 8g + e, the
// A operand uses the same map.
// ln = float32 [3][128] in LDS: weight, bias, offset.
__device__ __forceinline__ void load_rows(const __hip_bfloat16* s, long long tok0, int lane, uint4 (&u)[4]) {
    const uint4* src = reinterpret_cast<const uint4*>(s + (tok0 + (lane & 15)) * C128 + 8 * (lane >> 4));
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = src[4 * k];
}
// The same loads issued from inline assembly: the compiler does not know they are pending, so it puts no wait in front of their
// first use — the caller does, with a COUNTED s_waitcnt (rows_landed<N>: N = the vector-memory operations the wave issues after
// these four loads and before the wait, i.e. the stores of the group in between).  hipcc's own wait for loads it can see is
// vmcnt(0) at the head of the loop — the loop-entry edge carries no stores, and it takes the stricter of the two edges — which
// makes every group wait for its predecessor's stores to be acknowledged.
typedef __attribute__((ext_vector_type(4))) unsigned int row16;
__device__ __forceinline__ void load_rows_async(const __hip_bfloat16* s, long long tok0, int lane, row16 (&u)[4]) {
    const __hip_bfloat16* src = s + (tok0 + (lane & 15)) * C128 + 8 * (lane >> 4);
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:64\n\t"
                 "global_load_dwordx4 %2, %4, off offset:128\n\tglobal_load_dwordx4 %3, %4, off offset:192"
                 : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]) : "v"(src) : "memory");
}
template <int N>
__device__ __forceinline__ void rows_landed(row16 (&u)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]) : "n"(N) : "memory");
}
__device__ __forceinline__ void ln_rows(const row16 (&v)[4], int lane, const float* ln, float eps, bf16x8 (&yf)[4]);
__device__ __forceinline__ void ln_rows(const uint4 (&u)[4], int lane, const float* ln, float eps, bf16x8 (&yf)[4]) {
    const int g = lane >> 4;
    float x[32];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 o0 = *reinterpret_cast<const float4*>(ln + 256 + 32 * k + 8 * g), o1 = *reinterpret_cast<const float4*>(ln + 256 + 32 * k + 8 * g + 4);
        x[8 * k + 0] = bf_lo(u[k].x) + o0.x; x[8 * k + 1] = bf_hi(u[k].x) + o0.y;
        x[8 * k + 2] = bf_lo(u[k].y) + o0.z; x[8 * k + 3] = bf_hi(u[k].y) + o0.w;
        x[8 * k + 4] = bf_lo(u[k].z) + o1.x; x[8 * k + 5] = bf_hi(u[k].z) + o1.y;
        x[8 * k + 6] = bf_lo(u[k].w) + o1.z; x[8 * k + 7] = bf_hi(u[k].w) + o1.w;
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) sum += x[k];
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / C128);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) { const float d = x[k] - mean; q = fmaf(d, d, q); }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q * (1.0f / C128) + eps);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 w0 = *reinterpret_cast<const float4*>(ln + 32 * k + 8 * g), w1 = *reinterpret_cast<const float4*>(ln + 32 * k + 8 * g + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(ln + 128 + 32 * k + 8 * g), b1 = *reinterpret_cast<const float4*>(ln + 128 + 32 * k + 8 * g + 4);
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) yf[k][e] = (__bf16)fmaf((x[8 * k + e] - mean) * rstd, wv[e], bv[e]);
    }
}

__device__ __forceinline__ void ln_rows(const row16 (&v)[4], int lane, const float* ln, float eps, bf16x8 (&yf)[4]) {
    const uint4 u[4] = {make_uint4(v[0].x, v[0].y, v[0].z, v[0].w), make_uint4(v[1].x, v[1].y, v[1].z, v[1].w),
                        make_uint4(v[2].x, v[2].y, v[2].z, v[2].w), make_uint4(v[3].x, v[3].y, v[3].z, v[3].w)};
    ln_rows(u, lane, ln, eps, yf);
}

// Stage a [rows][128] bf16 weight as A-operand pieces: LDS row r holds source row src_row(r); its 16 chunks of 8 channels are
// stored at chunk position c ^ (r & 15).  (Fragment read of k-step ks by lane (n, g): chunk 4ks + g of row n — conflict-free in
// every 16-lane group ds_read_b128 is served in.)
template <typename RowMap>
__device__ __forceinline__ void stage_k128(unsigned char* dst, const __hip_bfloat16* __restrict__ w, int rows, RowMap src_row) {
    for (int i = threadIdx.x; i < rows * 16; i += NAT128_THREADS) {
        const int r = i >> 4, c = i & 15;
        const uint4 v = *reinterpret_cast<const uint4*>(w + (size_t)src_row(r) * C128 + 8 * c);
        *reinterpret_cast<uint4*>(dst + ((size_t)r * 16 + (c ^ (r & 15))) * 16) = v;
    }
}

__device__ __forceinline__ void stage_ln(float* ln, const __hip_bfloat16* __restrict__ w, const __hip_bfloat16* __restrict__ b,
                                         const float* __restrict__ off) {
    for (int i = threadIdx.x; i < C128; i += NAT128_THREADS) {
        ln[i] = __bfloat162float(w[i]);
        ln[128 + i] = __bfloat162float(b[i]);
        ln[256 + i] = off ? off[i] : 0.0f;
    }
}

__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* base, int row, int chunk, int chunks_per_row) {
    return *reinterpret_cast<const bf16x8*>(base + ((size_t)row * chunks_per_row + (chunk ^ (row & 15))) * 16);
}

__device__ __forceinline__ uint2 pack4(f32x4 v) { return make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])); }

}  // namespace

// ---- LN -> qkv -----------------------------------------------------------------------------------------------------
constexpr int QKV_N = 384, QKV_TILES = QKV_N / 16;
constexpr int QKV_LDS = QKV_N * 256 + QKV_N * 4 + 3 * 128 * 4;

__global__ __launch_bounds__(NAT128_THREADS, 1) void nat128_ln_qkv_kernel(const __hip_bfloat16* __restrict__ s, const float* __restrict__ off,
                                                                          const __hip_bfloat16* __restrict__ lnw, const __hip_bfloat16* __restrict__ lnb,
                                                                          const __hip_bfloat16* __restrict__ w, const __hip_bfloat16* __restrict__ bias,
                                                                          __hip_bfloat16* __restrict__ qkv, long long groups, float eps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* wl = lds;
    float* bl = reinterpret_cast<float*>(lds + QKV_N * 256);
    float* ln = bl + QKV_N;
    stage_k128(wl, w, QKV_N, [](int r) { return out_channel(r >> 4, r & 15); });
    for (int i = threadIdx.x; i < QKV_N; i += NAT128_THREADS) bl[i] = bias ? __bfloat162float(bias[out_channel(i >> 4, i & 15)]) : 0.0f;
    stage_ln(ln, lnw, lnb, off);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    // The rows of the NEXT group are requested as soon as the LayerNorm has consumed the current ones (into the same registers):
    // they land while the products run, instead of a full memory latency at the head of every group.
    const long long step = (long long)gridDim.x * NAT128_WAVES;
    long long grp = (long long)blockIdx.x * NAT128_WAVES + wave;
    row16 u[4];
    if (grp < groups) {
        load_rows_async(s, grp * 16, lane, u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the first group: nothing behind its loads to count
    }
    for (; grp < groups; grp += step) {
        const long long tok0 = grp * 16;
        bf16x8 yf[4];
        rows_landed<2 * (QKV_TILES / 4)>(u);                          // the previous group's 12 stores may still be in flight
        ln_rows(u, lane, ln, eps, yf);
        load_rows_async(s, (grp + step < groups ? grp + step : grp) * 16, lane, u);   // (the last group re-reads its own rows)
        __hip_bfloat16* orow = qkv + (tok0 + n) * QKV_N + 8 * g;
#pragma unroll
        for (int qd = 0; qd < QKV_TILES / 4; ++qd) {                 // unrolled so that the compiler can COUNT the stores between the prefetch and
                                                                     // its use (vmcnt(12) at the head of the next group, not vmcnt(0)); the
                                                                     // scheduling barriers keep 8 fragment reads in flight, not 96
            f32x4 acc[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int t = qd * 4 + tt;
                acc[tt] = *reinterpret_cast<const f32x4*>(bl + t * 16 + 4 * g);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(wl, t * 16 + n, 4 * ks + g, 16), yf[ks], acc[tt], 0, 0, 0);
                if (tt & 1) __builtin_amdgcn_sched_barrier(0);
            }
            const uint2 p0 = pack4(acc[0]), p1 = pack4(acc[1]), p2 = pack4(acc[2]), p3 = pack4(acc[3]);
            uint4* dst = reinterpret_cast<uint4*>(orow + qd * 64);
            dst[0] = make_uint4(p0.x, p0.y, p1.x, p1.y);             // channels 64 qd + 8 g .. + 7
            dst[4] = make_uint4(p2.x, p2.y, p3.x, p3.y);             // channels 64 qd + 32 + 8 g .. + 7
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- LN -> fc1 -> GELU -> fc2 -> residual --------------------------------------------------------------------------
constexpr int MLP_H = 256;
constexpr int MLP_LDS = MLP_H * 256 + C128 * 512 + MLP_H * 4 + 3 * 128 * 4 + C128 * 4;

__global__ __launch_bounds__(NAT128_THREADS, 1) void nat128_ln_mlp_kernel(__hip_bfloat16* __restrict__ s, const float* __restrict__ off,
                                                                          const __hip_bfloat16* __restrict__ lnw, const __hip_bfloat16* __restrict__ lnb,
                                                                          const __hip_bfloat16* __restrict__ w1, const __hip_bfloat16* __restrict__ b1,
                                                                          const __hip_bfloat16* __restrict__ w2, const float* __restrict__ add,
                                                                          long long groups, float eps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* w1l = lds;                                  // [256 hidden][16 chunks]
    unsigned char* w2l = lds + MLP_H * 256;                    // [128 rows = (tile, row)][32 chunks]
    float* b1l = reinterpret_cast<float*>(w2l + C128 * 512);
    float* ln = b1l + MLP_H;
    float* addl = ln + 3 * 128;                                // [tile][row]: the per-channel constant the result starts from (or zeros)
    for (int i = threadIdx.x; i < C128; i += NAT128_THREADS) addl[i] = add ? add[out_channel(i >> 4, i & 15)] : 0.0f;
    stage_k128(w1l, w1, MLP_H, [](int r) { return r; });
    // fc2's k-slot (ks2, g, e) is hidden unit (2 ks2 + (e >> 2)) * 16 + 4g + (e & 3) — the order GELU(fc1) comes out of the
    // matrix pipe in: chunk ks2 * 4 + g of a row is two 8-byte pieces of the source row
    for (int i = threadIdx.x; i < C128 * 64; i += NAT128_THREADS) {
        const int r = i >> 6, c = (i >> 1) & 31, half = i & 1;
        const int ks2 = c >> 2, g = c & 3;
        const uint2 v = *reinterpret_cast<const uint2*>(w2 + (size_t)out_channel(r >> 4, r & 15) * MLP_H + (2 * ks2 + half) * 16 + 4 * g);
        *reinterpret_cast<uint2*>(w2l + ((size_t)r * 32 + (c ^ (r & 15))) * 16 + 8 * half) = v;
    }
    for (int i = threadIdx.x; i < MLP_H; i += NAT128_THREADS) b1l[i] = __bfloat162float(b1[i]);
    stage_ln(ln, lnw, lnb, off);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    // A lane's input channels (32 k + 8 g + e) ARE its output channels (pieces k of the header's map): the accumulators start from
    // the rows the LayerNorm read — no second read of s in the epilogue — and the rows of the NEXT group are requested into the same
    // registers as soon as both have consumed them, so they land while the products run.
    const long long step = (long long)gridDim.x * NAT128_WAVES;
    long long grp = (long long)blockIdx.x * NAT128_WAVES + wave;
    uint4 u[4];
    if (grp < groups) load_rows(s, grp * 16, lane, u);
    for (; grp < groups; grp += step) {
        const long long tok0 = grp * 16;
        bf16x8 yf[4];
        if (C128_ABL & 32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) yf[k] = __builtin_bit_cast(bf16x8, u[k]);
        } else {
            ln_rows(u, lane, ln, eps, yf);
        }
        f32x4 acc[8];                                                // the residual rows + `add` (the level's accumulated biases, last layer) or 0
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(addl + (2 * h) * 16 + 4 * g), c1 = *reinterpret_cast<const f32x4*>(addl + (2 * h + 1) * 16 + 4 * g);
            acc[2 * h] = f32x4{bf_lo(u[h].x) + c0[0], bf_hi(u[h].x) + c0[1], bf_lo(u[h].y) + c0[2], bf_hi(u[h].y) + c0[3]};
            acc[2 * h + 1] = f32x4{bf_lo(u[h].z) + c1[0], bf_hi(u[h].z) + c1[1], bf_lo(u[h].w) + c1[2], bf_hi(u[h].w) + c1[3]};
        }
        if (!(C128_ABL & 1)) load_rows(s, (grp + step < groups ? grp + step : grp) * 16, lane, u);   // unconditional: see nat128_proj_add_kernel
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            bf16x8 hf[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 h[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int t = half * 8 + 2 * k + e;
                    h[e] = *reinterpret_cast<const f32x4*>(b1l + t * 16 + 4 * g);
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        if (C128_ABL & 8) { h[e][ks] += (float)yf[ks][e]; continue; }
                        h[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(w1l, t * 16 + n, 4 * ks + g, 16), yf[ks], h[e], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f32x2 gl = (C128_ABL & 4) ? f32x2{h[e >> 2][e & 3], h[e >> 2][(e & 3) + 1]} : gelu128_pair(f32x2{h[e >> 2][e & 3], h[e >> 2][(e & 3) + 1]});
                    hf[k][e] = (__bf16)gl.x;
                    hf[k][e + 1] = (__bf16)gl.y;
                }
                __builtin_amdgcn_sched_barrier(0);                   // keep the fragment reads of later steps from being hoisted (spills)
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (C128_ABL & 16) { acc[t][k] += (float)hf[k][t]; continue; }
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(w2l, t * 16 + n, (half * 4 + k) * 4 + g, 32), hf[k], acc[t], 0, 0, 0);
                }
                if (t & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        uint4* p = reinterpret_cast<uint4*>(s + (tok0 + n) * C128 + 8 * g);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const uint2 lo = pack4(acc[2 * h]), hi = pack4(acc[2 * h + 1]);
            if ((C128_ABL & 2) && lo.x != 0x12345678u) continue;      // (keeps the arithmetic alive)
            p[4 * h] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
}

// ---- output projection + residual: s += a Wp^T ---------------------------------------------------------------------------
// (reference SegNet/nat.py:144-146: x = shortcut + drop_path(gamma1 * attn(...)); LayerScale folded into Wp by the host, the bias
// carried in the level's offset.)  HBM-bound: reads a and s, writes s — one wave per 16 tokens, Wp (32 KB) resident in LDS, the
// attention output's rows ARE the B operand (64 contiguous bytes per lane), the result is added to s in the store.  A GEMM
// library (or the 256 x 256-tile GEMM: N = 128 fills half a tile) needs 0.15 - 0.25 ms for the 1 M tokens of a batch; this is
// the three tensor passes.
constexpr int PROJ_LDS = C128 * 256;

__global__ __launch_bounds__(NAT128_THREADS, 1) void nat128_proj_add_kernel(__hip_bfloat16* __restrict__ s, const __hip_bfloat16* __restrict__ a,
                                                                            const __hip_bfloat16* __restrict__ w, long long groups) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    stage_k128(lds, w, C128, [](int r) { return out_channel(r >> 4, r & 15); });
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    // The rows of the NEXT group are requested before the current one is computed: a wave's load -> MFMA -> store chain would
    // otherwise leave the memory pipe idle for most of a group (16 waves x one group in flight is not enough to cover the latency).
    const long long step = (long long)gridDim.x * NAT128_WAVES;
    long long grp = (long long)blockIdx.x * NAT128_WAVES + wave;
    uint4 u[4], r[4];
    auto request = [&](long long gq, uint4 (&uu)[4], uint4 (&rr)[4]) {
        const uint4* src = reinterpret_cast<const uint4*>(a + (gq * 16 + n) * C128 + 8 * g);
        const uint4* row = reinterpret_cast<const uint4*>(s + (gq * 16 + n) * C128 + 8 * g);
#pragma unroll
        for (int k = 0; k < 4; ++k) { uu[k] = src[4 * k]; rr[k] = row[4 * k]; }     // rr[k]: channels 32 k + 8 g .. + 7
    };
    // Two register sets in turn (no copies: a register move of a loaded value waits for the load it was meant to overlap).
    auto body = [&](long long gq, uint4 (&uc)[4], uint4 (&rc)[4], uint4 (&un)[4], uint4 (&rn)[4]) {
        request(gq + step < groups ? gq + step : gq, un, rn);           // unconditional (the last one re-reads its own rows): behind a
                                                                      // branch the compiler's wait counts fall back to vmcnt(0) at the join
        asm volatile("" ::: "memory");      // the weight fragments are loop-invariant: hoisted out of the loop they spill (32 x 4 registers)
        f32x4 acc[8];                                                 // starts from the residual rows (their registers die here)
#pragma unroll
        for (int h = 0; h < 4; ++h) {                                 // 16-byte piece h: channels 32 h + 8 g .. + 7 = tiles 2 h, 2 h + 1
            acc[2 * h] = f32x4{bf_lo(rc[h].x), bf_hi(rc[h].x), bf_lo(rc[h].y), bf_hi(rc[h].y)};
            acc[2 * h + 1] = f32x4{bf_lo(rc[h].z), bf_hi(rc[h].z), bf_lo(rc[h].w), bf_hi(rc[h].w)};
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(lds, t * 16 + n, 4 * ks + g, 16), __builtin_bit_cast(bf16x8, uc[ks]), acc[t], 0, 0, 0);
            if (t & 1) __builtin_amdgcn_sched_barrier(0);            // 8 fragment reads in flight, not 32 (registers)
        }
        uint4* p = reinterpret_cast<uint4*>(s + (gq * 16 + n) * C128 + 8 * g);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const uint2 lo = pack4(acc[2 * h]), hi = pack4(acc[2 * h + 1]);
            p[4 * h] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    };
    uint4 u2[4], r2[4];
    if (grp < groups) request(grp, u, r);
    while (grp < groups) {
        body(grp, u, r, u2, r2);
        grp += step;
        if (grp >= groups) break;
        body(grp, u2, r2, u, r);
        grp += step;
    }
}

static int blocks_for(long long groups) {
    int cus = device_cu_count();
    if (!cus) cus = 256;
    const long long need = (groups + NAT128_WAVES - 1) / NAT128_WAVES;
    return (int)(need < cus ? need : cus);
}

int nat128_ln_qkv_launch(const void* s, const float* off, const void* lnw, const void* lnb, const void* w, const void* bias, void* qkv, long long tokens,
                         float eps, hipStream_t stream) {
    static DeviceOnce attr;
    if (const int e = dynamic_lds_once(attr, (const void*)nat128_ln_qkv_kernel, QKV_LDS)) return e;
    const long long groups = tokens / 16;
    hipLaunchKernelGGL(nat128_ln_qkv_kernel, dim3(blocks_for(groups)), dim3(NAT128_THREADS), QKV_LDS, stream, (const __hip_bfloat16*)s, off,
                       (const __hip_bfloat16*)lnw, (const __hip_bfloat16*)lnb, (const __hip_bfloat16*)w, (const __hip_bfloat16*)bias, (__hip_bfloat16*)qkv,
                       groups, eps);
    return (int)hipGetLastError();
}

int nat128_proj_add_launch(void* s, const void* a, const void* w, long long tokens, hipStream_t stream) {
    static DeviceOnce attr;
    if (const int e = dynamic_lds_once(attr, (const void*)nat128_proj_add_kernel, PROJ_LDS)) return e;
    const long long groups = tokens / 16;
    // the kernel is HBM-bound and small (32 KB of LDS): several workgroups per CU keep enough loads in flight
    int blocks = blocks_for(groups);
    const long long need = (groups + NAT128_WAVES - 1) / NAT128_WAVES;
    if (need > blocks) blocks = (int)(need < 2LL * blocks ? need : 2LL * blocks);
    hipLaunchKernelGGL(nat128_proj_add_kernel, dim3(blocks), dim3(NAT128_THREADS), PROJ_LDS, stream, (__hip_bfloat16*)s, (const __hip_bfloat16*)a,
                       (const __hip_bfloat16*)w, groups);
    return (int)hipGetLastError();
}

int nat128_ln_mlp_launch(void* s, const float* off, const void* lnw, const void* lnb, const void* w1, const void* b1, const void* w2, const float* add,
                         long long tokens, float eps, hipStream_t stream) {
    static DeviceOnce attr;
    if (const int e = dynamic_lds_once(attr, (const void*)nat128_ln_mlp_kernel, MLP_LDS)) return e;
    const long long groups = tokens / 16;
    hipLaunchKernelGGL(nat128_ln_mlp_kernel, dim3(blocks_for(groups)), dim3(NAT128_THREADS), MLP_LDS, stream, (__hip_bfloat16*)s, off,
                       (const __hip_bfloat16*)lnw, (const __hip_bfloat16*)lnb, (const __hip_bfloat16*)w1, (const __hip_bfloat16*)b1, (const __hip_bfloat16*)w2,
                       add, groups, eps);
    return (int)hipGetLastError();
}

}  // namespace ppn
