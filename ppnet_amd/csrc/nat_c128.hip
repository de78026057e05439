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
// touches fall in 16 different bank groups.  Output channels are assigned to (tile, row) so that a lane ends up with 16
// CONSECUTIVE channels per four tiles: 32-byte stores, a full 128-byte line per token per four tiles.
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

// output channel of (tile t, row i) — see the header: a lane (token j, group g) holds rows 4g..4g+3 of every tile
__device__ __forceinline__ int out_channel(int t, int i) { return (t >> 2) * 64 + 16 * (i >> 2) + 4 * (t & 3) + (i & 3); }

__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// LayerNorm of 16 tokens in B-operand form.  Lane (j = lane & 15, g = lane >> 4) loads channels [32g, 32g + 32) of token j
// (64 contiguous bytes); MFMA k-slot (ks, g, e) is channel 32g + 8ks + e, the A operand uses the same map.
// ln = float32 [3][128] in LDS: weight, bias, offset.
__device__ __forceinline__ void ln_tokens(const __hip_bfloat16* __restrict__ s, long long tok0, int lane, const float* ln, float eps,
                                          bf16x8 (&yf)[4]) {
    const int j = lane & 15, g = lane >> 4;
    const uint4* src = reinterpret_cast<const uint4*>(s + (tok0 + j) * C128 + 32 * g);
    uint4 u[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = src[k];
    float x[32];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 o0 = *reinterpret_cast<const float4*>(ln + 256 + 32 * g + 8 * k), o1 = *reinterpret_cast<const float4*>(ln + 256 + 32 * g + 8 * k + 4);
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
        const float4 w0 = *reinterpret_cast<const float4*>(ln + 32 * g + 8 * k), w1 = *reinterpret_cast<const float4*>(ln + 32 * g + 8 * k + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(ln + 128 + 32 * g + 8 * k), b1 = *reinterpret_cast<const float4*>(ln + 128 + 32 * g + 8 * k + 4);
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) yf[k][e] = (__bf16)fmaf((x[8 * k + e] - mean) * rstd, wv[e], bv[e]);
    }
}

// Stage a [rows][128] bf16 weight as A-operand pieces: LDS row r holds source row src_row(r); its 16 chunks of 8 channels are
// stored at chunk position c ^ (r & 15).  (Fragment read of k-step ks by lane (n, g): chunk 4g + ks of row n.)
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
    for (long long grp = (long long)blockIdx.x * NAT128_WAVES + wave; grp < groups; grp += (long long)gridDim.x * NAT128_WAVES) {
        const long long tok0 = grp * 16;
        bf16x8 yf[4];
        ln_tokens(s, tok0, lane, ln, eps, yf);
        __hip_bfloat16* orow = qkv + (tok0 + n) * QKV_N + 16 * g;
#pragma unroll 1
        for (int qd = 0; qd < QKV_TILES / 4; ++qd) {                 // not unrolled: 16 fragment reads in flight are enough
            f32x4 acc[4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int t = qd * 4 + tt;
                acc[tt] = *reinterpret_cast<const f32x4*>(bl + t * 16 + 4 * g);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(wl, t * 16 + n, 4 * g + ks, 16), yf[ks], acc[tt], 0, 0, 0);
                if (tt & 1) __builtin_amdgcn_sched_barrier(0);
            }
            const uint2 p0 = pack4(acc[0]), p1 = pack4(acc[1]), p2 = pack4(acc[2]), p3 = pack4(acc[3]);
            uint4* dst = reinterpret_cast<uint4*>(orow + qd * 64);
            dst[0] = make_uint4(p0.x, p0.y, p1.x, p1.y);
            dst[1] = make_uint4(p2.x, p2.y, p3.x, p3.y);
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
    for (long long grp = (long long)blockIdx.x * NAT128_WAVES + wave; grp < groups; grp += (long long)gridDim.x * NAT128_WAVES) {
        const long long tok0 = grp * 16;
        bf16x8 yf[4];
        ln_tokens(s, tok0, lane, ln, eps, yf);
        f32x4 acc[8];                                                // starts from `add` (the level's accumulated biases, last layer) or 0
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = *reinterpret_cast<const f32x4*>(addl + t * 16 + 4 * g);
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
                    for (int ks = 0; ks < 4; ++ks) h[e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(w1l, t * 16 + n, 4 * g + ks, 16), yf[ks], h[e], 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) hf[k][e] = (__bf16)gelu_erf128(h[e >> 2][e & 3]);
                __builtin_amdgcn_sched_barrier(0);                   // keep the fragment reads of later steps from being hoisted (spills)
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(w2l, t * 16 + n, (half * 4 + k) * 4 + g, 32), hf[k], acc[t], 0, 0, 0);
                if (t & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __hip_bfloat16* row = s + (tok0 + n) * C128 + 16 * g;
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
            uint4* p = reinterpret_cast<uint4*>(row + qd * 64);
            const uint4 r0 = p[0], r1 = p[1];
            const f32x4 a0 = acc[qd * 4 + 0], a1 = acc[qd * 4 + 1], a2 = acc[qd * 4 + 2], a3 = acc[qd * 4 + 3];
            p[0] = make_uint4(pack_bf16x2(bf_lo(r0.x) + a0[0], bf_hi(r0.x) + a0[1]), pack_bf16x2(bf_lo(r0.y) + a0[2], bf_hi(r0.y) + a0[3]),
                              pack_bf16x2(bf_lo(r0.z) + a1[0], bf_hi(r0.z) + a1[1]), pack_bf16x2(bf_lo(r0.w) + a1[2], bf_hi(r0.w) + a1[3]));
            p[1] = make_uint4(pack_bf16x2(bf_lo(r1.x) + a2[0], bf_hi(r1.x) + a2[1]), pack_bf16x2(bf_lo(r1.y) + a2[2], bf_hi(r1.y) + a2[3]),
                              pack_bf16x2(bf_lo(r1.z) + a3[0], bf_hi(r1.z) + a3[1]), pack_bf16x2(bf_lo(r1.w) + a3[2], bf_hi(r1.w) + a3[3]));
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
    for (long long grp = (long long)blockIdx.x * NAT128_WAVES + wave; grp < groups; grp += (long long)gridDim.x * NAT128_WAVES) {
        const long long tok0 = grp * 16;
        const uint4* src = reinterpret_cast<const uint4*>(a + (tok0 + n) * C128 + 32 * g);
        __hip_bfloat16* row = s + (tok0 + n) * C128 + 16 * g;
        uint4 u[4], r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = src[k];
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) { r[2 * qd] = reinterpret_cast<const uint4*>(row + qd * 64)[0]; r[2 * qd + 1] = reinterpret_cast<const uint4*>(row + qd * 64)[1]; }
        f32x4 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(lds, t * 16 + n, 4 * g + ks, 16), __builtin_bit_cast(bf16x8, u[ks]), acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
            uint4* p = reinterpret_cast<uint4*>(row + qd * 64);
            const uint4 r0 = r[2 * qd], r1 = r[2 * qd + 1];
            const f32x4 a0 = acc[qd * 4 + 0], a1 = acc[qd * 4 + 1], a2 = acc[qd * 4 + 2], a3 = acc[qd * 4 + 3];
            p[0] = make_uint4(pack_bf16x2(bf_lo(r0.x) + a0[0], bf_hi(r0.x) + a0[1]), pack_bf16x2(bf_lo(r0.y) + a0[2], bf_hi(r0.y) + a0[3]),
                              pack_bf16x2(bf_lo(r0.z) + a1[0], bf_hi(r0.z) + a1[1]), pack_bf16x2(bf_lo(r0.w) + a1[2], bf_hi(r0.w) + a1[3]));
            p[1] = make_uint4(pack_bf16x2(bf_lo(r1.x) + a2[0], bf_hi(r1.x) + a2[1]), pack_bf16x2(bf_lo(r1.y) + a2[2], bf_hi(r1.y) + a2[3]),
                              pack_bf16x2(bf_lo(r1.z) + a3[0], bf_hi(r1.z) + a3[1]), pack_bf16x2(bf_lo(r1.w) + a3[2], bf_hi(r1.w) + a3[3]));
        }
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
