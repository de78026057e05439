// gennet_trunk.hip — the three ViT blocks of GenNet's AE-ViT as ONE kernel (reference GenNet/networks/ae_vit.py:38-42,68-70 and
// vit.py:88-111 Attention.forward, :127-133 Mlp.forward, :158-161 Block.forward; dim 24, 3 heads of 8, MLP x4, LayerNorm eps
// 1e-6, erf GELU, pre-norm residual blocks).
//
// The whole token set of a planning problem is N <= 1024 tokens x 24 channels: one 512-thread workgroup per problem keeps the
// residual stream in registers (float32, two tokens per thread) for all three blocks, K and V of the block in LDS (bfloat16;
// 96 KiB), and touches HBM twice: the 48-byte token rows in, the 48-byte token rows out.  Replaces, per batch, 3 library
// attention launches, 12 projection GEMMs, 6 LayerNorm launches and the residual adds between them.
//   LN1 -> qkv (weights are wave-uniform: scalar loads) -> K rows / V key-pairs to LDS, q packed in registers
//   attention per head: every K / V read is a broadcast (all lanes the same key) serving both of a thread's queries;
//       q.k on v_dot2_f32_bf16 over channel pairs, online softmax in base 2 (q carries scale * log2 e), rescale only when a
//       chunk of 8 keys raises a running maximum in the wave, P rounded to bfloat16 pairs, P.V on v_dot2 over key pairs
//   proj + residual, LN2, fc1 -> GELU -> fc2 streamed one hidden unit at a time, residual
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "ppn_device.h"
#include "ppn_kernels.h"
// The library is built with -ffp-contract=off because the generator kernels promise unfused IEEE double arithmetic (the parity
// contract with the oracle).  This file holds network arithmetic checked against float32 / float64 references to a tolerance:
// here a * b + c is one v_fma (otherwise every multiply-add of the LayerNorm, the GELU polynomial and the per-token linear
// phases is two instructions — these kernels are VALU-bound).
#pragma clang fp contract(fast)

namespace ppn {

namespace {
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int TC = 24, TH = 3, THD = 8, THID = 96;
// float32 parameter block of one ViT block, in this order
constexpr int O_LN1W = 0, O_LN1B = O_LN1W + TC, O_WQKV = O_LN1B + TC, O_BQKV = O_WQKV + 3 * TC * TC, O_WPROJ = O_BQKV + 3 * TC,
              O_BPROJ = O_WPROJ + TC * TC, O_LN2W = O_BPROJ + TC, O_LN2B = O_LN2W + TC, O_W1 = O_LN2B + TC, O_B1 = O_W1 + THID * TC,
              O_W2T = O_B1 + THID, O_B2 = O_W2T + THID * TC, BLOCK_PARAMS = O_B2 + TC;
static_assert(BLOCK_PARAMS == PPN_GENNET_BLOCK_PARAMS, "parameter block layout");

__device__ __forceinline__ float gelu_fast(float x) {                     // erf GELU, A&S 7.1.26 (|err| <= 1.5e-7), see mfma_gemm.h
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.0f - poly * __expf(-z * z);
    return 0.5f * x + 0.5f * fabsf(x) * e;
}

__device__ __forceinline__ void layer_norm24(const float (&x)[TC], const float* __restrict__ w, const float* __restrict__ b, float (&y)[TC]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < TC; ++c) s += x[c];
    const float mean = s * (1.0f / TC);
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < TC; ++c) { const float d = x[c] - mean; v += d * d; }
    const float rstd = rsqrtf(v * (1.0f / TC) + 1e-6f);
#pragma unroll
    for (int c = 0; c < TC; ++c) y[c] = (x[c] - mean) * rstd * w[c] + b[c];
}
}  // namespace

// x, y: [B][N][24] bfloat16 (NHWC feature map = token rows).  params: [n_blocks][BLOCK_PARAMS] float32.
__global__ __launch_bounds__(512) void gennet_trunk_kernel(const __bf16* __restrict__ xin, __bf16* __restrict__ yout,
                                                           const float* __restrict__ params, int N, int n_blocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tl[];
    // K rows: [N][24] bf16 (48 B per key, head h at +16 h);  V: [head][N/2 key pairs][8 channels] dwords {v[2j][c], v[2j+1][c]}
    unsigned char* Kl = tl;
    uint32_t* Vl = reinterpret_cast<uint32_t*>(tl + (size_t)N * 48);
    const int tid = threadIdx.x;
    const int prob = blockIdx.x;
    const int tok[2] = {tid, tid + 512};
    const bool live[2] = {tok[0] < N, tok[1] < N};
    const int half_n = N >> 1;

    float x[2][TC];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (live[s]) {
            const __bf16* src = xin + ((size_t)prob * N + tok[s]) * TC;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + 8 * p);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[s][8 * p + e] = (float)v[e];
            }
        } else {
#pragma unroll
            for (int c = 0; c < TC; ++c) x[s][c] = 0.f;
        }
    }

    const float qscale = 0.35355339059327373f * 1.4426950408889634f;      // head_dim^-0.5 * log2(e)
    for (int blk = 0; blk < n_blocks; ++blk) {
        const float* P = params + (size_t)blk * BLOCK_PARAMS;
        float y[2][TC];
        uint32_t qp[2][TH][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) layer_norm24(x[s], P + O_LN1W, P + O_LN1B, y[s]);
        if (blk > 0) __syncthreads();                                     // everyone is done reading the previous block's K / V
        // ---- qkv: 9 groups of 8 outputs (part, head); weights are uniform -> scalar loads
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
            for (int h = 0; h < TH; ++h) {
                float o8[2][THD];
#pragma unroll
                for (int c = 0; c < THD; ++c) {
                    const int row = part * TC + h * THD + c;
                    const float* wr = P + O_WQKV + row * TC;
                    float a0 = P[O_BQKV + row], a1 = a0;
#pragma unroll
                    for (int i = 0; i < TC; ++i) { a0 += wr[i] * y[0][i]; a1 += wr[i] * y[1][i]; }
                    o8[0][c] = a0; o8[1][c] = a1;
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (part == 0) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) qp[s][h][c] = pack_bf16x2(o8[s][2 * c] * qscale, o8[s][2 * c + 1] * qscale);
                    } else if (live[s]) {
                        if (part == 1) {
                            *reinterpret_cast<uint4*>(Kl + (size_t)tok[s] * 48 + h * 16) =
                                make_uint4(pack_bf16x2(o8[s][0], o8[s][1]), pack_bf16x2(o8[s][2], o8[s][3]), pack_bf16x2(o8[s][4], o8[s][5]),
                                           pack_bf16x2(o8[s][6], o8[s][7]));
                        } else {
                            uint16_t* vp = reinterpret_cast<uint16_t*>(Vl + ((size_t)h * half_n + (tok[s] >> 1)) * THD) + (tok[s] & 1);
#pragma unroll
                            for (int c = 0; c < THD; ++c) vp[2 * c] = (uint16_t)(pack_bf16x2(o8[s][c], 0.f) & 0xffffu);
                        }
                    }
                }
            }
        __syncthreads();

        // ---- attention: each thread's two queries against all N keys, head by head
        float att[2][TC];
#pragma unroll
        for (int h = 0; h < TH; ++h) {
            float m[2] = {-1e30f, -1e30f}, l[2] = {0.f, 0.f}, o[2][THD];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int c = 0; c < THD; ++c) o[s][c] = 0.f;
            const unsigned char* kh = Kl + h * 16;
            const uint32_t* vh = Vl + (size_t)h * half_n * THD;
            for (int k0 = 0; k0 < N; k0 += 8) {
                // all 16 reads of the chunk are issued before the first use (broadcast reads: every lane the same key)
                uint4 kv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) kv[i] = *reinterpret_cast<const uint4*>(kh + (size_t)(k0 + i) * 48);
                uint4 va[4], vb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t* vrow = vh + (size_t)((k0 >> 1) + j) * THD;
                    va[j] = *reinterpret_cast<const uint4*>(vrow); vb[j] = *reinterpret_cast<const uint4*>(vrow + 4);
                }
                float sc[2][8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        float a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, qp[s][h][0]), __builtin_bit_cast(bf2, kv[i].x), 0.f, false);
                        a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, qp[s][h][1]), __builtin_bit_cast(bf2, kv[i].y), a, false);
                        a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, qp[s][h][2]), __builtin_bit_cast(bf2, kv[i].z), a, false);
                        sc[s][i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, qp[s][h][3]), __builtin_bit_cast(bf2, kv[i].w), a, false);
                    }
                }
                bool grow = false;
                float cm[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    cm[s] = fmaxf(fmaxf(fmaxf(sc[s][0], sc[s][1]), fmaxf(sc[s][2], sc[s][3])), fmaxf(fmaxf(sc[s][4], sc[s][5]), fmaxf(sc[s][6], sc[s][7])));
                    grow = grow || cm[s] > m[s];
                }
                if (__any(grow)) {                                         // wave-uniform: rare after the first chunks
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const float mn = fmaxf(m[s], cm[s]);
                        const float corr = __builtin_amdgcn_exp2f(m[s] - mn);
                        m[s] = mn;
                        l[s] *= corr;
#pragma unroll
                        for (int c = 0; c < THD; ++c) o[s][c] *= corr;
                    }
                }
                uint32_t pp[2][4];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float pe[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) { pe[i] = __builtin_amdgcn_exp2f(sc[s][i] - m[s]); l[s] += pe[i]; }     // raw v_exp_f32: arguments <= 0
#pragma unroll
                    for (int j = 0; j < 4; ++j) pp[s][j] = pack_bf16x2(pe[2 * j], pe[2 * j + 1]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t vv[8] = {va[j].x, va[j].y, va[j].z, va[j].w, vb[j].x, vb[j].y, vb[j].z, vb[j].w};
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int c = 0; c < THD; ++c)
                            o[s][c] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, pp[s][j]), __builtin_bit_cast(bf2, vv[c]), o[s][c], false);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float inv = 1.0f / l[s];
#pragma unroll
                for (int c = 0; c < THD; ++c) att[s][h * THD + c] = o[s][c] * inv;
            }
        }

        // ---- proj + residual
#pragma unroll
        for (int oc = 0; oc < TC; ++oc) {
            const float* wr = P + O_WPROJ + oc * TC;
            float a0 = P[O_BPROJ + oc], a1 = a0;
#pragma unroll
            for (int i = 0; i < TC; ++i) { a0 += wr[i] * att[0][i]; a1 += wr[i] * att[1][i]; }
            x[0][oc] += a0; x[1][oc] += a1;
        }
        // ---- MLP: one hidden unit at a time, never materialised
#pragma unroll
        for (int s = 0; s < 2; ++s) layer_norm24(x[s], P + O_LN2W, P + O_LN2B, y[s]);
        float acc[2][TC];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < TC; ++c) acc[s][c] = P[O_B2 + c];
        for (int j = 0; j < THID; ++j) {
            const float* w1 = P + O_W1 + j * TC;
            const float* w2 = P + O_W2T + j * TC;
            float h0 = P[O_B1 + j], h1 = h0;
#pragma unroll
            for (int i = 0; i < TC; ++i) { h0 += w1[i] * y[0][i]; h1 += w1[i] * y[1][i]; }
            h0 = gelu_fast(h0); h1 = gelu_fast(h1);
#pragma unroll
            for (int c = 0; c < TC; ++c) { acc[0][c] += h0 * w2[c]; acc[1][c] += h1 * w2[c]; }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int c = 0; c < TC; ++c) x[s][c] += acc[s][c];
    }

#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (!live[s]) continue;
        __bf16* dst = yout + ((size_t)prob * N + tok[s]) * TC;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *reinterpret_cast<uint4*>(dst + 8 * p) = make_uint4(pack_bf16x2(x[s][8 * p], x[s][8 * p + 1]), pack_bf16x2(x[s][8 * p + 2], x[s][8 * p + 3]),
                                                                pack_bf16x2(x[s][8 * p + 4], x[s][8 * p + 5]), pack_bf16x2(x[s][8 * p + 6], x[s][8 * p + 7]));
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Matrix-core form of the same trunk.  v_dot2_f32_bf16 issues at a quarter of the VALU rate on gfx950, and the kernel above
// spends 16 of them per (two queries, key): 1.56 ms per 256-problem batch.  Here the attention products run on
// v_mfma_f32_16x16x32_bf16 with the head dimension (8) zero-padded to the instruction's K of 32 — lanes 16-63 hold zero
// fragments — which wastes three quarters of a pipe that is otherwise idle:
//   Q, K, V rows of the block in LDS (bf16, 48 B per token each; q carries scale * log2 e);
//   per (head, 16-query tile) of a wave:  pass 1  S^T = K Q^T tile by tile, only the row maximum is kept (4 max per tile);
//                                         pass 2  S^T again, p = 2^(s - max), P^T is the accumulator converted in place (k slot
//                                                 8g+j <-> key 4g+(j&3) of tile 2kp+(j>>2)), O^T += V^T P^T with V^T read by
//                                                 ds_read_b64_tr_b16 from the row-major V image (rows = channels h*8 .., 8 used)
//   exact softmax (two passes over the logits instead of an online rescale: the second S^T costs MFMA time only);
//   the attention output overwrites the tile's Q rows (bf16) for the per-token proj / MLP phase.
typedef __attribute__((ext_vector_type(4))) short ts16x4;
typedef __attribute__((ext_vector_type(8))) short ts16x8;
typedef __attribute__((ext_vector_type(4))) float tf32x4;

template <int NTHR>
__global__ __launch_bounds__(NTHR) void gennet_trunk_mfma_kernel(const __bf16* __restrict__ xin, __bf16* __restrict__ yout,
                                                                const float* __restrict__ params, int N, int n_blocks, float bound_max) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tl[];
    unsigned char* Ql = tl;                                                // [N][48 B]: q (scaled), later the attention output
    unsigned char* Kl = tl + (size_t)N * 48;
    unsigned char* Vl = tl + (size_t)N * 96;                               // + 64 B of slack behind it (transposed reads of head 2)
    float* Qn = reinterpret_cast<float*>(tl + (size_t)N * 144 + 64);       // [N][3]: |q| of every token and head (upper bound, see below)
    float* Kred = Qn + (size_t)N * 3;                                      // [NWAVES][3]: per-wave max |k|^2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int prob = blockIdx.x;
    constexpr int TPT = 1024 / NTHR, NWAVES = NTHR / 64;                  // tokens per thread (N <= 1024), waves
    int tok[TPT];
    bool live[TPT];
#pragma unroll
    for (int s = 0; s < TPT; ++s) { tok[s] = tid + s * NTHR; live[s] = tok[s] < N; }
    const int j = lane & 15, g = lane >> 4, q4 = j >> 2, p4 = j & 3;
    const int ntile = N >> 4;

    float x[TPT][TC];
#pragma unroll
    for (int s = 0; s < TPT; ++s) {
        if (live[s]) {
            const __bf16* src = xin + ((size_t)prob * N + tok[s]) * TC;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + 8 * p);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[s][8 * p + e] = (float)v[e];
            }
        } else {
#pragma unroll
            for (int c = 0; c < TC; ++c) x[s][c] = 0.f;
        }
    }
    if (tid < 16) reinterpret_cast<uint32_t*>(Vl + (size_t)N * 48)[tid] = 0u;     // the slack stays finite

    const float qscale = 0.35355339059327373f * 1.4426950408889634f;      // head_dim^-0.5 * log2(e)
    for (int blk = 0; blk < n_blocks; ++blk) {
        const float* P = params + (size_t)blk * BLOCK_PARAMS;
        float y[TPT][TC];
#pragma unroll
        for (int s = 0; s < TPT; ++s) layer_norm24(x[s], P + O_LN1W, P + O_LN1B, y[s]);
        // ---- qkv rows to LDS (the previous block's phase C read only this thread's own Q rows; K / V readers are past the
        // barrier that closed the attention phase)
        float k2max[TH] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
            for (int h = 0; h < TH; ++h) {
                float o8[TPT][THD];
#pragma unroll
                for (int c = 0; c < THD; ++c) {
                    const int row = part * TC + h * THD + c;
                    const float* wr = P + O_WQKV + row * TC;
                    float a[TPT];
#pragma unroll
                    for (int s = 0; s < TPT; ++s) a[s] = P[O_BQKV + row];
#pragma unroll
                    for (int i = 0; i < TC; ++i)
#pragma unroll
                        for (int s = 0; s < TPT; ++s) a[s] += wr[i] * y[s][i];
#pragma unroll
                    for (int s = 0; s < TPT; ++s) o8[s][c] = a[s];
                }
                unsigned char* base = part == 0 ? Ql : (part == 1 ? Kl : Vl);
                const float sc = part == 0 ? qscale : 1.0f;
                if (part < 2) {
                    // |q| (scaled) per token and the workgroup's max |k|^2 per head: with them |q . k| <= |q| max|k| bounds every
                    // logit of a query, which replaces the pass over all keys that only found the row maximum (the 1 % margin
                    // covers the bfloat16 rounding of the stored q and k: 2^-9 per component)
#pragma unroll
                    for (int s = 0; s < TPT; ++s) {
                        float n2 = 0.f;
#pragma unroll
                        for (int c = 0; c < THD; ++c) n2 += (o8[s][c] * sc) * (o8[s][c] * sc);
                        if (part == 0) { if (live[s]) Qn[(size_t)tok[s] * 3 + h] = sqrtf(n2) * 1.01f; }
                        else k2max[h] = fmaxf(k2max[h], live[s] ? n2 : 0.f);
                    }
                }
#pragma unroll
                for (int s = 0; s < TPT; ++s)
                    if (live[s])
                        *reinterpret_cast<uint4*>(base + (size_t)tok[s] * 48 + h * 16) =
                            make_uint4(pack_bf16x2(o8[s][0] * sc, o8[s][1] * sc), pack_bf16x2(o8[s][2] * sc, o8[s][3] * sc),
                                       pack_bf16x2(o8[s][4] * sc, o8[s][5] * sc), pack_bf16x2(o8[s][6] * sc, o8[s][7] * sc));
            }
#pragma unroll
        for (int h = 0; h < TH; ++h) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) k2max[h] = fmaxf(k2max[h], __shfl_xor(k2max[h], o, 64));
            if (lane == 0) Kred[wave * 3 + h] = k2max[h];
        }
        __syncthreads();
        float kmax[TH];
#pragma unroll
        for (int h = 0; h < TH; ++h) {
            float m2 = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) m2 = fmaxf(m2, Kred[w * 3 + h]);
            kmax[h] = sqrtf(m2) * 1.01f;
        }

        // ---- attention on the matrix cores: this wave's query tiles, head by head
        const bf16x8 zf = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int qt = wave; qt < ntile; qt += NWAVES) {
#pragma unroll 1
            for (int h = 0; h < TH; ++h) {
                const unsigned char* kh = Kl + (size_t)j * 48 + h * 16;
                bf16x8 qf = zf;
                if (g == 0) qf = *reinterpret_cast<const bf16x8*>(Ql + (size_t)(qt * 16 + j) * 48 + h * 16);
                // The softmax's stabiliser: any mx >= the row maximum works as long as exp2(s - mx) stays a normal number, and
                // |s| <= bound = |q| max|k| gives exp2(s - bound) in [2^(-2 bound), 1].  Up to bound = 40 (logits of +-28 in natural
                // units; bound_max) that is what is used — no pass over the keys; beyond it the wave takes the exact row maximum (pass 1).
                float mx = Qn[(size_t)(qt * 16 + j) * 3 + h] * kmax[h];
                if (__any(mx > bound_max)) {
                mx = -1.0e30f;
                for (int kt = 0; kt < ntile; kt += 4) {
                    bf16x8 kf[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { kf[u] = zf; if (g == 0) kf[u] = *reinterpret_cast<const bf16x8*>(kh + (size_t)(kt + u) * 16 * 48); }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const tf32x4 sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[u], qf, tf32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        mx = fmaxf(mx, fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])));
                    }
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                }
                // pass 2: probabilities and P V
                const tf32x4 nmx = {-mx, -mx, -mx, -mx};
                // the denominator: an all-ones A operand sums each query's (bfloat16-rounded) probabilities over the k-step's 32
                // keys into every row of its column — one more MFMA per step instead of eight adds per lane and two exchanges
                const bf16x8 onesf = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};
                tf32x4 lacc = {0.f, 0.f, 0.f, 0.f};
                tf32x4 oacc = {0.f, 0.f, 0.f, 0.f};
                const unsigned char* vb = Vl + (size_t)(4 * g + q4) * 48 + h * 16 + 8 * p4;
                for (int kp = 0; kp < ntile; kp += 4) {                    // two k-steps of 32 keys per iteration
                    bf16x8 kf[4];
                    ts16x4 vt[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        kf[u] = zf; if (g == 0) kf[u] = *reinterpret_cast<const bf16x8*>(kh + (size_t)(kp + u) * 16 * 48);
                        vt[u] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ts16x4*)(vb + (size_t)(kp + u) * 16 * 48));
                    }
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        // the row maximum rides in as the initial accumulator (this lane's four rows all belong to query j):
                        // S' = K Q^T - max leaves the matrix pipe ready for exp2, no subtraction per logit
                        tf32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[2 * half], qf, nmx, 0, 0, 0);
                        tf32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[2 * half + 1], qf, nmx, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { s0[r] = __builtin_amdgcn_exp2f(s0[r]); s1[r] = __builtin_amdgcn_exp2f(s1[r]); }
                        const bf16x8 pf = {(__bf16)s0[0], (__bf16)s0[1], (__bf16)s0[2], (__bf16)s0[3], (__bf16)s1[0], (__bf16)s1[1], (__bf16)s1[2], (__bf16)s1[3]};
                        const ts16x4 lo = vt[2 * half], hi = vt[2 * half + 1];
                        const ts16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        oacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vv), pf, oacc, 0, 0, 0);
                        lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(onesf, pf, lacc, 0, 0, 0);     // row sums on the matrix pipe
                    }
                }
                const float inv = 1.0f / lacc[0];
                // O^T rows = channels 4g + r of this head (g < 2), column = query j: 8 bytes over the query's own q row
                if (g < 2)
                    *reinterpret_cast<uint2*>(Ql + (size_t)(qt * 16 + j) * 48 + h * 16 + g * 8) =
                        make_uint2(pack_bf16x2(oacc[0] * inv, oacc[1] * inv), pack_bf16x2(oacc[2] * inv, oacc[3] * inv));
            }
        }
        __syncthreads();

        // ---- proj + residual, from the attention rows in LDS
        float att[TPT][TC];
#pragma unroll
        for (int s = 0; s < TPT; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const bf16x8 v = live[s] ? *reinterpret_cast<const bf16x8*>(Ql + (size_t)tok[s] * 48 + p * 16) : zf;
#pragma unroll
                for (int e = 0; e < 8; ++e) att[s][8 * p + e] = (float)v[e];
            }
#pragma unroll
        for (int oc = 0; oc < TC; ++oc) {
            const float* wr = P + O_WPROJ + oc * TC;
            float a[TPT];
#pragma unroll
            for (int s = 0; s < TPT; ++s) a[s] = P[O_BPROJ + oc];
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int s = 0; s < TPT; ++s) a[s] += wr[i] * att[s][i];
#pragma unroll
            for (int s = 0; s < TPT; ++s) x[s][oc] += a[s];
        }
        // ---- MLP: one hidden unit at a time, never materialised
#pragma unroll
        for (int s = 0; s < TPT; ++s) layer_norm24(x[s], P + O_LN2W, P + O_LN2B, y[s]);
        float acc[TPT][TC];
#pragma unroll
        for (int s = 0; s < TPT; ++s)
#pragma unroll
            for (int c = 0; c < TC; ++c) acc[s][c] = P[O_B2 + c];
        for (int jh = 0; jh < THID; ++jh) {
            const float* w1 = P + O_W1 + jh * TC;
            const float* w2 = P + O_W2T + jh * TC;
            float hh[TPT];
#pragma unroll
            for (int s = 0; s < TPT; ++s) hh[s] = P[O_B1 + jh];
#pragma unroll
            for (int i = 0; i < TC; ++i)
#pragma unroll
                for (int s = 0; s < TPT; ++s) hh[s] += w1[i] * y[s][i];
#pragma unroll
            for (int s = 0; s < TPT; ++s) hh[s] = gelu_fast(hh[s]);
#pragma unroll
            for (int c = 0; c < TC; ++c)
#pragma unroll
                for (int s = 0; s < TPT; ++s) acc[s][c] += hh[s] * w2[c];
        }
#pragma unroll
        for (int s = 0; s < TPT; ++s)
#pragma unroll
            for (int c = 0; c < TC; ++c) x[s][c] += acc[s][c];
    }

#pragma unroll
    for (int s = 0; s < TPT; ++s) {
        if (!live[s]) continue;
        __bf16* dst = yout + ((size_t)prob * N + tok[s]) * TC;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *reinterpret_cast<uint4*>(dst + 8 * p) = make_uint4(pack_bf16x2(x[s][8 * p], x[s][8 * p + 1]), pack_bf16x2(x[s][8 * p + 2], x[s][8 * p + 3]),
                                                                pack_bf16x2(x[s][8 * p + 4], x[s][8 * p + 5]), pack_bf16x2(x[s][8 * p + 6], x[s][8 * p + 7]));
    }
}

int gennet_trunk_launch(const void* x, void* y, const float* params, int B, int N, int n_blocks, hipStream_t stream) {
    // the matrix-core form needs whole groups of 4 key tiles (N % 64 == 0: 1024 tokens at R = 256 / 512 / 64); PPNET_TRUNK_VALU=1
    // keeps the v_dot2 kernel (A/B runs), which also serves the other token counts (784 at R = 224)
    static const bool valu = getenv("PPNET_TRUNK_VALU") != nullptr;
    // 1024 threads (one token per thread, 4 waves per SIMD) unless PPNET_TRUNK_512=1 keeps the first form (two tokens per thread,
    // 2 waves per SIMD): the kernel is a chain of MFMA -> exp -> MFMA and scalar-load -> FMA latencies, and there is exactly one
    // workgroup per CU at batch 256, so the only latency hiding there is comes from the waves of that workgroup
    static const bool t512 = getenv("PPNET_TRUNK_512") != nullptr;
    if (!valu && N % 64 == 0) {
        const size_t lds_m = (size_t)N * 144 + 64 + (size_t)N * 12 + 16 * 3 * 4;      // Q, K, V rows + slack, |q| per (token, head), per-wave max |k|^2
        static DeviceOnce attr_512, attr_1024;
        if (const int e = dynamic_lds_once(attr_512, (const void*)gennet_trunk_mfma_kernel<512>, 156 * 1024 + 64 + 192)) return e;
        if (const int e = dynamic_lds_once(attr_1024, (const void*)gennet_trunk_mfma_kernel<1024>, 156 * 1024 + 64 + 192)) return e;
        // PPNET_TRUNK_EXACT_MAX=1: always take the exact row maximum (the two-pass softmax) instead of the norm bound — A/B and tests
        const float bound_max = getenv("PPNET_TRUNK_EXACT_MAX") ? -1.0f : 40.0f;
        if (t512) hipLaunchKernelGGL(gennet_trunk_mfma_kernel<512>, dim3(B), dim3(512), lds_m, stream, (const __bf16*)x, (__bf16*)y, params, N, n_blocks, bound_max);
        else hipLaunchKernelGGL(gennet_trunk_mfma_kernel<1024>, dim3(B), dim3(1024), lds_m, stream, (const __bf16*)x, (__bf16*)y, params, N, n_blocks, bound_max);
        return (int)hipGetLastError();
    }
    const size_t lds = (size_t)N * 48 + (size_t)3 * (N / 2) * 8 * 4;
    static DeviceOnce attr;
    if (const int e = dynamic_lds_once(attr, (const void*)gennet_trunk_kernel, 96 * 1024)) return e;
    hipLaunchKernelGGL(gennet_trunk_kernel, dim3(B), dim3(512), lds, stream, (const __bf16*)x, (__bf16*)y, params, N, n_blocks);
    return (int)hipGetLastError();
}

}  // namespace ppn
