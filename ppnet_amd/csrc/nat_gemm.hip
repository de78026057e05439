// nat_gemm.hip — the dense half of a NAT layer at C = 256 / 512 / 1024 (DiNAT-B levels 1-3) on the matrix cores, as three
// persistent GEMM kernels whose epilogues carry everything between two projections (reference SegNet/nat.py:62-85 `Mlp`,
// :101-153 `NATLayer`: LN -> qkv -> NA -> proj (+ residual); LN -> fc1 -> GELU -> fc2 (+ residual)):
//
//   LN_BIAS        qkv = LN(s) W^T + b      computed WITHOUT a LayerNorm pass: with W' = W diag(gamma), b' = b + W beta,
//   LN_BIAS_GELU   h = gelu(LN(s) W^T + b)  LN(s) W^T + b = rstd (s W'^T - mean colsum(W')) + b'.  The GEMM reads the raw residual
//                                           stream s; mean / rstd of every row come from the row sums the accumulating kernel
//                                           below left behind.
//   ACC_STATS      s += a W^T + b           in place (LayerScale folded into W, b by the host).  The old s enters through the MATRIX
//                                           PIPE: the tile's 256 columns of s are four more k-tiles of the A operand against a
//                                           256 x 256 identity as the B operand ([a | s_tile] [W | I]^T — exact: bf16 x 1.0 summed
//                                           in float32), so it rides the same LDS-DMA ring as everything else: no register-
//                                           destination loads in the pipeline, no extra pass.  These GEMMs are HBM-bound (N = C),
//                                           the four k-tiles of MFMA work hide under their memory time.  The epilogue also emits
//                                           (sum, sum of squares) of every new row of s — of the bf16 values it stored, the values
//                                           the next GEMM reads — as one partial per 256-column tile, summed by the reader in a
//                                           fixed order (no atomics: bit-reproducible).
//
// Core: the 256 x 256 x 64 tile / 8-wave / LDS-DMA ring of mfma_gemm.h (same staging geometry, swizzle, fragment maps and
// 4-phase k-loop with the two wave groups staggered by a barrier).  What is new here:
//   * persistent (one workgroup per CU walks its tiles) with the epilogue of tile t spread over the phases of t's last k-tile and
//     t+1's first: quadrant q is converted and stored in the phase after its last MFMAs, in the segment where this wave group
//     only reads and the other group owns the matrix pipe.  The pipe never drains at a tile boundary and the store tail, the
//     epilogue's VALU work and the next tile's cold loads all hide behind MFMAs;
//   * vector-memory waits are counted at run time: every wave keeps the number of vector-memory instructions it has issued and a
//     mark per thing it will wait for (each phase's DMA, the per-tile vectors, the old-C loads); a wait is
//     `s_waitcnt vmcnt(issued - mark)` picked from a table of immediates.  Stores, DMA and loads share one in-order counter, so
//     this is exact whatever mix a phase issued — no hand-counted constants to get wrong when the schedule changes;
//   * the per-tile vectors (bias', colsum, row statistics) are staged into LDS by the same DMA path: there is no register-
//     destination global load inside the loop (hipcc would drain the whole DMA pipeline in front of its first use,
//     cdna_hip_programming.md section 5 trap (b); an inline-asm load's destination may be spilled before it lands).
// erf-GELU is evaluated as x * sigmoid(x (p0 + p1 x^2 + p2 x^4)) with the quintic fitted to erf (|error| < 3.0e-5 over the reals,
// far below the bf16 rounding of the result; tests/test_gpu_mfma.py): 9 VALU instructions instead of 17.
#pragma clang fp contract(fast)
#include "ppn_kernels.h"
#include <type_traits>
#include <vector>

namespace ppn {
namespace ngemm {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define PPN_INL __attribute__((always_inline))

constexpr int BM = 256, BN = 256, BK = 64, NTHREADS = 512;
// LDS: A tiles [2 parities][256 rows x 128 B] | B part 1 [2 parities][128 compact rows x 128 B] | B part 0 [3 slots][128 x 128 B] |
// per-tile vectors | (ACC) row partials.  B part 0 is read in phase 1 and again in phase 4 of its k-tile (its fragments do not stay
// in registers: 16 VGPRs), so its staging for k-tile g + 2 in phase 4 of k-tile g goes to a THIRD slot, not over the one being read.
constexpr int A_BYTES = 256 * 128, BH_BYTES = 128 * 128;
constexpr int A_OFF = 0, B1_OFF = 2 * A_BYTES, B0_OFF = B1_OFF + 2 * BH_BYTES, VEC_OFF = B0_OFF + 3 * BH_BYTES;     // 147 456
constexpr int MAX_P = 4;                                             // row-statistics partials per row (C / 256 <= 4)
constexpr int VEC_BYTES_LN = 2048 + MAX_P * 2048;                    // bias' f32[256] | colsum f32[256] | stats float2[P][256]
constexpr int VEC_BYTES_ACC = 1024;                                  // bias f32[256]
constexpr int RED_BYTES = 4 * 256 * 8;                               // ACC: float2 red[4 wave columns][256 rows]
constexpr int ID_KTILES = BN / BK;                                   // ACC: the identity segment's k-tiles
constexpr int ROWS_BYTES = 256 * 8;                                  // LN: float2 (rstd, -rstd * mean) of the tile's rows
__host__ __device__ constexpr int lds_bytes(bool acc) { return VEC_OFF + (acc ? VEC_BYTES_ACC + RED_BYTES : VEC_BYTES_LN + ROWS_BYTES); }   // 156 672 / 159 744 of 163 840

struct Params {
    const __bf16* A;        // [M][K] activations (LN modes: the raw residual stream)
    const __bf16* B;        // [N][K] weights (torch Linear layout; LN modes: W diag(gamma))
    __bf16* C;              // [M][N]
    const float* bias;      // [N]
    const float* colsum;    // [N] LN modes: sum_k B[n][k] (of the bf16 values)
    const float* stats_in;  // [P_in][M][2] LN modes: partial (sum, sum of squares) of every row of A
    float* stats_out;       // [N / 256][M][2] ACC mode: partials of the rows of the new C
    const __bf16* ident;    // [256][256] ACC mode: the identity matrix
    int M, N, K, P_in;
    float inv_k, eps;
};

template <int V> using I = std::integral_constant<int, V>;

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 0);
}
// the same with the non-temporal policy (aux = 2): for bytes this launch reads once (MI355X_MICROARCH.md, nt-weights row: issue ->
// landed ~18 % sooner for a once-read stream)
__device__ __forceinline__ void glds16_nt(const void* g, unsigned char* lds_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 2);
}

// s_waitcnt vmcnt(n'), n' = n rounded down to a multiple of 2 (conservative) and capped at 62; n is wave-uniform.  The steady state
// of the k-loop (four quarter-tiles in flight: 8) is tested first; everything else goes down a five-level tree of scalar branches.
__device__ __forceinline__ void wait_vm(int n) {
    if (n == 8) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); return; }
    const int h = n >= 62 ? 31 : (n < 0 ? 0 : n >> 1);
    if (h < 16) {
        if (h < 8) {
            if (h < 4) {
                if (h < 2) {
                    if (h < 1) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    }
                } else {
                    if (h < 3) {
                        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    }
                }
            } else {
                if (h < 6) {
                    if (h < 5) {
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                    }
                } else {
                    if (h < 7) {
                        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                    }
                }
            }
        } else {
            if (h < 12) {
                if (h < 10) {
                    if (h < 9) {
                        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                    }
                } else {
                    if (h < 11) {
                        asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
                    }
                }
            } else {
                if (h < 14) {
                    if (h < 13) {
                        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
                    }
                } else {
                    if (h < 15) {
                        asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
                    }
                }
            }
        }
    } else {
        if (h < 24) {
            if (h < 20) {
                if (h < 18) {
                    if (h < 17) {
                        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
                    }
                } else {
                    if (h < 19) {
                        asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
                    }
                }
            } else {
                if (h < 22) {
                    if (h < 21) {
                        asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(42)" ::: "memory");
                    }
                } else {
                    if (h < 23) {
                        asm volatile("s_waitcnt vmcnt(44)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(46)" ::: "memory");
                    }
                }
            }
        } else {
            if (h < 28) {
                if (h < 26) {
                    if (h < 25) {
                        asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(50)" ::: "memory");
                    }
                } else {
                    if (h < 27) {
                        asm volatile("s_waitcnt vmcnt(52)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(54)" ::: "memory");
                    }
                }
            } else {
                if (h < 30) {
                    if (h < 29) {
                        asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(58)" ::: "memory");
                    }
                } else {
                    if (h < 31) {
                        asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(62)" ::: "memory");
                    }
                }
            }
        }
    }
}

// erf-GELU through a logistic fit of erf: gelu(x) = x / (1 + exp(-x (p0 + p1 x^2 + p2 x^4))), x^2 clamped to 64 (beyond |x| = 8
// the result is x or 0 to 1e-14).  Coefficients times log2(e): the exponential is one v_exp_f32 (2^x).
__device__ __forceinline__ float gelu_logistic(float x) {
    const float x2 = fminf(x * x, 64.0f);
    const float t = x * (2.3009787f + x2 * (0.10690469f - 1.0350827e-3f * x2));
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-t));
}
// two at a time: the multiplies / adds as packed float32 instructions (v_pk_mul_f32, v_pk_fma_f32, v_pk_add_f32 process a register
// pair per issue slot; the epilogue is VALU-issue bound)
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_logistic2(f32x2 x) {
    f32x2 x2 = x * x;
    x2 = f32x2{fminf(x2.x, 64.0f), fminf(x2.y, 64.0f)};
    const f32x2 k0 = {2.3009787f, 2.3009787f}, k1 = {0.10690469f, 0.10690469f}, k2 = {-1.0350827e-3f, -1.0350827e-3f}, one = {1.0f, 1.0f};
    const f32x2 t = x * __builtin_elementwise_fma(x2, __builtin_elementwise_fma(x2, k2, k1), k0);
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(-t.x), __builtin_amdgcn_exp2f(-t.y)} + one;
    return x * f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}

// A 16-byte LDS read as four floats, through the same vector type as the MFMA fragments.  (hipcc's waitcnt pass puts
// `s_waitcnt vmcnt(0)` in front of a float4 / float2 LDS read while LDS-DMA is in flight — it cannot tell which bytes the DMA
// writes — and would drain the ring at every epilogue; it leaves reads of this type alone, as it does the fragment reads.)
__device__ __forceinline__ float4 lds_f4(const void* ptr) {
    const bf16x8 raw = *reinterpret_cast<const bf16x8*>(ptr);
    return __builtin_bit_cast(float4, raw);
}

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__device__ __forceinline__ float2 lds_f2(const void* ptr) {          // the 8-byte forms of the same
    const bf16x4 raw = *reinterpret_cast<const bf16x4*>(ptr);
    return __builtin_bit_cast(float2, raw);
}
__device__ __forceinline__ void lds_st_f2(void* ptr, float2 v) { *reinterpret_cast<bf16x4*>(ptr) = __builtin_bit_cast(bf16x4, v); }

struct TileSrc { int m0, n0; };      // a tile's origin (wave-uniform: lives in SGPRs)

template <bool LN, bool GELU, bool ACC>
__global__ __launch_bounds__(NTHREADS, 1) void nat_gemm_kernel(const Params p) {
#ifdef PPN_NG_NARROW
    constexpr bool WIDE = false;
#else
    constexpr bool WIDE = true;                                      // 16-byte stores (a lane-row exchange of the packed results) or 2 x 8-byte
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_n = p.N / BN, tiles_m = p.M / BM;
    const int nblk = tiles_m * tiles_n;
    const int nkw = p.K / BK;                                       // k-tiles of the weights
    const int nk = nkw + (ACC ? ID_KTILES : 0);                     // k-tiles of a tile: >= 3 (host)
    const int ldc = p.N;

    // ---- staging geometry.  A quarter-tile ("unit") = 128 rows of one operand = 16 blocks of 8 rows = 2 wave-instructions per wave;
    // lane l of a wave-instruction fills LDS bytes [16 l, 16 l + 16) of its block: row l >> 3, physical 16-byte chunk l & 7, which
    // holds logical chunk (l & 7) ^ ((LDS row >> 1) & 7) — the XOR swizzle, applied on the SOURCE address here and on the fragment
    // reads below.  The block a wave fills always has the wave's parity, so the key is ((wave & 1) * 4 + (l >> 4)) & 7 for every
    // unit: ONE per-lane source offset serves all of them, the rest of an address is wave-uniform.
    //   A unit u, pass j: block j * 16 + u * 8 + wave of the 256-row tile (rows with bit 6 == u: what a wave row reads as its part u)
    //   B unit u, pass j: compact block c = j * 8 + wave of the 128-row unit = tile columns (c >> 2) * 64 + u * 32 + (c & 3) * 8 + [0, 8)
    //                     (what a wave column reads as its part u: compact row wc * 32 + nt * 16 + r)
    const int srow = lane >> 3;
    const uint32_t lchunk = (uint32_t)(((lane & 7) ^ (((wave & 1) * 4 + (srow >> 1)) & 7)) * 16);        // bytes into the row
    const uint32_t lofs = (uint32_t)srow * (uint32_t)p.K * 2u + lchunk;
    // every global address below is (wave-uniform 64-bit base) + (32-bit per-lane offset): the SGPR-base addressing form, so that
    // no per-lane 64-bit pointers are formed (and hoisted, and spilled)
    const uint32_t lofs_c = ACC ? (uint32_t)srow * (uint32_t)p.N * 2u + lchunk : 0u;     // a row of C (stride N) + the chunk
    const uint32_t lofs_i = ACC ? (uint32_t)srow * (uint32_t)BN * 2u + lchunk : 0u;      // a row of the identity (stride 256)
    auto tile_src = [&](int v, TileSrc& t) PPN_INL {
        const int q = nblk >> 3, r = nblk & 7, x = v & 7;              // XCD-aware order: ids v, v + 8 share an XCD; each XCD a contiguous run
        const int id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (v >> 3);
        t.m0 = (id / tiles_n) * BM;
        t.n0 = (id % tiles_n) * BN;
    };
    const char* baseA = reinterpret_cast<const char*>(p.A);
    const char* baseB = reinterpret_cast<const char*>(p.B);
    int issued = 0;                                                  // vector-memory instructions this wave has issued (wave-uniform)
    // the two LDS-DMA instructions of a unit of k-tile kt of tile t.  op 0: A unit u -> A ring parity `slot`; op 1: B unit 1 -> B1
    // ring parity `slot`; op 2: B unit 0 -> B0 ring slot `slot`
    // ACC: k-tiles kt >= nkw are the identity segment: A = columns n0 + 64 (kt - nkw) .. of the tile's rows of C (row stride N),
    // B = columns 64 (kt - nkw) .. of the identity's rows [0, 256) (row stride 256)
    auto stage = [&](int op, int u, const TileSrc& t, int kt, int slot) PPN_INL {
        const bool idseg = ACC && kt >= nkw;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (op == 0) {
                const int blk = j * 16 + u * 8 + wave;
                const char* ub = !idseg ? baseA + ((size_t)(t.m0 + blk * 8) * p.K + kt * BK) * 2
                                        : reinterpret_cast<const char*>(p.C) + ((size_t)(t.m0 + blk * 8) * ldc + t.n0 + (kt - nkw) * BK) * 2;
#ifdef PPN_NG_NT_A
                if (ACC) glds16_nt(ub + (!idseg ? lofs : lofs_c), lds + A_OFF + slot * A_BYTES + blk * 1024);
                else
#endif
                glds16(ub + (!idseg ? lofs : lofs_c), lds + A_OFF + slot * A_BYTES + blk * 1024);
            } else {
                const int c = j * 8 + wave, unit = op == 1 ? 1 : 0;
                const int col = (c >> 2) * 64 + unit * 32 + (c & 3) * 8;
                const char* ub = !idseg ? baseB + ((size_t)(t.n0 + col) * p.K + kt * BK) * 2
                                        : reinterpret_cast<const char*>(p.ident) + ((size_t)col * BN + (kt - nkw) * BK) * 2;
                glds16(ub + (!idseg ? lofs : lofs_i), lds + (op == 1 ? B1_OFF : B0_OFF) + slot * BH_BYTES + c * 1024);
            }
        }
        issued += 2;
    };

    // ---- fragment reads (bytes within an A tile / a B unit)
    const int frow = lane & 15, fq = lane >> 4, fswz = frow >> 1;
    int f_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) f_rd[kk] = frow * 128 + (((kk * 4 + fq) ^ fswz) << 4);
    f32x4 acc[2][4][2][2];                                           // [A part][m tile][B part][n tile]
    bf16x8 fa[4][2], fb[2][2];                                       // the A part and the B part in use
    auto load_a = [&](int par, int part) PPN_INL {
        const unsigned char* base = lds + A_OFF + par * A_BYTES + (wr * 128 + part * 64) * 128;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[mt][kk] = *reinterpret_cast<const bf16x8*>(base + f_rd[kk] + mt * 16 * 128);
    };
    auto load_b = [&](const unsigned char* unit) PPN_INL {
        const unsigned char* base = unit + wc * 32 * 128;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fb[nt][kk] = *reinterpret_cast<const bf16x8*>(base + f_rd[kk] + nt * 16 * 128);
    };
    // zero: the quadrant's first MFMAs of a tile take a zero C operand (an inline constant) instead of reading the accumulators
    auto mma = [&](int ap, int bp, bool zero) PPN_INL {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const f32x4 c0 = (zero && kk == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[ap][mt][bp][nt];
                    acc[ap][mt][bp][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt][kk], fa[mt][kk], c0, 0, 0, 0);
                }
    };

    // ---- per-tile vectors in LDS
    const float* vec_bias = reinterpret_cast<const float*>(lds + VEC_OFF);
    const float* vec_csum = reinterpret_cast<const float*>(lds + VEC_OFF + 1024);
    const float2* vec_stat = reinterpret_cast<const float2*>(lds + VEC_OFF + 2048);
    float2* red = reinterpret_cast<float2*>(lds + VEC_OFF + VEC_BYTES_ACC);
    const int n_vec = LN ? 2 + 2 * p.P_in : 1;
    auto stage_vec = [&](const TileSrc& t) PPN_INL {                 // 1-KiB pieces dealt over the waves (<= 2 per wave); ACC: the bias only
        for (int pc = wave; pc < n_vec; pc += 8) {
            const char* src;
            if (pc == 0) src = reinterpret_cast<const char*>(p.bias + t.n0);
            else if (pc == 1) src = reinterpret_cast<const char*>(p.colsum + t.n0);
            else src = reinterpret_cast<const char*>(p.stats_in + ((size_t)((pc - 2) >> 1) * p.M + t.m0 + ((pc - 2) & 1) * 128) * 2);
            glds16(src + lane * 16, lds + VEC_OFF + pc * 1024);
            issued += 1;
        }
    };

    // ---- epilogue pieces
    const uint32_t lst = ((uint32_t)frow * (uint32_t)ldc + (uint32_t)fq * 4u) * 2u;                // this lane's store offset inside a quadrant's rows
    const uint32_t lst16 = ((uint32_t)frow * (uint32_t)ldc + (uint32_t)((fq & 1) * 16 + (fq >> 1) * 8)) * 2u;   // the same after the lane-row exchange
    // LN: (rstd, -rstd * mean) of the tile's 256 rows from the (sum, sum of squares) partials, once per tile: thread 2 r computes row
    // r — rows [0, 128) by waves 0-3, the wave group that reads them, [128, 256) by waves 4-7 — one phase before the first finish
    float2* vec_rows = reinterpret_cast<float2*>(lds + VEC_OFF + VEC_BYTES_LN);
    auto row_prepass = [&]() PPN_INL {
        const int row = tid >> 1;
        const float4 q0 = lds_f4(vec_stat + (row & ~1));
        float sx = (row & 1) ? q0.z : q0.x, sy = (row & 1) ? q0.w : q0.y;
        for (int pp = 1; pp < p.P_in; ++pp) {
            const float4 q = lds_f4(vec_stat + pp * 256 + (row & ~1));
            sx += (row & 1) ? q.z : q.x; sy += (row & 1) ? q.w : q.y;
        }
        const float mean = sx * p.inv_k;
        const float rstd = __builtin_amdgcn_rsqf(fmaxf(sy * p.inv_k - mean * mean, 0.f) + p.eps);
        if ((tid & 1) == 0) lds_st_f2(vec_rows + row, make_float2(rstd, -rstd * mean));
    };
    // sum over the four lanes that share a row (lanes l, l ^ 16, l ^ 32, l ^ 48) on the VALU: v_permlane16/32_swap of a value with
    // itself leaves (even rows, odd rows) / (lower half, upper half) copies whose sum is the xor-16 / xor-32 partner sum
    auto quad_sum = [&](float v) PPN_INL {
        auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
        t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(t[0]) + __uint_as_float(t[1]);
    };
    // finish quadrant (ap, bp) of the tile at (m0, n0): 4 x 16-byte stores per lane.  first_of_part: the first quadrant of its A
    // part to finish ((0,0) and (1,1)); the second one closes the part's row sums (ACC).  Every LDS read of the quadrant is issued
    // up front (one latency, not one per row: the epilogue sits on its wave group's critical path).
    auto finish = [&](int ap, int bp, int m0, int n0, bool first_of_part) PPN_INL {
#ifdef PPN_NG_NOFINISH
        {   // diagnostic build: keep the accumulators alive, do nothing else
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) { asm volatile("" :: "v"(acc[ap][mt][bp][0]), "v"(acc[ap][mt][bp][1])); }
            return;
        }
#endif
        const int nl = wc * 64 + bp * 32 + fq * 4;                    // this lane's columns: nl .. nl + 3 and nl + 16 .. nl + 19
        const int row0 = wr * 128 + ap * 64 + frow;
        const float4 b0 = lds_f4(vec_bias + nl), b1 = lds_f4(vec_bias + nl + 16);
        float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
        float2 rn[4], prev[4];
        if (LN) {
            c0 = lds_f4(vec_csum + nl); c1 = lds_f4(vec_csum + nl + 16);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) rn[mt] = lds_f2(vec_rows + row0 + mt * 16);
        }
        if (ACC && !first_of_part) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) prev[mt] = lds_f2(red + wc * 256 + row0 + mt * 16);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            // the lane's accumulators: 4 consecutive n of row frow in each of the quadrant's two 16-column tiles -> two 8-byte stores
            // (an exchange between lane rows would make one 16-byte store of them: 8 more VALU instructions per row, and the epilogue
            // is VALU-issue bound, not store-issue bound, once its stores trickle out beside the next tile's MFMAs)
            uint32_t w[4];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4 v = acc[ap][mt][bp][nt];
                const float4 b4 = nt == 0 ? b0 : b1;
                f32x2 o01 = {v[0], v[1]}, o23 = {v[2], v[3]};
                if (LN) {
                    const float4 c4 = nt == 0 ? c0 : c1;
                    const f32x2 r2 = {rn[mt].x, rn[mt].x}, n2 = {rn[mt].y, rn[mt].y};
                    o01 = __builtin_elementwise_fma(r2, o01, __builtin_elementwise_fma(n2, f32x2{c4.x, c4.y}, f32x2{b4.x, b4.y}));
                    o23 = __builtin_elementwise_fma(r2, o23, __builtin_elementwise_fma(n2, f32x2{c4.z, c4.w}, f32x2{b4.z, b4.w}));
                } else {
                    o01 += f32x2{b4.x, b4.y}; o23 += f32x2{b4.z, b4.w};
                }
                if (GELU) { o01 = gelu_logistic2(o01); o23 = gelu_logistic2(o23); }
                typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                const bf16x2 p0 = {(__bf16)o01.x, (__bf16)o01.y}, p1 = {(__bf16)o23.x, (__bf16)o23.y};
                w[2 * nt] = __builtin_bit_cast(uint32_t, p0); w[2 * nt + 1] = __builtin_bit_cast(uint32_t, p1);
                if (ACC) {                                            // sums of the ROUNDED values: what the next GEMM will read
                    const float f0 = __uint_as_float(w[2 * nt] << 16), f1 = __uint_as_float(w[2 * nt] & 0xffff0000u);
                    const float f2 = __uint_as_float(w[2 * nt + 1] << 16), f3 = __uint_as_float(w[2 * nt + 1] & 0xffff0000u);
                    s1 += (f0 + f1) + (f2 + f3); s2 += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
                }
            }
            char* ub = reinterpret_cast<char*>(p.C) + ((size_t)(m0 + wr * 128 + ap * 64 + mt * 16) * ldc + n0 + wc * 64 + bp * 32) * 2;
#ifndef PPN_NG_NOSTORE
            if (WIDE) {
                // v_permlane16_swap exchanges the PACKED pairs between neighbouring lane rows (2 instructions per row, on bf16 pairs —
                // not 8 on floats): lane row g ends with tile g & 1, columns 8 (g >> 1) .. + 7 of it -> one 16-byte store
                auto t0 = __builtin_amdgcn_permlane16_swap(w[0], w[2], false, false);
                auto t1 = __builtin_amdgcn_permlane16_swap(w[1], w[3], false, false);
                *reinterpret_cast<uint4*>(ub + lst16) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
            } else {
                *reinterpret_cast<uint2*>(ub + lst) = make_uint2(w[0], w[1]);
                *reinterpret_cast<uint2*>(ub + lst + 32) = make_uint2(w[2], w[3]);
            }
#else
            asm volatile("" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(ub + lst));
#endif
            if (ACC) {
                // over the quadrant's 32 columns of the row -> red[wc][row]: the part's first quadrant writes, its second one adds (the
                // same lane of the same wave a phase later: the LDS operations of a wave are in order and the sum order is fixed)
                s1 = quad_sum(s1); s2 = quad_sum(s2);
                if (!first_of_part) { s1 += prev[mt].x; s2 += prev[mt].y; }
                if (fq == 0) lds_st_f2(red + wc * 256 + row0 + mt * 16, make_float2(s1, s2));
            }
        }
#if !defined(PPN_NG_NOSTORE) && !defined(PPN_NG_NOFINISH)
        issued += WIDE ? 4 : 8;
#endif
    };
    // ACC: the row partials of the finished tile -> stats_out[tile column][row] (one float per thread, fixed summation order)
    auto store_stats = [&](int m0, int n0) PPN_INL {
        const int row = tid >> 1, st = tid & 1;
        const float2 r0 = lds_f2(red + row), r1 = lds_f2(red + 256 + row), r2 = lds_f2(red + 512 + row), r3 = lds_f2(red + 768 + row);
        const float v = st ? ((r0.y + r1.y) + r2.y) + r3.y : ((r0.x + r1.x) + r2.x) + r3.x;
        p.stats_out[((size_t)(n0 / BN) * p.M + m0) * 2 + tid] = v;
        issued += 1;
    };

    // ---- this workgroup's stream of k-tiles
    const int G = gridDim.x;
    const int my_tiles = (nblk - (int)blockIdx.x + G - 1) / G;
    const int total = my_tiles * nk;
    TileSrc cur, nxt;
    tile_src(blockIdx.x, cur);
    nxt = cur;
    if (my_tiles > 1) tile_src(blockIdx.x + G, nxt);
    int g = 0, s3 = 0;                                               // the current k-tile's index in the stream, and index mod 3 (its B0 slot)
    int mark_p[4];                                                   // `issued` right after the DMA of the latest phase 1 / 2 / 3 / 4
    int mark_vec = 0;
    // a unit of k-tile g + d (d = 1, 2) of the stream, seen from k-tile kt of tile `cur`: in this tile or at the head of the next
    auto stage_ahead = [&](int op, int u, int kt, int d, int slot) PPN_INL {
        if (g + d >= total) return;
        const int k = kt + d;
        if (k < nk) stage(op, u, cur, k, slot); else stage(op, u, nxt, k - nk, slot);
    };

#ifdef PPN_NG_SKEW
    for (int i = 0; i < (int)((blockIdx.x >> 3) & 7); ++i) __builtin_amdgcn_s_sleep(PPN_NG_SKEW);   // diagnostic: desynchronise the CUs' tile boundaries
#endif
    // prologue: the first tile's vectors; then k-tile 0 whole and k-tile 1's first halves
    stage_vec(cur);
    mark_vec = issued;
    stage(0, 0, cur, 0, 0); stage(2, 0, cur, 0, 0);
    stage(1, 1, cur, 0, 0); mark_p[0] = issued;
    stage(0, 1, cur, 0, 0); mark_p[1] = issued;
    const int mark_k0 = issued;
    stage(0, 0, cur, 1, 1); mark_p[2] = issued;
    stage(2, 0, cur, 1, 1); mark_p[3] = issued;
    wait_vm(issued - mark_k0);                                      // the vectors and k-tile 0 have landed
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                       // the stagger: this group's barriers pair with the other's next ones

#define PPN_NG_MMA(ap, bp, zero) do {                          \
        __builtin_amdgcn_s_barrier();                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_sched_barrier(0);                     \
        __builtin_amdgcn_s_setprio(1);                         \
        mma(ap, bp, zero);                                     \
        __builtin_amdgcn_s_setprio(0);                         \
        __builtin_amdgcn_s_barrier();                          \
    } while (0)

    // One k-tile = 4 phases, each {fragment reads + one quarter-tile of LDS-DMA + counted wait | barrier | 16 MFMA | barrier}; the
    // quadrants (A part, B part) in MFMA order: (0,0) (0,1) (1,1) (1,0).
    //   reads of k-tile g:  phase 1: A0, B0   phase 2: B1   phase 3: A1   phase 4: B0 again
    //   DMA issue:          phase 1: B1(g+1)  phase 2: A1(g+1)  phase 3: A0(g+2)  phase 4: B0(g+2) -> slot (g+2) % 3
    //   a unit is waited for one phase before its first read: the wait of phases 1, 2 and 4 retires the DMA of four phases ago.
    //   MODE 0: inside a tile.
    //   MODE 1: the stream's first k-tile (zero C operand).
    //   MODE 2: a tile's LAST k-tile: each quadrant is finished one phase after its MFMAs — (0,0) in phase 2, (0,1) in 3, (1,1) in 4.
    //   MODE 3: the first k-tile of the NEXT tile right behind a MODE 2 one: phase 1 finishes the previous tile's (1,0); zero C
    //           operand; the new tile's vectors are staged in phase 3, when every wave is done with the old ones.
    // (pm0, pn0): origin of the tile being finished (MODE 2: the current one; MODE 3: the previous one).
    auto ktile = [&](auto MODE_, int kt, int pm0, int pn0) PPN_INL {
        constexpr int MODE = decltype(MODE_)::value;
        constexpr bool ZERO = MODE == 1 || MODE == 3;
        const int par = g & 1;
        const unsigned char* b0 = lds + B0_OFF + s3 * BH_BYTES;
        const int s3n = s3 == 0 ? 2 : s3 - 1;                        // (g + 2) % 3
        // In every phase the DMA is issued (and its mark taken) BEFORE the epilogue's stores: the counter is in order, so a wait for
        // the DMA of four phases ago also waits for every store issued before that DMA, and stores drain slowly (a tile's 128 KB per
        // CU, every CU at once) — this way a store has five phases to complete before a wait reaches back to it, not four.
        // ---- phase 1
        load_b(b0);
        __builtin_amdgcn_sched_barrier(0);
        load_a(par, 0);
        stage_ahead(1, 1, kt, 1, par ^ 1);
        const int om0 = mark_p[0]; mark_p[0] = issued;
        if (MODE == 2 && LN) row_prepass();                         // the vectors are visible (waited for a phase ago); its output is read next phase
        if (MODE == 3) finish(1, 0, pm0, pn0, false);
        wait_vm(issued - om0);
        PPN_NG_MMA(0, 0, ZERO);
        // ---- phase 2
        load_b(lds + B1_OFF + par * BH_BYTES);
        stage_ahead(0, 1, kt, 1, par ^ 1);
        const int om1 = mark_p[1]; mark_p[1] = issued;
        if (MODE == 2) finish(0, 0, pm0, pn0, true);
        if (MODE == 3 && ACC) store_stats(pm0, pn0);
        wait_vm(issued - om1);
        PPN_NG_MMA(0, 1, ZERO);
        // ---- phase 3
        load_a(par, 1);
        stage_ahead(0, 0, kt, 2, par);
        mark_p[2] = issued;
        if (MODE == 2) finish(0, 1, pm0, pn0, false);
        if (MODE == 3) { stage_vec(cur); mark_vec = issued; }       // every wave has finished with the previous tile's vectors
        PPN_NG_MMA(1, 1, ZERO);
        // ---- phase 4
        load_b(b0);
        stage_ahead(2, 0, kt, 2, s3n);
        const int om3 = mark_p[3]; mark_p[3] = issued;
        if (MODE == 2) finish(1, 1, pm0, pn0, true);
        if (MODE != 2 && kt == nk - 2) wait_vm(issued - mark_vec);  // the tile's vectors: read from the next phase (its last k-tile's first) on
        wait_vm(issued - om3);
        PPN_NG_MMA(1, 0, ZERO);
        ++g;
        s3 = s3 == 2 ? 0 : s3 + 1;
    };

    ktile(I<1>{}, 0, 0, 0);
    for (int it = 0; it < my_tiles; ++it) {
        for (int kt = 1; kt < nk - 1; ++kt) ktile(I<0>{}, kt, 0, 0);
        const bool more = it + 1 < my_tiles;
        const int pm0 = cur.m0, pn0 = cur.n0;
        ktile(I<2>{}, nk - 1, pm0, pn0);
        if (more) {
            cur = nxt;
            if (it + 2 < my_tiles) tile_src(blockIdx.x + (it + 2) * G, nxt);
            ktile(I<3>{}, 0, pm0, pn0);
        } else {
            finish(1, 0, pm0, pn0, false);                           // the stream's last quadrant
            if (ACC) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                        // every wave of this group has written its row partials
                store_stats(pm0, pn0);
            }
        }
    }
#undef PPN_NG_MMA
    if (wr == 0) __builtin_amdgcn_s_barrier();                       // rebalance the barrier count of the two groups
}

// (sum, sum of squares) of every row of a bf16 [rows][C] tensor -> stats[rows][2]: the statistics of a level's first residual
// stream (the tokenizer's / downsampler's output), which no accumulating GEMM produced.  C / 8 lanes per row (at most a wave; C =
// 1024: two 16-byte pieces per lane), float32 sums of the bf16 values in a fixed order.
template <int PASSES>
__global__ __launch_bounds__(256) void row_stats_kernel(const __bf16* __restrict__ x, long long rows, int C, int lpr, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int rows_per_wave = 64 / lpr;
    const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * rows_per_wave + lane / lpr;
    const int li = lane % lpr;
    float s = 0.f, q = 0.f;
    if (row < rows) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            if ((ps * lpr + li) * 8 >= C) continue;                   // C / 8 need not be a power of two: the group's last lanes idle
            const uint4 u = *reinterpret_cast<const uint4*>(x + row * C + (ps * lpr + li) * 8);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float f0 = __uint_as_float(w[k] << 16), f1 = __uint_as_float(w[k] & 0xffff0000u);
                s += f0 + f1; q += f0 * f0 + f1 * f1;
            }
        }
    }
    for (int o = lpr >> 1; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (row < rows && li == 0) { stats[row * 2] = s; stats[row * 2 + 1] = q; }
}

template <bool LN, bool GELU, bool ACC>
int launch(const Params& p, int n_cu, hipStream_t stream) {
    static DeviceOnce attr;
    if (const int e = dynamic_lds_once(attr, (const void*)nat_gemm_kernel<LN, GELU, ACC>, lds_bytes(ACC))) return e;
    const int tiles = (p.M / BM) * (p.N / BN);
    int grid = tiles < n_cu ? tiles : n_cu;
    if (grid > 8) grid &= ~7;                                        // the XCD-aware tile order wants a multiple of 8
    hipLaunchKernelGGL((nat_gemm_kernel<LN, GELU, ACC>), dim3(grid), dim3(NTHREADS), lds_bytes(ACC), stream, p);
    return (int)hipGetLastError();
}

}  // namespace ngemm

namespace {
// the 256 x 256 bf16 identity the accumulating mode multiplies the old C by (128 KB of device memory, built once per device)
const void* identity_256() {
    static DeviceBuffer zb;
    const int dev = current_device();
    if (dev < 0) return nullptr;
    std::atomic<void*>& z = zb.p[dev];
    void* p = z.load();
    if (!p) {
        void* q = nullptr;
        std::vector<uint16_t> host(256 * 256, 0);
        for (int i = 0; i < 256; ++i) host[i * 256 + i] = 0x3F80;      // bf16 1.0
        if (hipMalloc(&q, host.size() * 2) != hipSuccess) return nullptr;
        if (hipMemcpy(q, host.data(), host.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(q); return nullptr; }
        void* expect = nullptr;
        if (!z.compare_exchange_strong(expect, q)) { (void)hipFree(q); q = expect; }
        p = q;
    }
    return p;
}
}  // namespace

// mode 0: C = LN-folded bias epilogue; 1: + GELU; 2: C += A W^T + bias in place, row partials out.  See include/ppnet_hip.h.
int nat_gemm_launch(const void* a, const void* w, const float* bias, const float* colsum, const float* stats_in, int p_in,
                    float* stats_out, void* c, long long M, int N, int K, int mode, float eps, hipStream_t stream) {
    ngemm::Params p{};
    p.A = (const __bf16*)a; p.B = (const __bf16*)w; p.C = (__bf16*)c; p.bias = bias; p.colsum = colsum;
    p.stats_in = stats_in; p.stats_out = stats_out; p.M = (int)M; p.N = N; p.K = K; p.P_in = p_in;
    p.inv_k = 1.0f / (float)K; p.eps = eps;
    if (mode == 2) {
        p.ident = (const __bf16*)identity_256();
        if (!p.ident) return (int)hipErrorOutOfMemory;
    }
    const int n_cu = device_cu_count();
    if (!n_cu) return -2;
    switch (mode) {
        case 0: return ngemm::launch<true, false, false>(p, n_cu, stream);
        case 1: return ngemm::launch<true, true, false>(p, n_cu, stream);
        case 2: return ngemm::launch<false, false, true>(p, n_cu, stream);
        default: return -1;
    }
}

int row_stats_launch(const void* x, long long rows, int C, float* stats, hipStream_t stream) {
    int lpr = 8;
    while (lpr < 64 && lpr * 8 < C) lpr *= 2;                        // lanes per row: a power of two, at most a wave
    const int passes = (C / 8 + lpr - 1) / lpr;
    const long long rows_per_block = 4 * (64 / lpr);
    const unsigned grid = (unsigned)((rows + rows_per_block - 1) / rows_per_block);
    if (passes == 1) hipLaunchKernelGGL((ngemm::row_stats_kernel<1>), dim3(grid), dim3(256), 0, stream, (const __bf16*)x, rows, C, lpr, stats);
    else if (passes == 2) hipLaunchKernelGGL((ngemm::row_stats_kernel<2>), dim3(grid), dim3(256), 0, stream, (const __bf16*)x, rows, C, lpr, stats);
    else return -1;
    return (int)hipGetLastError();
}

}  // namespace ppn
