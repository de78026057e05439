// mfma_gemm.h — bf16 MFMA GEMM core for gfx950 (CDNA4), shared by the dense projections of SegNet's NAT layers
// (reference SegNet/nat.py:62-85,111-120: qkv / proj / fc1 / fc2) and the implicit-GEMM 3x3 convolutions of the SETR-UP head
// (reference SegNet/mmseg/decode_heads/setr_up_head.py:53-66).
//
//   C[m][n] = epilogue( sum_k A[m][k] * B[n][k] )       A: activations, K contiguous;  B: weights [N][K] (torch Linear layout)
//
// Block tile 256 x 256 x 64, 512 threads = 8 waves as 2 (M) x 4 (N), each wave a 128 x 64 output in 32 accumulators of
// v_mfma_f32_16x16x32_bf16 (weights as the MFMA's A operand, activations as its B operand, so a lane holds 4 consecutive
// n of one row m).  Both operand tiles go global -> LDS by 16-byte LDS-DMA (global_load_lds_dwordx4): the LDS image is
// lane-linear per wave-instruction (8 rows x 128 B), the XOR swizzle chunk ^= (row >> 1) & 7 is applied on the SOURCE
// address and again on the fragment read (conflict-free ds_read_b128 for the 16x16x32 operand maps).  Double-buffered
// LDS (2 x 64 KiB); the K loop runs 4 phases per K-tile (one 64 x 32 output quadrant of every wave each), one
// quarter-tile is issued per phase and only counted vmcnt waits leave four of them in flight across the barriers; the
// two waves of a SIMD run staggered by one barrier (cdna_hip_programming.md, "The 256^2 8-phase template").
//
// A modes:  DENSE   A is [M][lda] row-major;
//           CONV3   A is an NHWC image [B][H][W][Cin]; row m = output pixel (stride 1 or 2, padding 1), k = (ky*3 + kx)*Cin + ci;
//                   out-of-image taps read a zero line.  Cin % 64 == 0.  B is then [N][9*Cin] with the same k order.
//                   The k-tiles are WALKED channel block by channel block (all 9 taps of 64 channels, then the next 64): the taps of
//                   a block re-read the same pixels' 128 bytes shifted by a pixel or a row, so the tiles an XCD has in flight touch
//                   ~1 MB per 9 k-tiles and the re-reads hit its L2; tap-major (the memory order of k) the nine visits of a pixel lie
//                   Cin / 64 k-tiles apart with 4 MB of other rows in between, and every visit missed the L2 (a third of all requests: ≈12.8 GB fetched for
//                   the 1 GB image of the SETR-UP head's last stage).
// Epilogues: see Epi.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ppn {
namespace gemm {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 256, BN = 256, BK = 64, NTHREADS = 512;
constexpr int TILE_BYTES = 256 * 128;          // one operand tile: 256 rows x 64 bf16
constexpr int BUF_BYTES = 2 * TILE_BYTES;      // A + B of one K-tile
constexpr int LDS_BYTES = 2 * BUF_BYTES;       // 128 KiB: double buffer

enum AMode { DENSE = 0, CONV3 = 1 };
enum Epi {
    EPI_BIAS = 0,        // C = acc + bias[n]
    EPI_BIAS_GELU = 1,   // C = gelu(acc + bias[n])              (erf form, torch.nn.GELU default)
    EPI_ACCUM = 2,       // C = C + acc                           (residual stream update, bias carried outside)
    EPI_BIAS_RELU = 3,   // C = max(acc + bias[n], 0)             (conv + folded BatchNorm + ReLU)
    EPI_RELU_DOT2 = 4,   // partial[slot][m][c] = sum_{n in slot} max(acc + bias[n], 0) * w2[c][n], c = 0, 1, slot = a 256-column block
                         // (the 1x1 classifier on top; a block's four wave columns are added in order inside the workgroup, the slots in
                         // order by classify2_reduce_kernel: no atomics.  N <= 256: one slot, added straight onto the logits)
    EPI_ACCUM_STATS = 5, // C = C + acc + bias[n] in place, and (sum, sum of squares) of every row of the NEW C (of the bf16 values stored)
                         // per 128 or 256 columns -> stats[partial][m][2]: the accumulating projections of a NAT layer (proj, fc2:
                         // SegNet/nat.py:145-153), whose row statistics the next LayerNorm-folded projection reads (nat_gemm.hip).
                         // Persistent launches only.  Round 5: the old C is READ IN THE EPILOGUE — round 3-4's nat_gemm.hip fed it
                         // through the matrix pipe as four identity k-tiles (+25 ... +100 % MFMA and LDS work at K = 1024 ... 256)
    EPI_LN_BIAS = 6,     // C = LN(A) W^T + b WITHOUT a LayerNorm pass (nat_gemm.hip's algebra): with B = W diag(gamma), bias = b + W beta,
    EPI_LN_BIAS_GELU = 7 // colsum[n] = sum_k B[n][k]:  C = rstd_m (acc - mean_m colsum[n]) + bias[n]  [then erf-GELU, logistic fit, |err| < 3e-5];
                         // mean / rstd of row m of A from the partial (sum, sum of squares) in stats[P][M][2] (EPI_ACCUM_STATS of the
                         // projection in front, the fused MLP kernel, or ppn_row_stats_bf16).  Persistent launches only.
};
__host__ __device__ constexpr int lds_bytes(int epi) { return LDS_BYTES + (epi == EPI_ACCUM_STATS ? 4 * BM * 8 : 0); }   // + float2 red[4 wave columns][256 rows]

struct Params {
    const __bf16* A;
    const __bf16* B;
    __bf16* C;
    const float* bias;     // [N] float32 (may be null for EPI_ACCUM)
    int M, N, K;
    int lda, ldc;          // elements
    // CONV3: input image [B][H][W][Cin], output pixels [B][Ho][Wo] (row m), stride 1 or 2, padding 1
    int H, W, Cin, Ho, Wo, stride;
    const __bf16* zero;    // >= 128 bytes of zeros
    // EPI_RELU_DOT2
    const float* w2;       // [2][N]
    float* logits;         // EPI_RELU_DOT2: N > 256: the partial sums [ceil(N / 256)][M][2]; else the logits [M][2] themselves (+=)
    // EPI_ACCUM_STATS
    float* stats;          // [N / 128 or N / 256][M][2]
    int stats_p128;        // 1: one partial per 128 columns (streams of width 256: what the fused MLP kernel emits too), 0: per 256
    // EPI_LN_BIAS / EPI_LN_BIAS_GELU: `stats` is the INPUT [stats_parts][M][2]
    const float* colsum;   // [N]
    int stats_parts;       // 1..4
    float inv_k, eps;
};

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 0);
}

// erf-form GELU (torch.nn.GELU default).  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of
// the result): one v_exp, one v_rcp and six FMAs instead of libm's ~40-instruction erff — the epilogue holds 128 values per lane.
// 8-byte LDS accesses through the fragment element type: hipcc's waitcnt pass puts `s_waitcnt vmcnt(0)` in front of float2 / float4-
// typed LDS accesses while an LDS-DMA is in flight (it cannot tell which bytes the DMA writes) and leaves these alone (nat_gemm.hip)
__device__ __forceinline__ float2 lds_f2(const void* ptr) { return __builtin_bit_cast(float2, *reinterpret_cast<const bf16x4*>(ptr)); }
__device__ __forceinline__ void lds_st_f2(void* ptr, float2 v) { *reinterpret_cast<bf16x4*>(ptr) = __builtin_bit_cast(bf16x4, v); }

__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.0f - poly * __expf(-z * z);                     // erf(|x| / sqrt 2)
    return 0.5f * x + 0.5f * fabsf(x) * e;                            // 0.5 x (1 + sign(x) e)
}

// erf-GELU through a logistic fit of erf (nat_gemm.hip): x / (1 + 2^(-x (p0 + p1 x^2 + p2 x^4))), |error| < 3.0e-5
__device__ __forceinline__ float gelu_logistic(float x) {
    const float x2 = fminf(x * x, 64.0f);
    const float t = x * (2.3009787f + x2 * (0.10690469f - 1.0350827e-3f * x2));
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-t));
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_logistic2(f32x2 x) {               // two at a time: the multiplies / adds packed
    f32x2 x2 = x * x;
    x2 = f32x2{fminf(x2.x, 64.0f), fminf(x2.y, 64.0f)};
    const f32x2 k0 = {2.3009787f, 2.3009787f}, k1 = {0.10690469f, 0.10690469f}, k2 = {-1.0350827e-3f, -1.0350827e-3f}, one = {1.0f, 1.0f};
    const f32x2 t = x * __builtin_elementwise_fma(x2, __builtin_elementwise_fma(x2, k2, k1), k0);
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(-t.x), __builtin_amdgcn_exp2f(-t.y)} + one;
    return x * f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}

// sum over the four lanes that share a row of the MFMA layout (lanes l, l ^ 16, l ^ 32, l ^ 48) on the vector ALU: v_permlane16/32_swap of a
// value with itself leaves (even rows, odd rows) / (lower half, upper half) copies whose sum is the xor-16 / xor-32 partner sum — no
// trip through the LDS crossbar (ds_bpermute) and no wait (nat_gemm.hip)
__device__ __forceinline__ float quad_sum(float v) {
    auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(t[0]) + __uint_as_float(t[1]);
    t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}

// What one output tile needs from this lane: source pointers of its 16-byte pieces of the four quarter-tiles.
struct TileSrc {
    uint32_t a[2][2];           // [unit][pass]: BYTE offset from p.A at k-tile 0 (CONV3: the centre tap of the pixel); < 2^32
    uint32_t b[2][2];           // byte offset from p.B
    uint32_t valid[2];          // CONV3: 9-bit masks of in-image taps, [unit], pass 0 in bits 0-8, pass 1 in bits 16-24
    int m0, n0;
};

// Persistent form: `gridDim.x` blocks (one per CU) walk the tiles v = blockIdx.x, blockIdx.x + gridDim.x, ... and the
// LDS-DMA stream runs across tile boundaries — the first two k-tiles of the next tile are in flight while the current
// tile's epilogue stores go out, so neither the cold first loads nor the store tail idle the matrix pipe.  The epilogue's
// stores sit in the same in-order vmcnt queue as the DMA: the three counted waits that follow it are widened by the number
// of store instructions (EPI_STORES), which requires every lane's stores to issue — launch with gridDim.x == number of
// tiles (one tile per block, nothing follows an epilogue) when M or N is not a multiple of 256.
template <int AMODE, int EPI>
__global__ __launch_bounds__(NTHREADS, 1) void gemm_bf16_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nblk = tiles_m * tiles_n;
    const int nk = p.K / BK;

    // ---- staging geometry: a quarter-tile ("unit") = 128 rows of one operand = 16 row-blocks of 8 rows, 2 passes x 8 waves.
    // A unit u = the rows every wave row reads as its part u (rows with bit 6 == u); B unit u = the columns every wave column
    // reads as its part u (cols with bit 5 == u).  Lane l of a wave-instruction fills LDS bytes [16 l, 16 l + 16) of its
    // row-block: row l >> 3, physical chunk l & 7, which holds logical chunk (l & 7) ^ ((row >> 1) & 7).
    const int srow = lane >> 3;
    auto a_block = [&](int u, int j) { return j * 16 + u * 8 + wave; };                       // A row-block: bit 3 == u
    auto b_block = [&](int u, int j) { const int x4 = j * 8 + wave; return (x4 >> 2) * 8 + u * 4 + (x4 & 3); };   // bit 2 == u
    // tile v of the launch order -> (m0, n0), XCD-aware: virtual ids v and v + 8 run on one XCD (gridDim.x % 8 == 0 or one tile
    // per block); each XCD gets a contiguous run of tiles, n-tiles of one m-tile adjacent (A panel and weights from its L2)
    auto tile_src = [&](int v, TileSrc& t) {
        const int q = nblk >> 3, r = nblk & 7, x = v & 7;
        const int id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (v >> 3);
        t.m0 = (id / tiles_n) * BM;
        t.n0 = (id % tiles_n) * BN;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            t.valid[u] = 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ra = a_block(u, j) * 8 + srow;
                const int ca = ((lane & 7) ^ ((ra >> 1) & 7)) * 8;
                int gm = t.m0 + ra;
                if (gm >= p.M) gm = p.M - 1;                       // clamp: rows past M are computed and never stored
                if (AMODE == DENSE) {
                    t.a[u][j] = ((uint32_t)gm * (uint32_t)p.lda + ca) * 2u;
                } else {
                    const int xo = gm % p.Wo, yo = (gm / p.Wo) % p.Ho, img = gm / (p.Wo * p.Ho);
                    const int x0 = xo * p.stride, y0 = yo * p.stride;                    // centre tap in the input image
                    t.a[u][j] = ((uint32_t)((img * p.H + y0) * p.W + x0) * (uint32_t)p.Cin + ca) * 2u;
                    uint32_t vm = 0;
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) {
                        const int yy = y0 + tp / 3 - 1, xx = x0 + tp % 3 - 1;
                        if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) vm |= 1u << tp;
                    }
                    t.valid[u] |= vm << (16 * j);
                }
                const int rb = b_block(u, j) * 8 + srow;
                const int cb = ((lane & 7) ^ ((rb >> 1) & 7)) * 8;
                int gn = t.n0 + rb;
                if (gn >= p.N) gn = p.N - 1;
                t.b[u][j] = ((uint32_t)gn * (uint32_t)p.K + cb) * 2u;
            }
        }
    };
    const __bf16* zero_src = (AMODE == CONV3) ? p.zero + (lane & 7) * 8 : nullptr;

    // op: 0 = A, 1 = B.  Issues the two LDS-DMA instructions of unit `u` of k-tile `kt` of tile `t` into buffer `par`.
    const char* baseA = reinterpret_cast<const char*>(p.A);
    const char* baseB = reinterpret_cast<const char*>(p.B);
    auto stage = [&](int op, int u, const TileSrc& t, int kt, int par) {
        unsigned char* buf = lds + par * BUF_BYTES;
        if (op == 0) {
            if (AMODE == DENSE) {
#pragma unroll
                for (int j = 0; j < 2; ++j) glds16(baseA + (t.a[u][j] + (uint32_t)(kt * BK * 2)), buf + a_block(u, j) * 1024);
            } else {
                const int cblk = kt / 9, tap = kt - cblk * 9, c0 = cblk * BK;
                const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                const int off = ((dy * p.W + dx) * p.Cin + c0) * 2;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const void* src = ((t.valid[u] >> (16 * j + tap)) & 1u) ? (const void*)(baseA + (t.a[u][j] + (uint32_t)off)) : (const void*)zero_src;
                    glds16(src, buf + a_block(u, j) * 1024);
                }
            }
        } else {
            int kb = kt * BK;                                          // element offset of the k-tile in a row of B
            if (AMODE == CONV3) { const int cblk = kt / 9, tap = kt - cblk * 9; kb = tap * p.Cin + cblk * BK; }
#pragma unroll
            for (int j = 0; j < 2; ++j) glds16(baseB + (t.b[u][j] + (uint32_t)(kb * 2)), buf + TILE_BYTES + b_block(u, j) * 1024);
        }
    };

    // ---- fragment read addresses (bytes within a buffer), one per k-substep kk; part / tile offsets are immediates
    const int frow = lane & 15, fq = lane >> 4, fswz = frow >> 1;
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ch = ((kk * 4 + fq) ^ fswz) << 4;
        a_rd[kk] = (wr * 128 + frow) * 128 + ch;
        b_rd[kk] = TILE_BYTES + (wc * 64 + frow) * 128 + ch;
    }

    f32x4 acc[2][4][2][2];       // [A part][m tile][B part][n tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[i][a][j][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];       // A part in use [m tile][kk]; B parts [n tile][kk]

    auto load_a = [&](const unsigned char* buf, int part) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                fa[mt][kk] = *reinterpret_cast<const bf16x8*>(buf + a_rd[kk] + (part * 64 + mt * 16) * 128);
    };
    auto load_b = [&](const unsigned char* buf, int part, bf16x8 (&fb)[2][2]) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                fb[nt][kk] = *reinterpret_cast<const bf16x8*>(buf + b_rd[kk] + (part * 32 + nt * 16) * 128);
    };
    auto mma = [&](int ap, int bp, const bf16x8 (&fb)[2][2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[ap][mt][bp][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt][kk], fa[mt][kk], acc[ap][mt][bp][nt], 0, 0, 0);
    };

    // ---- per-quadrant epilogue.  A lane holds 4 consecutive n of row m in each of the quadrant's two 16-wide column tiles;
    // v_permlane16_swap exchanges them between neighbouring lane rows so that a lane ends with 8 consecutive n of ONE tile
    // (row g: tile g & 1, columns 8 (g >> 1) ..): half as many, 16-byte, stores — the store tail of a tile is issue-bound.
    const int qcol = (fq & 1) * 16 + (fq >> 1) * 8;                      // this lane's 8 columns inside a quadrant's 32
    auto swap_rows = [&](f32x4& x, f32x4& y) {                             // x: tile 0, y: tile 1 -> x: columns 0-3, y: columns 4-7 of this lane's 8
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[r]), __float_as_uint(y[r]), false, false);
            x[r] = __uint_as_float(t[0]); y[r] = __uint_as_float(t[1]);
        }
    };
    auto load_prev = [&](int ap, int bp, int m0, int n0, uint4 (&prev)[4]) {   // EPI_ACCUM: the old C values of a quadrant
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m0 + wr * 128 + ap * 64 + mt * 16 + frow, n = n0 + wc * 64 + bp * 32 + qcol;
            prev[mt] = (m < p.M && n < p.N) ? *reinterpret_cast<const uint4*>(p.C + (size_t)m * p.ldc + n) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    float rs1[2][4], rs2[2][4];                                            // EPI_ACCUM_STATS: this lane's share of its rows' (sum, sum of squares)
    constexpr bool EPI_LN = (EPI == EPI_LN_BIAS || EPI == EPI_LN_BIAS_GELU);
    float rrs[2][4], rnm[2][4];                                            // EPI_LN_*: (rstd, -rstd * mean) of this lane's 8 rows of the tile
    auto row_norms = [&](int m0) {
        // Lane quarter fq of a row's four lanes fetches partial fq (an unconditional load of a clamped index, weighted 0 beyond
        // stats_parts: a load inside a branch or a run-time loop is waited for one at a time — 8 x P dependent round trips, 4 us per
        // tile when it was written that way); the four are added across the quarters in a fixed tree (bit-reproducible).
        const int pi = fq < p.stats_parts ? fq : p.stats_parts - 1;
        const float wgt = fq < p.stats_parts ? 1.0f : 0.0f;
        float2 v[2][4];
#pragma unroll
        for (int ap = 0; ap < 2; ++ap)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                int m = m0 + wr * 128 + ap * 64 + mt * 16 + frow;
                if (m >= p.M) m = p.M - 1;
                v[ap][mt] = *reinterpret_cast<const float2*>(p.stats + ((size_t)pi * p.M + m) * 2);
            }
#pragma unroll
        for (int ap = 0; ap < 2; ++ap)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float sx = quad_sum(v[ap][mt].x * wgt), sq = quad_sum(v[ap][mt].y * wgt);
                const float mean = sx * p.inv_k;
                const float rstd = __builtin_amdgcn_rsqf(fmaxf(sq * p.inv_k - mean * mean, 0.f) + p.eps);
                rrs[ap][mt] = rstd; rnm[ap][mt] = -rstd * mean;
            }
    };
    // the per-column vectors of a quadrant's 8 columns of this lane (bias, and colsum for the LayerNorm-folded forms)
    auto load_cols = [&](int bp, int n0, float (&bq)[8], float (&cq)[8]) {
        int n = n0 + wc * 64 + bp * 32 + qcol;
        if (n > p.N - 8) n = p.N - 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { bq[e] = 0.f; cq[e] = 0.f; }
        // (what these loads and the queue drain in front of their first use cost was measured by building the epilogue without
        // them — wrong results, timing only: 1 us per tile, 0.23 ms per PPNet batch at most, profiles/r05_gemm_epilogue_loads.txt;
        // staging them through LDS by DMA as nat_gemm.hip does would win back part of that and was not built)
        if (EPI != EPI_ACCUM) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
            bq[0] = b0.x; bq[1] = b0.y; bq[2] = b0.z; bq[3] = b0.w; bq[4] = b1.x; bq[5] = b1.y; bq[6] = b1.z; bq[7] = b1.w;
        }
        if (EPI_LN) {
            const float4 c0 = *reinterpret_cast<const float4*>(p.colsum + n), c1 = *reinterpret_cast<const float4*>(p.colsum + n + 4);
            cq[0] = c0.x; cq[1] = c0.y; cq[2] = c0.z; cq[3] = c0.w; cq[4] = c1.x; cq[5] = c1.y; cq[6] = c1.z; cq[7] = c1.w;
        }
    };
    auto finish_quadrant = [&](int ap, int bp, int m0, int n0, const uint4 (&prev)[4], const float (&bq)[8], const float (&cq)[8]) {
        int n = n0 + wc * 64 + bp * 32 + qcol;
        const bool n_in = n < p.N;                                         // N % 8 == 0: a group of 8 is in or out
        if (n > p.N - 8) n = p.N - 8;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x4 x = acc[ap][mt][bp][0], y = acc[ap][mt][bp][1];
            swap_rows(x, y);
            float o[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
            if constexpr (EPI_LN) {
                // two values per instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): no MFMA runs beside an epilogue, so the
                // packed forms are what they are elsewhere, half the issue slots (beside MFMAs they cost more than they save:
                // nat_c128.hip) — the LayerNorm algebra 16 -> 8 instructions per 8 values, the GELU's non-transcendental part 7 -> 4 per value
                const f32x2 rr2 = {rrs[ap][mt], rrs[ap][mt]}, nm2 = {rnm[ap][mt], rnm[ap][mt]};
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    f32x2 v = {o[e], o[e + 1]};
                    v = __builtin_elementwise_fma(rr2, v, __builtin_elementwise_fma(nm2, f32x2{cq[e], cq[e + 1]}, f32x2{bq[e], bq[e + 1]}));
                    if (EPI == EPI_LN_BIAS_GELU) v = gelu_logistic2(v);
                    o[e] = v.x; o[e + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += bq[e];
            }
            if (EPI == EPI_ACCUM || EPI == EPI_ACCUM_STATS) {
                const bf16x8 s8 = __builtin_bit_cast(bf16x8, prev[mt]);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += (float)s8[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (EPI == EPI_BIAS_GELU) o[e] = gelu_erf(o[e]);
                if (EPI == EPI_BIAS_RELU) o[e] = fmaxf(o[e], 0.f);
            }
            const bf16x8 w = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3], (__bf16)o[4], (__bf16)o[5], (__bf16)o[6], (__bf16)o[7]};
            const int m = m0 + wr * 128 + ap * 64 + mt * 16 + frow;
            if (m < p.M && n_in) *reinterpret_cast<bf16x8*>(p.C + (size_t)m * p.ldc + n) = w;
            if (EPI == EPI_ACCUM_STATS) {                                     // of the ROUNDED values: what the next projection reads
                f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 8; e += 2) { const f32x2 f = {(float)w[e], (float)w[e + 1]}; s1 += f; s2 = __builtin_elementwise_fma(f, f, s2); }
                const float a1 = s1.x + s1.y, a2 = s2.x + s2.y;
                if (bp == 0) { rs1[ap][mt] = a1; rs2[ap][mt] = a2; } else { rs1[ap][mt] += a1; rs2[ap][mt] += a2; }
            }
            acc[ap][mt][bp][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[ap][mt][bp][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    // ---- K loop: 4 phases per k-tile, each {fragment reads + one quarter-tile of LDS-DMA + counted wait | barrier | 16 MFMA |
    // barrier}.  Waves 4-7 (the second wave of every SIMD) run one barrier behind waves 0-3, so one group's MFMA segment
    // covers the other group's read / DMA segment instead of both idling the matrix pipe together.
    //   reads of k-tile g:  phase 1: A0, B0 (B0 stays in registers)   phase 2: B1   phase 3: A1   phase 4: none
    //   DMA issue:          phase 1: B1(g+1)   phase 2: A1(g+1)   phase 3: A0(g+2)   phase 4: B0(g+2)
    // (g runs over the k-tiles of all of this block's tiles.)  A unit is overwritten two or more phases after its last read
    // (with the stagger, the other group's reads of it finish one barrier later), and it is waited for one phase before its
    // first read: vmcnt(8) in phases 4, 1 and 2 leaves the four youngest quarter-tiles (8 DMA instructions per lane) in flight;
    // the barrier of that phase publishes the rest.  After an epilogue the same three waits allow EPI_STORES more.
    constexpr int EPI_STORES = 16;                                      // vector-memory instructions an epilogue leaves in the queue (EPI_ACCUM_STATS: +1 / +2 in
                                                                        // wave group 0, whose first waits then reach one or two stores further back: conservative)
    const int G = gridDim.x;
    const int my_tiles = (nblk - (int)blockIdx.x + G - 1) / G;
    const int total = my_tiles * nk;                                    // k-tiles in this block's stream
    TileSrc cur, nxt;
    tile_src(blockIdx.x, cur);
    nxt = cur;
    if (my_tiles > 1) tile_src(blockIdx.x + G, nxt);
    stage(0, 0, cur, 0, 0); stage(1, 0, cur, 0, 0); stage(1, 1, cur, 0, 0); stage(0, 1, cur, 0, 0);
    stage(0, 0, cur, 1, 1); stage(1, 0, cur, 1, 1);                    // nk >= 2
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();           // the stagger: this group's barriers pair with the other's next ones

    // (EPI_ACCUM_STATS: wave group 0 also stores the row partials — one or two more instructions in ITS queue; a wait that allowed
    // only 8 + 16 would reach back to the tile's first stores, which drain slowly: +3 us per tile boundary when it did)
    const int extra_stores = (EPI == EPI_ACCUM_STATS && wr == 0) ? (p.stats_p128 ? 2 : 1) : 0;
#define PPN_GEMM_WAIT(full, wide) do {                                                          \
        if (!(full)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           \
        else if (wide) {                                                                        \
            if (extra_stores == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + EPI_STORES) : "memory");          \
            else if (extra_stores == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(9 + EPI_STORES) : "memory");     \
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(10 + EPI_STORES) : "memory");        \
        } else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                 \
    } while (0)
#define PPN_GEMM_MMA(ap, bp, fb) do {                          \
        __builtin_amdgcn_s_barrier();                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_sched_barrier(0);                     \
        __builtin_amdgcn_s_setprio(1);                         \
        mma(ap, bp, fb);                                       \
        __builtin_amdgcn_s_setprio(0);                         \
        __builtin_amdgcn_s_barrier();                          \
    } while (0)

    // One tile per workgroup (the default launch): the tile's LAST k-tile is peeled and each quadrant is finished — converted,
    // exchanged, stored — in the phase after its final MFMAs, in the segment where this wave group only reads and the other group
    // owns the matrix pipe; three quarters of the store tail then overlap the remaining quadrants' MFMAs.
    const bool overlap = (my_tiles == 1) && (EPI != EPI_RELU_DOT2) && (EPI != EPI_ACCUM_STATS) && !EPI_LN;
    int g = 0, cur_m0 = 0, cur_n0 = 0;
    for (int it = 0; it < my_tiles; ++it) {
        const int nk_loop = overlap ? nk - 1 : nk;
        for (int kt = 0; kt < nk_loop; ++kt, ++g) {
            const unsigned char* buf = lds + (g & 1) * BUF_BYTES;
            const bool wide = (kt == 0) && (it > 0);                    // an epilogue's stores are in the queue
            const bool has1 = g + 1 < total, has2 = g + 2 < total;
            // k-tile g + 1 / g + 2 of the stream: in this tile, or the head of the next one
            const bool in1 = kt + 1 < nk, in2 = kt + 2 < nk;
            const int k1 = in1 ? kt + 1 : 0, k2 = in2 ? kt + 2 : kt + 2 - nk;
            // phase 1
            load_b(buf, 0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            load_a(buf, 0);
            if (has1) { if (in1) stage(1, 1, cur, k1, (g + 1) & 1); else stage(1, 1, nxt, k1, (g + 1) & 1); }
            PPN_GEMM_WAIT(has1, wide);
            PPN_GEMM_MMA(0, 0, fb0);
            // phase 2
            load_b(buf, 1, fb1);
            if (has1) { if (in1) stage(0, 1, cur, k1, (g + 1) & 1); else stage(0, 1, nxt, k1, (g + 1) & 1); }
            PPN_GEMM_WAIT(has1, wide);
            PPN_GEMM_MMA(0, 1, fb1);
            // phase 3
            load_a(buf, 1);
            if (has2) { if (in2) stage(0, 0, cur, k2, g & 1); else stage(0, 0, nxt, k2, g & 1); }
            PPN_GEMM_MMA(1, 1, fb1);
            // phase 4: no fragment reads (A1 and B0 are in registers)
            if (has2) { if (in2) stage(1, 0, cur, k2, g & 1); else stage(1, 0, nxt, k2, g & 1); }
            PPN_GEMM_WAIT(has2, wide);
            PPN_GEMM_MMA(1, 0, fb0);
        }

        const int m0 = cur.m0, n0 = cur.n0;
        cur_m0 = m0; cur_n0 = n0;
        if (overlap) {
            const unsigned char* buf = lds + (g & 1) * BUF_BYTES;           // the last k-tile; everything has landed (vmcnt(0) above)
            uint4 pa[4] = {}, pb[4] = {};
            load_b(buf, 0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            load_a(buf, 0);
            if (EPI == EPI_ACCUM) load_prev(0, 0, m0, n0, pa);
            PPN_GEMM_MMA(0, 0, fb0);
            load_b(buf, 1, fb1);
            { float bq[8], cq[8]; load_cols(0, n0, bq, cq); finish_quadrant(0, 0, m0, n0, pa, bq, cq); }
            if (EPI == EPI_ACCUM) load_prev(0, 1, m0, n0, pb);
            PPN_GEMM_MMA(0, 1, fb1);
            load_a(buf, 1);
            { float bq[8], cq[8]; load_cols(1, n0, bq, cq); finish_quadrant(0, 1, m0, n0, pb, bq, cq); }
            if (EPI == EPI_ACCUM) load_prev(1, 1, m0, n0, pa);
            PPN_GEMM_MMA(1, 1, fb1);
            { float bq[8], cq[8]; load_cols(1, n0, bq, cq); finish_quadrant(1, 1, m0, n0, pa, bq, cq); }
            if (EPI == EPI_ACCUM) load_prev(1, 0, m0, n0, pb);
            PPN_GEMM_MMA(1, 0, fb0);
            { float bq[8], cq[8]; load_cols(0, n0, bq, cq); finish_quadrant(1, 0, m0, n0, pb, bq, cq); }
            ++g;
        } else if (EPI == EPI_RELU_DOT2) {
            // max(acc + bias, 0) . w2[c] over this wave's 64 columns: reduce over the 4 lane quarters, then ONE plain 8-byte store per
            // row into this wave's column slot (n0 / 64 + wc): every (slot, row) has exactly one writer, and the slots are added in
            // a fixed order afterwards — the logits are bit-reproducible (round 3 used float atomics: the last bit of a logit
            // depended on the arrival order of its eight adders)
            float4 bv[2][2], w0v[2][2], w1v[2][2];
#pragma unroll
            for (int bp = 0; bp < 2; ++bp)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    int n = n0 + wc * 64 + bp * 32 + nt * 16 + fq * 4;
                    const bool in = n < p.N;                                      // N % 4 == 0: a group of 4 is in or out
                    if (n > p.N - 4) n = p.N - 4;
                    bv[bp][nt] = *reinterpret_cast<const float4*>(p.bias + n);
                    w0v[bp][nt] = in ? *reinterpret_cast<const float4*>(p.w2 + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                    w1v[bp][nt] = in ? *reinterpret_cast<const float4*>(p.w2 + p.N + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
            for (int ap = 0; ap < 2; ++ap)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    float s0 = 0.f, s1 = 0.f;
#pragma unroll
                    for (int bp = 0; bp < 2; ++bp)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const f32x4 a4 = acc[ap][mt][bp][nt];
                            const float4 b4 = bv[bp][nt], u4 = w0v[bp][nt], w4 = w1v[bp][nt];
                            const float v0 = fmaxf(a4[0] + b4.x, 0.f), v1 = fmaxf(a4[1] + b4.y, 0.f), v2 = fmaxf(a4[2] + b4.z, 0.f), v3 = fmaxf(a4[3] + b4.w, 0.f);
                            s0 += v0 * u4.x + v1 * u4.y + v2 * u4.z + v3 * u4.w;
                            s1 += v0 * w4.x + v1 * w4.y + v2 * w4.z + v3 * w4.w;
                            acc[ap][mt][bp][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    s0 += __shfl_xor(s0, 16, 64); s0 += __shfl_xor(s0, 32, 64);
                    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
                    // into the LDS buffer that held k-tile g - 2 (nobody reads it any more: the other wave group is at most one
                    // barrier behind, inside k-tile g - 1): red[wc][row of the tile] — the four wave columns of a row are added in
                    // order by ONE thread after the closing barrier, so a 256-column block leaves one partial per row, not four
                    if (fq == 0) reinterpret_cast<float2*>(lds + (g & 1) * BUF_BYTES)[wc * BM + wr * 128 + ap * 64 + mt * 16 + frow] = make_float2(s0, s1);
                }
        } else {
            // persistent form: the whole tile is finished here; exactly EPI_STORES vector-memory instructions stay in the queue.
            // The two wave groups run one barrier apart, so left alone their epilogues SERIALISE: group 0's next barrier (phase 1 of
            // the next tile) pairs with group 1's last one of this tile, group 1 waits there through group 0's whole epilogue and
            // group 0 then waits through group 1's (measured round 5: 2 x ~2 us per tile).  One extra barrier of group 0 in front
            // aligns the groups, both finish their halves of the tile at the same time, and one extra barrier of group 1 behind the
            // epilogue restores the stagger for the next tile's k-loop.
            if (wr == 0) __builtin_amdgcn_s_barrier();
            if (EPI == EPI_ACCUM || EPI == EPI_ACCUM_STATS) {
                uint4 pq[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) load_prev(q >> 1, q & 1, m0, n0, pq[q]);
                float bq[2][8], cq[2][8];
                load_cols(0, n0, bq[0], cq[0]); load_cols(1, n0, bq[1], cq[1]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the 16 + 4 reads drain the queue once
#pragma unroll
                for (int q = 0; q < 4; ++q) finish_quadrant(q >> 1, q & 1, m0, n0, pq[q], bq[q & 1], cq[q & 1]);
                if (EPI == EPI_ACCUM_STATS) {
                    // row sums: over the row's 4 lanes (xor 16, 32), then red[wave column][row of the tile] in LDS; thread r of wave
                    // group 0 adds the wave columns IN ORDER (bit-reproducible, no atomics) and is the only writer of (partial, row).
                    float2* red = reinterpret_cast<float2*>(lds + LDS_BYTES);
#pragma unroll
                    for (int ap = 0; ap < 2; ++ap)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt) {
                            const float a1 = quad_sum(rs1[ap][mt]), a2 = quad_sum(rs2[ap][mt]);
                            if (fq == 0) lds_st_f2(red + wc * BM + wr * 128 + ap * 64 + mt * 16 + frow, make_float2(a1, a2));
                        }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                          // (the groups are aligned here: one barrier publishes all eight waves' sums)
                    if (wr == 0) {
                        const float2 r0 = lds_f2(red + tid), r1 = lds_f2(red + BM + tid), r2 = lds_f2(red + 2 * BM + tid), r3 = lds_f2(red + 3 * BM + tid);
                        float2* so = reinterpret_cast<float2*>(p.stats);
                        if (p.stats_p128) {
                            so[(size_t)(n0 / 128) * p.M + m0 + tid] = make_float2(r0.x + r1.x, r0.y + r1.y);
                            so[(size_t)(n0 / 128 + 1) * p.M + m0 + tid] = make_float2(r2.x + r3.x, r2.y + r3.y);
                        } else {
                            so[(size_t)(n0 / BN) * p.M + m0 + tid] = make_float2(((r0.x + r1.x) + r2.x) + r3.x, ((r0.y + r1.y) + r2.y) + r3.y);
                        }
                    }
                }
            } else {
                // every load of the epilogue in front of its first store: a load's first use drains the WHOLE queue (hipcc waits
                // vmcnt(0) for an ordinary load beside LDS-DMA), and behind a quadrant's stores that drain would wait for them too
                const uint4 none[4] = {};
                float bq[2][8], cq[2][8];
                load_cols(0, n0, bq[0], cq[0]); load_cols(1, n0, bq[1], cq[1]);
                if (EPI_LN) row_norms(m0);
#pragma unroll
                for (int q = 0; q < 4; ++q) finish_quadrant(q >> 1, q & 1, m0, n0, none, bq[q & 1], cq[q & 1]);
            }
            if (wr == 1) __builtin_amdgcn_s_barrier();       // the stagger again: pairs with group 0's first barrier of the next tile (or its closing one)
        }
        // next tile
        cur = nxt;
        if (it + 2 < my_tiles) tile_src(blockIdx.x + (it + 2) * G, nxt);
    }
#undef PPN_GEMM_WAIT
#undef PPN_GEMM_MMA
    if (wr == 0) __builtin_amdgcn_s_barrier();           // rebalance the barrier count of the two groups
    if (EPI == EPI_RELU_DOT2) {
        // one tile per workgroup (the launcher guarantees it): g == nk, the partial sums of the tile's 256 rows x 4 wave columns sit
        // in buffer g & 1.  Thread r adds row r's four in wave-column order — a fixed order: bit-reproducible — and is the only
        // writer of (column block, row): with one column block (N <= 256) straight onto the logits, which hold the classifier's bias.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < BM) {
            const float2* red = reinterpret_cast<const float2*>(lds + (g & 1) * BUF_BYTES);
            float2 a = red[tid];
#pragma unroll
            for (int c = 1; c < 4; ++c) { const float2 v = red[c * BM + tid]; a.x += v.x; a.y += v.y; }
            const int m = cur_m0 + tid;
            if (m < p.M) {
                float2* dst = reinterpret_cast<float2*>(p.logits) + ((size_t)(tiles_n > 1 ? cur_n0 / BN : 0) * p.M + m);
                if (tiles_n == 1) { const float2 b = *dst; a.x += b.x; a.y += b.y; }
                *dst = a;
            }
        }
    }
}

}  // namespace gemm
}  // namespace ppn
