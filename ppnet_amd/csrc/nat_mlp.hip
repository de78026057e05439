// nat_mlp.hip — the MLP half of a NAT layer as ONE kernel for the HBM-bound level C = 256 (DiNAT-B / NAT-B level 1):
//
//     s += GELU(LN(s) W1^T + b1) W2^T + b2          (SegNet/nat.py:62-85 `Mlp.forward`, :147-153 the second half of NATLayer.forward;
//                                                    LayerNorm folded into W1 / b1, LayerScale into W2 / b2 by the host)
//
// The hidden activation h [tokens][2C] never exists: as two GEMMs it is written and read once per layer (268 MB at level 1, a
// third of each projection's traffic).  Structure — a "flash" MLP: the hidden dimension is walked in CHUNKS of 32 units,
//
//     P   = W1'[chunk] . s^T            (stage 1: K = C, raw residual rows as the B operand)
//     h   = GELU(rstd (P - mean colsum) + b1')      LayerNorm applied to the product: rstd (s W'^T - mean colsum(W')) + b'
//     O  += W2'[:, chunk] . h           (stage 2: K = 32)
//
// with every product TRANSPOSED (D^T[out][token] = W . Y^T, as in nat_c128.hip) so that the accumulator layout — lane = token
// column, four consecutive rows per register group — IS the B-operand layout of the next product: P feeds stage 2 without a
// transpose.  Two waves (a PAIR) share 32 tokens (two 16-token MFMA columns: every weight fragment read from LDS feeds two
// MFMAs): wave h of the pair computes hidden tile h of every chunk (16 of its 32 units) and the output channels [128 h, 128 h +
// 128) — O = 64 accumulator registers, the raw rows of s = 64 registers (the stage-1 operand of every chunk); the other half of
// a chunk's GELU(fc1) arrives from the partner through 8 bytes per lane of LDS.  (A wave that owns all 256 output channels of
// its tokens needs 128 + 64 + 16 + 16 registers before the first fragment: hipcc spills 400 of them.)  The output channel <->
// (tile, row) assignment makes a lane's O values 16 consecutive channels per four tiles, and the stage-1 k-slot <-> channel
// assignment is chosen to hold THE SAME channels: O starts as s + b2 converted in registers, the residual add costs nothing, and
// once the last chunk's stage 1 is issued the s registers are free for the next row block's prefetch.
//
// Weights stream: the host packs W1' and W2' chunk by chunk in MFMA A-fragment order (1 KiB per fragment, lane-linear), so a
// chunk (2 x C/32 + C/16 fragments = 32 KiB) is one contiguous run in memory AND in LDS: 16-byte LDS-DMA (global_load_lds),
// conflict-free ds_read_b128, no swizzle.  Ring of 4 chunk slots; a workgroup = 4 pairs = 128 tokens, one per CU, persistent over
// its row blocks; the chunk sequence is cyclic, so the ring never drains between row blocks.  ONE barrier per chunk: it retires
// the DMA of chunk t, frees the slot of chunk t-2 and publishes the pair's halves of GELU(P) of chunk t-1.  In a chunk's
// iteration a wave issues stage 1 of chunk t, then stage 2 of chunk t-1 with the GELU of chunk t's P between its MFMAs.
//
// Row statistics of the NEW s — (sum, sum of squares) of the bfloat16 values stored, per 128 columns (= per wave of a pair) —
// leave with the epilogue: what the next layer's LayerNorm-folded qkv projection (ppn_nat_gemm_bf16 mode 0) reads as stats_in.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "ppn_device.h"
#include "ppn_kernels.h"
#pragma clang fp contract(fast)      // network arithmetic checked to a tolerance (see nat_c128.hip)

#ifndef NMLP_TRACE
#define NMLP_TRACE 0      // diagnostic build: s_memtime stamps of workgroup 0's waves into the stats buffer (tools/mlp_trace.py)
#endif
#ifndef NMLP_STAGGER
#define NMLP_STAGGER 0
#endif
#ifndef NMLP_ABL
#define NMLP_ABL 0        // diagnostic builds (tools/r04_mlp_abl.sh): bit 0 no ring DMA after the prologue, 1 no GELU, 2 no per-chunk
#endif                    // barrier, 3 no stage-2 MFMAs, 4 no stage-1 MFMAs, 5 no fragment reads.  Never shipped.

namespace ppn {
namespace nmlp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

constexpr int NTHR = 512, NWAVE = 8, TOK = 32;      // 4 pairs of waves x 32 tokens = 128 rows per workgroup pass
constexpr int HC = 32;                              // hidden units per chunk (two 16-row MFMA tiles = one stage-2 k-step)
constexpr int NS = 4;                               // ring slots: chunk t-1 (stage 2), t (stage 1), t+1, t+2 (in flight)
constexpr int ROWS = (NWAVE / 2) * TOK;

template <int C>
struct Geo {
    static constexpr int KS = C / 32;               // stage-1 k-steps
    static constexpr int OT = C / 16;               // output tiles (a wave computes OT / 2 of them)
    static constexpr int FR = 2 * KS + OT;          // fragments per chunk
    static constexpr int CHUNK = FR * 1024;         // bytes
    static constexpr int DMA_PER_WAVE = FR / NWAVE; // 16-byte LDS-DMA instructions a wave issues per chunk
    static constexpr int XCH = 2 * NWAVE * 2 * 64 * 8;   // [parity][wave][token tile][lane] 8 bytes: the pair's exchange of GELU(P) halves
    static_assert(FR % NWAVE == 0, "a chunk is a whole number of DMA instructions per wave");
    static_assert(C == 256, "a pair covers 2 x 128 output channels and one partial of row statistics each");
};

struct Params {
    __bf16* s;                 // [M][C] in / out
    const __bf16* wpk;         // [NCH][FR][64][8] packed fragments (nat_mlp_pack)
    const float* hb;           // [HID][2]: (colsum(W1')[h], b1'[h])
    const float* b2;           // [C]
    float* stats_out;          // [C / 128][M][2] or null
    int M, NCH;                // rows, hidden chunks (HID / 32)
    float eps, inv_c;
};

// LDS-DMA of 16 bytes per lane, base + off -> lds_uniform + 16 * lane, as assembly on purpose (na2d_halo16.hip): behind a
// __builtin_amdgcn_global_load_lds the compiler cannot tell which LDS bytes are in flight and puts `s_waitcnt vmcnt(0)` in front of
// the next LDS read — here the first read of every chunk iteration, i.e. a full memory round trip per chunk (measured: 2 us per
// chunk, the whole kernel).  What it does not see it does not wait for; the kernel's own counted waits order the ring.
__device__ __forceinline__ void dma16(const void* base_uniform, unsigned off, unsigned lds_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base_uniform), "s"(lds_uniform) : "memory");
}
// erf-GELU through a logistic fit of erf (nat_gemm.hip: |error| < 3e-5), one v_exp + one v_rcp
__device__ __forceinline__ float gelu_logistic(float x) {
    const float x2 = fminf(x * x, 64.0f);
    const float t = x * (2.3009787f + x2 * (0.10690469f - 1.0350827e-3f * x2));
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-t));
}
// the polynomial form of nat_c128.hip (no transcendental; |error| <= 9.2e-5)
__device__ __forceinline__ float gelu_poly(float v) {
    constexpr float K[8] = {3.984200563e-01f, -6.545352466e-02f, 9.257837137e-03f, -9.404510850e-04f, 6.552754503e-05f, -2.938833893e-06f,
                            7.570944350e-08f, -8.460567657e-10f};
    const float vc = __builtin_amdgcn_fmed3f(v, -4.25f, 4.25f), u = vc * vc;
    float q = K[7];
#pragma unroll
    for (int k = 6; k >= 0; --k) q = fmaf(q, u, K[k]);
    return v * fmaf(vc, q, 0.5f);
}
#ifndef NMLP_GELU_SCALAR
#define NMLP_GELU_SCALAR 2      // 0 = logistic fit, two values per instruction (A/B builds), 1 = the same one value per instruction, 2 = polynomial one value per instruction (ships)
#endif
#if NMLP_GELU_SCALAR == 2
#define NMLP_GELU_FN gelu_poly
#else
#define NMLP_GELU_FN gelu_logistic
#endif
typedef __attribute__((ext_vector_type(2))) float f32x2;
// two at a time: the multiplies / adds as packed float32 instructions (v_pk_mul_f32, v_pk_fma_f32)
__device__ __forceinline__ f32x2 gelu_logistic2(f32x2 x) {
    f32x2 x2 = x * x;
    x2 = f32x2{fminf(x2.x, 64.0f), fminf(x2.y, 64.0f)};
    const f32x2 k0 = {2.3009787f, 2.3009787f}, k1 = {0.10690469f, 0.10690469f}, k2 = {-1.0350827e-3f, -1.0350827e-3f}, one = {1.0f, 1.0f};
    const f32x2 t = x * __builtin_elementwise_fma(x2, __builtin_elementwise_fma(x2, k2, k1), k0);
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(-t.x), __builtin_amdgcn_exp2f(-t.y)} + one;
    return x * f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
// LDS reads through the fragment type: hipcc drains the LDS-DMA queue (vmcnt(0)) in front of float4-typed LDS reads
__device__ __forceinline__ f32x4 lds_f4(const void* ptr) { return __builtin_bit_cast(f32x4, *reinterpret_cast<const bf16x8*>(ptr)); }
__device__ __forceinline__ uint2 lds_u2(const void* ptr) { return __builtin_bit_cast(uint2, *reinterpret_cast<const bf16x4*>(ptr)); }
__device__ __forceinline__ bf16x8 frag(const unsigned char* slot, int f, int lane) {
    return *reinterpret_cast<const bf16x8*>(slot + f * 1024 + lane * 16);
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// Channel maps (lane = (token n = lane & 15, group g = lane >> 4)):
//   stage-1 k-slot (ks, g, e)        = channel 64 (ks >> 1) + 16 g + 8 (ks & 1) + e          (what sraw[.][ks] element e holds)
//   output tile ot, accumulator row 4 g + r = channel 64 (ot >> 2) + 16 g + 4 (ot & 3) + r    (wave h of a pair: ot in [8 h, 8 h + 8))
//   stage-2 k-slot (g, e) of chunk j = hidden unit 32 j + 16 (e >> 2) + 4 g + (e & 3)          (the order P leaves the matrix pipe
//                                      in: elements 0-3 come from the pair's wave 0, elements 4-7 from its wave 1)

// HW = the wave's half of its pair, a template parameter so that every register index below is a constant (as a run-time value —
// even a wave-uniform one — the selects between the two halves of sraw keep both alive and spill into the chunk loop)
// ORD = the order of an iteration's phases.  All eight waves meet at one barrier per chunk, so without it they run in lock step:
// every wave reads its fragments, then every wave is in the matrix pipe, then every wave in the GELU's vector work — the units
// take turns (measured: 4300 cycles per chunk for 1024 of matrix work per SIMD).  Waves 0-3 (ORD 0) run stage 1 -> GELU ->
// stage 2, waves 4-7 (ORD 1: the other wave of every SIMD) stage 2 -> stage 1 -> GELU: one wave's vector phase lies beside the
// other's matrix phase.  The data flow is the same (stage 2 consumes the PREVIOUS chunk's GELU either way).
#if NMLP_TRACE
#define NMLP_T(slot) do { if (blockIdx.x == 0 && lane == 0 && tr_n < 4096) { tr[tr_n * 2] = __builtin_readcyclecounter(); tr[tr_n * 2 + 1] = (slot); ++tr_n; } } while (0)
#else
#define NMLP_T(slot) do { } while (0)
#endif
template <int C, int HW, int ORD>
__device__ __forceinline__ void nat_mlp_body(const Params& p, unsigned char* lds, const int wave) {
    using G = Geo<C>;
    constexpr int KS = G::KS, OTW = G::OT / 2, CHUNK = G::CHUNK;
    constexpr int EPI_STORES = 2 * (2 * 2 + 1);                        // vector-memory instructions of a row block's epilogue
    constexpr int hw = HW;
    unsigned char* ring = lds;
    unsigned char* xch = lds + NS * CHUNK;
    float* hbl = reinterpret_cast<float*>(xch + G::XCH);               // [HID][2]
    float* b2l = hbl + 2 * HC * p.NCH;                                 // [C]
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int pair = wave >> 1;                                        // the pair's 32 tokens
#if NMLP_TRACE
    unsigned long long* tr = reinterpret_cast<unsigned long long*>(p.stats_out) + (size_t)wave * 8192;
    int tr_n = 0;
#endif

    const int nblk = p.M / ROWS;
    const int my_blocks = ((int)blockIdx.x < nblk) ? (nblk - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (my_blocks == 0) return;
    const int total = my_blocks * p.NCH;                               // chunks this workgroup walks

    // this wave's share of a chunk's DMA: fragments [wave * DMA_PER_WAVE, +DMA_PER_WAVE)
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.wpk) + (size_t)wave * G::DMA_PER_WAVE * 1024;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned loff = (unsigned)lane * 16;
    auto dma = [&](int j, int slot) __attribute__((always_inline)) {
        const unsigned char* src = wsrc + (size_t)j * CHUNK;
        const unsigned dst = lds0 + slot * CHUNK + wave * G::DMA_PER_WAVE * 1024;
#pragma unroll
        for (int f = 0; f < G::DMA_PER_WAVE; ++f) dma16(src + f * 1024, loff, dst + f * 1024);
    };
    int jn = 0;                                                        // chunk index of the next DMA (t + 2 in steady state)
    dma(jn, 0);
    jn = (jn + 1 == p.NCH) ? 0 : jn + 1;
    if (total > 1) { dma(jn, 1); jn = (jn + 1 == p.NCH) ? 0 : jn + 1; }

    uint4 sraw[2][KS];                                                 // raw bf16 rows: [token tile][k-step] = 8 channels
    auto load_rows = [&](int blk) __attribute__((always_inline)) {
        const __bf16* base = p.s + ((size_t)blk * ROWS + pair * TOK + n) * C + 16 * g;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                sraw[tt][ks] = *reinterpret_cast<const uint4*>(base + (size_t)tt * 16 * C + 64 * (ks >> 1) + 8 * (ks & 1));
    };
    load_rows(blockIdx.x);
    // the exchange slots of this lane: own half out, the partner's half in
    unsigned char* x_own = xch + ((size_t)wave * 2 * 64 + lane) * 8;
    const unsigned char* x_par = xch + ((size_t)(wave ^ 1) * 2 * 64 + lane) * 8;
    constexpr int XPAR = NWAVE * 2 * 64 * 8;                           // bytes between the two parities

    int t = 0;
    for (int b = 0; b < my_blocks; ++b) {
        const int blk = blockIdx.x + b * gridDim.x;
        // this block's rows have landed (and every DMA issued before them); the previous block's stores — issued after the row
        // prefetch — stay in flight: 4 row stores + 1 statistics store per token tile
        NMLP_T(1);
        if (b == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(EPI_STORES) : "memory");
        NMLP_T(2);
        // ---- LayerNorm statistics of the raw rows (over all C channels: 4 lanes per token)
        float rstd[2], nmr[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float sx = 0.f, sq = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint32_t w[4] = {sraw[tt][ks].x, sraw[tt][ks].y, sraw[tt][ks].z, sraw[tt][ks].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = bf_lo(w[k]), c = bf_hi(w[k]);
                    sx += a + c;
                    sq = fmaf(a, a, fmaf(c, c, sq));
                }
            }
            sx += __shfl_xor(sx, 16, 64); sq += __shfl_xor(sq, 16, 64);
            sx += __shfl_xor(sx, 32, 64); sq += __shfl_xor(sq, 32, 64);
            const float mean = sx * p.inv_c;
            rstd[tt] = __builtin_amdgcn_rsqf(fmaxf(sq * p.inv_c - mean * mean, 0.f) + p.eps);
            nmr[tt] = -mean * rstd[tt];
        }
        // ---- O starts as s + b2 (the residual and the bias, in the accumulators' own channel order): tiles [8 hw, 8 hw + 8)
        f32x4 acc[OTW][2];
#pragma unroll
        for (int o = 0; o < OTW; ++o) {
            const int q = 2 * hw + (o >> 2), o4 = o & 3;              // (hw is wave-uniform: the selects below are scalar)
            const f32x4 bb = lds_f4(b2l + 64 * q + 16 * g + 4 * o4);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const uint4 v = sraw[tt][4 * hw + 2 * (o >> 2) + (o4 >> 1)];
                const uint32_t w0 = (o4 & 1) ? v.z : v.x, w1 = (o4 & 1) ? v.w : v.y;
                acc[o][tt] = f32x4{bf_lo(w0) + bb[0], bf_hi(w0) + bb[1], bf_lo(w1) + bb[2], bf_hi(w1) + bb[3]};
            }
        }

        NMLP_T(3);
        uint2 hown[2];                                                 // this wave's half of GELU(P) of the previous chunk, per token tile
        hown[0] = make_uint2(0u, 0u); hown[1] = make_uint2(0u, 0u);
#pragma unroll 1
        for (int j = 0; j < p.NCH; ++j, ++t) {
            // chunk t has landed (this wave's share), everyone is done with iteration t-1 (the slot of chunk t-2 may be refilled, the
            // halves of GELU(P) of chunk t-1 are in the exchange buffer)
            // (the first two chunks of a row block landed before the block's rows did — vector memory retires in order — and behind
            // them only the previous block's stores may still be in flight, which nobody here waits for)
            if (j >= 2) {
                if (t + 1 < total) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(G::DMA_PER_WAVE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            NMLP_T(10);
            // the ds_write of this wave's GELU(P) half (foot of the previous iteration) must have LANDED before the barrier
            // releases the partner's read: gfx950's back-off barrier does not imply the wait and hipcc does not add it
            if (!(NMLP_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            NMLP_T(11);
            if (!(NMLP_ABL & 1) && t + 2 < total) { dma(jn, (t + 2) & (NS - 1)); jn = (jn + 1 == p.NCH) ? 0 : jn + 1; }
            const unsigned char* cur = ring + (t & (NS - 1)) * CHUNK;
            const unsigned char* prev = ring + ((t - 1) & (NS - 1)) * CHUNK;
            const int par_prev = ((t - 1) & 1) * XPAR, par_cur = (t & 1) * XPAR;

            // the partner's half of chunk t-1: stage 2's B operand is (wave 0's four, wave 1's four) hidden units per lane.  A row
            // block's first chunk has no predecessor: its stage 2 runs on a ZERO operand against this chunk's own (finite) fragments
            // — 16 idle MFMAs per 528, instead of a branch around the accumulators (hipcc then keeps two copies of them: +64
            // registers, spilled into this loop)
            bf16x8 hf[2];
            const bool first = (j == 0);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const uint2 o = lds_u2(x_par + par_prev + tt * 64 * 8);
                uint4 v = hw ? make_uint4(o.x, o.y, hown[tt].x, hown[tt].y) : make_uint4(hown[tt].x, hown[tt].y, o.x, o.y);
                if (first) v = make_uint4(0u, 0u, 0u, 0u);
                hf[tt] = __builtin_bit_cast(bf16x8, v);
            }
            const unsigned char* s2 = first ? cur : prev;
            // ---- LDS reads are issued a phase ahead of their use (the latency of a fragment read is ~10 MFMAs): this wave's
            // stage-1 fragments a1 (hidden tile hw of chunk t), its stage-2 fragments a2 (output tiles of chunk t-1), (colsum, b1')
            // of this lane's four hidden units 32 j + 16 hw + 4 g + r
            bf16x8 a1[KS], a2[OTW];
            f32x4 P[2];
            float hv[2][4];
            const float* hbj = hbl + (size_t)(HC * j + 16 * hw + 4 * g) * 2;
            auto read_a1 = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (NMLP_ABL & 32) { a1[ks] = __builtin_bit_cast(bf16x8, sraw[0][ks]); asm volatile("" : "+v"(a1[ks])); }
                    else a1[ks] = frag(cur, hw * KS + ks, lane);
                }
            };
            auto read_a2 = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int o = 0; o < OTW; ++o) {
                    if (NMLP_ABL & 32) { a2[o] = __builtin_bit_cast(bf16x8, sraw[1][o]); asm volatile("" : "+v"(a2[o])); }
                    else a2[o] = frag(s2, 2 * KS + OTW * hw + o, lane);
                }
            };
            auto stage1 = [&]() __attribute__((always_inline)) {           // P[tt] = W1'[32 j + 16 hw ..][:] . s^T
                P[0] = f32x4{0.f, 0.f, 0.f, 0.f}; P[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (NMLP_ABL & 16) { asm volatile("" :: "v"(a1[ks])); continue; }
                    P[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[ks], __builtin_bit_cast(bf16x8, sraw[0][ks]), P[0], 0, 0, 0);
                    P[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[ks], __builtin_bit_cast(bf16x8, sraw[1][ks]), P[1], 0, 0, 0);
                }
            };
            auto stage2 = [&]() __attribute__((always_inline)) {           // O += W2'[:, chunk t-1] . h
#pragma unroll
                for (int o = 0; o < OTW; ++o) {
                    if (NMLP_ABL & 8) { asm volatile("" :: "v"(a2[o]), "v"(hf[0]), "v"(hf[1])); continue; }
                    acc[o][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[o], hf[0], acc[o][0], 0, 0, 0);
                    acc[o][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[o], hf[1], acc[o][1], 0, 0, 0);
                }
            };
            auto gelu = [&]() __attribute__((always_inline)) {             // the 8 values this lane owns
                const f32x4 c0 = lds_f4(hbj), c1 = lds_f4(hbj + 4);       // (cs, b1') of hidden units r = 0, 1 | 2, 3
#if NMLP_GELU_SCALAR
                // one value per instruction: beside MFMAs a packed float32 operation costs more issue time than the two plain ones it
                // replaces (MI355X_MICROARCH.md, cycle constants)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const float cs[4] = {c0[0], c0[2], c1[0], c1[2]}, bb[4] = {c0[1], c0[3], c1[1], c1[3]};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = fmaf(P[tt][r], rstd[tt], fmaf(nmr[tt], cs[r], bb[r]));
                        hv[tt][r] = (NMLP_ABL & 2) ? v : NMLP_GELU_FN(v);
                    }
                }
                return;
#endif
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const f32x2 rs = {rstd[tt], rstd[tt]}, nm = {nmr[tt], nmr[tt]};
                    const f32x2 v01 = __builtin_elementwise_fma(f32x2{P[tt][0], P[tt][1]}, rs, __builtin_elementwise_fma(nm, f32x2{c0[0], c0[2]}, f32x2{c0[1], c0[3]}));
                    const f32x2 v23 = __builtin_elementwise_fma(f32x2{P[tt][2], P[tt][3]}, rs, __builtin_elementwise_fma(nm, f32x2{c1[0], c1[2]}, f32x2{c1[1], c1[3]}));
                    const f32x2 g01 = (NMLP_ABL & 2) ? v01 : gelu_logistic2(v01), g23 = (NMLP_ABL & 2) ? v23 : gelu_logistic2(v23);
                    hv[tt][0] = g01.x; hv[tt][1] = g01.y; hv[tt][2] = g23.x; hv[tt][3] = g23.y;
                }
            };
            if constexpr (ORD == 0) {
                read_a1();
                __builtin_amdgcn_sched_barrier(0);
                stage1();
                read_a2();
                __builtin_amdgcn_sched_barrier(0);
                gelu();
                __builtin_amdgcn_sched_barrier(0);
                stage2();
            } else {
                read_a2();
                __builtin_amdgcn_sched_barrier(0);
                stage2();
                read_a1();
                __builtin_amdgcn_sched_barrier(0);
                stage1();
                __builtin_amdgcn_sched_barrier(0);
                gelu();
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                hown[tt] = make_uint2(pack_bf16x2(hv[tt][0], hv[tt][1]), pack_bf16x2(hv[tt][2], hv[tt][3]));
                *reinterpret_cast<uint2*>(x_own + par_cur + tt * 64 * 8) = hown[tt];
            }
        }
        NMLP_T(4);
        // the rows are dead as an operand: the next block's ride in behind the last stage 2
        if (b + 1 < my_blocks) load_rows(blk + gridDim.x);
        {
            // the last chunk's halves: published by a barrier of their own (every wave of the workgroup takes it)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned char* prev = ring + ((t - 1) & (NS - 1)) * CHUNK;
            const int par_prev = ((t - 1) & 1) * XPAR;
            bf16x8 hf[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const uint2 o = lds_u2(x_par + par_prev + tt * 64 * 8);
                const uint4 v = hw ? make_uint4(o.x, o.y, hown[tt].x, hown[tt].y) : make_uint4(hown[tt].x, hown[tt].y, o.x, o.y);
                hf[tt] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int o = 0; o < OTW; ++o) {
                const bf16x8 a = frag(prev, 2 * KS + OTW * hw + o, lane);
                acc[o][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, hf[0], acc[o][0], 0, 0, 0);
                acc[o][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, hf[1], acc[o][1], 0, 0, 0);
                if (o & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        NMLP_T(5);
        // ---- epilogue: 32-byte stores (a token's four lanes write 128 contiguous bytes per q), row statistics of this wave's 128 columns
        __bf16* orow = p.s + ((size_t)blk * ROWS + pair * TOK + n) * C + 128 * hw + 16 * g;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float sx = 0.f, sq = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                uint32_t w[8];
#pragma unroll
                for (int o4 = 0; o4 < 4; ++o4) {
                    const f32x4 v = acc[4 * q + o4][tt];
                    w[2 * o4] = pack_bf16x2(v[0], v[1]);
                    w[2 * o4 + 1] = pack_bf16x2(v[2], v[3]);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float a = bf_lo(w[k]), c = bf_hi(w[k]);
                    sx += a + c;
                    sq = fmaf(a, a, fmaf(c, c, sq));
                }
                uint4* dst = reinterpret_cast<uint4*>(orow + (size_t)tt * 16 * C + 64 * q);
                dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
                dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
            }
            {
                // (always one store instruction per token tile, so that the counted wait above is exact: without stats_out the
                // lanes are masked off and the instruction retires at once)
                sx += __shfl_xor(sx, 16, 64); sq += __shfl_xor(sq, 16, 64);
                sx += __shfl_xor(sx, 32, 64); sq += __shfl_xor(sq, 32, 64);
                float* sp = p.stats_out ? p.stats_out + ((size_t)hw * p.M + (size_t)blk * ROWS + pair * TOK + tt * 16 + n) * 2 : nullptr;
#if !NMLP_TRACE
                if (g == 0 && sp) *reinterpret_cast<float2*>(sp) = make_float2(sx, sq);
#else
                asm volatile("" :: "v"(sx), "v"(sq), "v"(sp));
#endif
            }
        }
        NMLP_T(6);
    }
#if NMLP_TRACE
    if (blockIdx.x == 0 && lane == 0) { tr[tr_n * 2] = 0; tr[tr_n * 2 + 1] = 0; }
#endif
}

template <int C>
__global__ __launch_bounds__(NTHR, 2) void nat_mlp_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    using G = Geo<C>;
    float* hbl = reinterpret_cast<float*>(lds + NS * G::CHUNK + G::XCH);
    float* b2l = hbl + 2 * HC * p.NCH;
    for (int i = threadIdx.x; i < 2 * HC * p.NCH; i += NTHR) hbl[i] = p.hb[i];
    for (int i = threadIdx.x; i < C; i += NTHR) b2l[i] = p.b2[i];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: the branch below is a scalar branch
#if NMLP_STAGGER
    // Workgroups start together and do identical work, so their row-block boundaries (a burst of row loads and stores, ~50 MB
    // over the chip, then nothing for 16 chunks) coincide: spread the starts over ~NMLP_STAGGER us so that the bursts interleave
    for (int k = 0; k < (int)((blockIdx.x >> 3) & 7) * NMLP_STAGGER; ++k) __builtin_amdgcn_s_sleep(32);     // ~32 x 64 cycles ~ 1 us
#endif
    if (wave & 4) {
        if (wave & 1) nat_mlp_body<C, 1, 1>(p, lds, wave);
        else nat_mlp_body<C, 0, 1>(p, lds, wave);
    } else {
        if (wave & 1) nat_mlp_body<C, 1, 0>(p, lds, wave);
        else nat_mlp_body<C, 0, 0>(p, lds, wave);
    }
}

}  // namespace nmlp

// Host-side packing order, as a device kernel (one thread per 16-byte piece): wpk[j][f][lane][8] from the torch Linear layouts
// w1 [HID][C] (LayerNorm-folded) and w2 [C][HID] (LayerScale-folded).  See the channel maps above.
template <int C>
__global__ void nat_mlp_pack_kernel(const __bf16* __restrict__ w1, const __bf16* __restrict__ w2, __bf16* __restrict__ wpk, int HID) {
    using G = nmlp::Geo<C>;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)(HID / nmlp::HC) * G::FR * 64;
    if (i >= total) return;
    const int lane = (int)(i & 63), f = (int)((i >> 6) % G::FR), j = (int)((i >> 6) / G::FR);
    const int n = lane & 15, g = lane >> 4;
    __bf16 v[8];
    if (f < 2 * G::KS) {                                                // stage 1: W1'[32 j + 16 ht + n][channels of k-slot (ks, g, .)]
        const int ht = f / G::KS, ks = f % G::KS;
        const __bf16* src = w1 + (size_t)(32 * j + 16 * ht + n) * C + 64 * (ks >> 1) + 16 * g + 8 * (ks & 1);
        for (int e = 0; e < 8; ++e) v[e] = src[e];
    } else {                                                            // stage 2: W2'[channel of (ot, row n)][hidden units of k-slot (g, .)]
        const int ot = f - 2 * G::KS;
        const int ch = 64 * (ot >> 2) + 16 * (n >> 2) + 4 * (ot & 3) + (n & 3);
        const __bf16* src = w2 + (size_t)ch * HID + 32 * j;
        for (int e = 0; e < 8; ++e) v[e] = src[16 * (e >> 2) + 4 * g + (e & 3)];
    }
    __bf16* dst = wpk + i * 8;
    for (int e = 0; e < 8; ++e) dst[e] = v[e];
}

int nat_mlp_lds_bytes(int C, int HID) { return nmlp::NS * nmlp::Geo<256>::CHUNK + nmlp::Geo<256>::XCH + HID * 8 + C * 4; }

bool nat_mlp_supported(long long M, int C, int HID) {
    return C == 256 && M > 0 && M % nmlp::ROWS == 0 && HID >= 64 && HID % nmlp::HC == 0 && HID <= 2048 && nat_mlp_lds_bytes(C, HID) <= 160 * 1024;
}

int nat_mlp_pack_launch(const void* w1, const void* w2, void* wpk, int C, int HID, hipStream_t stream) {
    if (C != 256) return -1;
    const long long total = (long long)(HID / nmlp::HC) * nmlp::Geo<256>::FR * 64;
    hipLaunchKernelGGL(nat_mlp_pack_kernel<256>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const __bf16*)w1, (const __bf16*)w2, (__bf16*)wpk, HID);
    return (int)hipGetLastError();
}

int nat_mlp_launch(void* s, const void* wpk, const float* hb, const float* b2, float* stats_out, long long M, int C, int HID, float eps, hipStream_t stream) {
    if (!nat_mlp_supported(M, C, HID)) return -1;
    nmlp::Params p{};
    p.s = (__bf16*)s; p.wpk = (const __bf16*)wpk; p.hb = hb; p.b2 = b2; p.stats_out = stats_out;
    p.M = (int)M; p.NCH = HID / nmlp::HC; p.eps = eps; p.inv_c = 1.0f / (float)C;
    const int lds = nat_mlp_lds_bytes(C, HID);
    static DeviceOnce attr;
    if (const int e = dynamic_lds_once(attr, (const void*)nmlp::nat_mlp_kernel<256>, 160 * 1024)) return e;
    int cus = device_cu_count();
    if (!cus) return -2;
    const int nblk = (int)(M / nmlp::ROWS);
    const int grid = nblk < cus ? nblk : cus;
    hipLaunchKernelGGL(nmlp::nat_mlp_kernel<256>, dim3(grid), dim3(nmlp::NTHR), lds, stream, p);
    return (int)hipGetLastError();
}

}  // namespace ppn
