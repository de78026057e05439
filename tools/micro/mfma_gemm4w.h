// mfma_gemm4w.h — EXPERIMENT (round 5, not shipped: measured slower, see the results at the end of this comment) — a second bf16 MFMA GEMM core for gfx950: the 256 x 256 x 64 tile on FOUR waves, one per SIMD, each owning a
// 128 x 128 quarter of the tile with its 256 accumulator registers in the accumulation half of the register file
// (one wave per SIMD may hold 512 registers: 256 AGPR accumulators + fragments, addresses and epilogue values in the VGPRs).
//
// Why a second core (round 5): mfma_gemm.h's 8-wave form (two waves per SIMD, 128 x 64 per wave) reads 28 KB of LDS fragments per
// wave and k-tile — 224 KB per workgroup for 2 048 cycles of matrix work per SIMD: LDS (128 B / clock) is 88 % as busy as the matrix
// pipe, and its 8 barriers per k-tile lock the two wave groups' phases together.  A 128 x 128 wave tile reads 32 KB per wave and
// k-tile for TWICE the MFMAs (128 KB per workgroup: 50 % of the matrix time), needs 4 barriers per k-tile, and — with a whole SIMD's
// register file — keeps TWO fragment sets per operand, so every LDS read is issued a quarter k-tile (512 matrix cycles) ahead of
// its first use and nothing in the loop waits for LDS.
//
//   C[m][n] = epilogue( sum_k A[m][k] * B[n][k] )       A [M][lda] row-major activations; B [N][K] weights (torch Linear layout)
//
// Pipeline.  A k-tile is four STEPS, one per 64 x 64 quadrant (A part, B part) of the wave's output, in the order (0,0) (0,1) (1,1)
// (1,0): consecutive quadrants share one part, so a step needs ONE new part = 8 fragment reads (4 tiles x 2 k-halves) for its 32
// MFMAs.  A "unit" is a quarter of a k-tile's LDS image: one part of one operand of all four waves (128 rows x 128 B = 16 KB, four
// 16-byte LDS-DMA instructions per wave).  Per step, in this order:
//     counted wait: this wave's copies of the unit READ in this step have landed          (issued 7 steps ago)
//     s_waitcnt lgkmcnt(0): the fragment reads issued in the previous step are complete   (=> that unit is free, its fragments usable)
//     s_barrier
//     LDS-DMA of the unit the previous step read, for two k-tiles later                   (7 steps before it is read: 112 KB in flight per CU)
//     8 fragment reads of the part the NEXT step needs, into the register set that went dead a step ago
//     32 MFMAs on the fragments read one / two steps ago
//   reads:  step 0: B1(g)   step 1: A1(g)   step 2: A0(g+1)   step 3: B0(g+1)
//   DMA:    step 0: B0(g+2) step 1: B1(g+2) step 2: A1(g+2)   step 3: A0(g+3)
// RAW (LDS-DMA -> ds_read): every wave's counted vmcnt precedes the barrier, the read follows it.  WAR (ds_read -> LDS-DMA): every
// wave's lgkmcnt(0) precedes the barrier, the copy follows it.  Register sets: A0 always in fa[0], A1 in fa[1]; B0(g) in fb[g & 1],
// B1(g) in fb[~g & 1] (the k-tile function is instantiated for both parities).  Vector-memory waits are counted at run time
// (`issued - mark`, as in nat_gemm.hip): stores of an epilogue and copies share one in-order counter.
// The LDS image, XOR swizzle and fragment maps are mfma_gemm.h's (conflict-free ds_read_b128).
//
// RESULT (tools/micro/gemm4w_bench, one MI355X, random operands, both kernels alternating in one process; gpurun_out/r05/gemm4w_v1.txt,
// profiles/r05_gemm4w_experiment.txt): correct on every shape, and SLOWER than the 8-wave core — 4096^3 1011 against 1286 TF/s, 8192^3 1115
// against 1337, the NAT projection shapes 0.93 - 1.24 x its time.  A k-tile takes 2.1 us for 2 048 matrix cycles per SIMD: with ONE wave
// per SIMD every instruction that is not an MFMA sits in the matrix pipe's issue stream — an LDS-DMA instruction costs its wave ~60
// cycles of issue (the CU's texture-address path takes 16 cycles per 1 KiB piece and the four waves issue theirs together), 16 of them
// per wave and k-tile, plus ~40 scalar instructions of address arithmetic and the counted-wait tree per step; in the 8-wave form the
// SIMD's other wave computes through those stalls.  (Toolchain notes: hipcc spread accumulators and fragments over both register
// halves and shuffled them with 964 v_accvgpr moves until the MFMAs were assembly with "+a" / "v" constraints; a branch that selects
// between two k-tile bodies doubles the 256 accumulators — 1 KB of scratch per lane — so the k-loop is straight-line for an even
// number of k-tiles.)  What it would take: DMA issued by waves that do not compute, which one kernel's uniform register allocation
// (512 per compute wave) leaves no room for.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace ppn {
namespace g4 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define PPN_G4_INL __attribute__((always_inline))

constexpr int BM = 256, BN = 256, BK = 64, NTHREADS = 256;
constexpr int TILE_BYTES = 256 * 128;          // one operand of one k-tile: 256 rows x 64 bf16
constexpr int BUF_BYTES = 2 * TILE_BYTES;      // A | B
constexpr int LDS_BYTES = 2 * BUF_BYTES;       // two k-tiles: 128 KiB

enum Epi { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_ACCUM = 2, EPI_BIAS_RELU = 3 };

struct Params {
    const __bf16* A;
    const __bf16* B;
    __bf16* C;
    const float* bias;     // [N] float32 (EPI_ACCUM: may be null)
    int M, N, K;           // M % 256 == 0, N % 256 == 0, K % 128 == 0 (an even number of k-tiles)
    int lda, ldc;          // elements
};

template <int V> using I = std::integral_constant<int, V>;

__device__ __forceinline__ void glds16(const void* g, unsigned char* lds_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_uniform, 16, 0, 0);
}

// s_waitcnt vmcnt(n'), n' = n rounded down to even and capped at 62 (n wave-uniform): a tree of scalar branches over immediates
__device__ __forceinline__ void wait_vm(int n) {
    if (n == 24) { asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); return; }          // the k-loop's steady state: six units in flight
    const int h = n >= 62 ? 31 : (n < 0 ? 0 : n >> 1);
#define PPN_G4_W(v) asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory")
#define PPN_G4_W2(lo, a, b) do { if (h == (lo)) PPN_G4_W(a); else PPN_G4_W(b); } while (0)
    if (h < 16) {
        if (h < 8) {
            if (h < 4) { if (h < 2) PPN_G4_W2(0, 0, 2); else PPN_G4_W2(2, 4, 6); }
            else { if (h < 6) PPN_G4_W2(4, 8, 10); else PPN_G4_W2(6, 12, 14); }
        } else {
            if (h < 12) { if (h < 10) PPN_G4_W2(8, 16, 18); else PPN_G4_W2(10, 20, 22); }
            else { if (h < 14) PPN_G4_W2(12, 24, 26); else PPN_G4_W2(14, 28, 30); }
        }
    } else {
        if (h < 24) {
            if (h < 20) { if (h < 18) PPN_G4_W2(16, 32, 34); else PPN_G4_W2(18, 36, 38); }
            else { if (h < 22) PPN_G4_W2(20, 40, 42); else PPN_G4_W2(22, 44, 46); }
        } else {
            if (h < 28) { if (h < 26) PPN_G4_W2(24, 48, 50); else PPN_G4_W2(26, 52, 54); }
            else { if (h < 30) PPN_G4_W2(28, 56, 58); else PPN_G4_W2(30, 60, 62); }
        }
    }
#undef PPN_G4_W2
#undef PPN_G4_W
}

__device__ __forceinline__ float gelu_erf(float x) {                      // A&S 7.1.26, as mfma_gemm.h
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.0f - poly * __expf(-z * z);
    return 0.5f * x + 0.5f * fabsf(x) * e;
}

struct TileSrc { int m0, n0; };

template <int EPI>
__global__ __launch_bounds__(NTHREADS) void gemm4w_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = p.N / BN, tiles_m = p.M / BM;
    const int nblk = tiles_m * tiles_n;
    const int nk = p.K / BK;

    // ---- staging.  A unit u of an operand = its rows with bit 6 == u (what the wave rows / columns read as their part u): row
    // blocks (8 rows, 1 KiB) {u * 8 + i, 16 + u * 8 + i}; wave w copies blocks (j >> 1) * 16 + u * 8 + (j & 1) * 4 + w, j = 0..3 —
    // always of its own parity, so the swizzle key ((block & 1) * 4 + (lane >> 4)) & 7 is one per-lane constant.
    const int srow = lane >> 3;
    const uint32_t lchunk = (uint32_t)(((lane & 7) ^ (((wave & 1) * 4 + (srow >> 1)) & 7)) * 16);
    const uint32_t lofs_a = (uint32_t)srow * (uint32_t)p.lda * 2u + lchunk;
    const uint32_t lofs_b = (uint32_t)srow * (uint32_t)p.K * 2u + lchunk;
    auto tile_src = [&](int v, TileSrc& t) PPN_G4_INL {
        const int q = nblk >> 3, r = nblk & 7, x = v & 7;              // XCD-aware order (mfma_gemm.h)
        const int id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (v >> 3);
        t.m0 = (id / tiles_n) * BM;
        t.n0 = (id % tiles_n) * BN;
    };
    const char* baseA = reinterpret_cast<const char*>(p.A);
    const char* baseB = reinterpret_cast<const char*>(p.B);
    int issued = 0;
    // op 0: A, 1: B; copy j (of 4) of unit u of k-tile kt of tile t into buffer par
    auto stage1 = [&](int op, int u, const TileSrc& t, int kt, int par, int j) PPN_G4_INL {
        const int blk = (j >> 1) * 16 + u * 8 + (j & 1) * 4 + wave;
        if (op == 0) {
            const char* ub = baseA + ((size_t)(t.m0 + blk * 8) * p.lda + kt * BK) * 2;
            glds16(ub + lofs_a, lds + par * BUF_BYTES + blk * 1024);
        } else {
            const char* ub = baseB + ((size_t)(t.n0 + blk * 8) * p.K + kt * BK) * 2;
            glds16(ub + lofs_b, lds + par * BUF_BYTES + TILE_BYTES + blk * 1024);
        }
        issued += 1;
    };
    auto stage = [&](int op, int u, const TileSrc& t, int kt, int par) PPN_G4_INL {
#pragma unroll
        for (int j = 0; j < 4; ++j) stage1(op, u, t, kt, par, j);
    };

    // ---- fragments
    const int frow = lane & 15, fq = lane >> 4, fswz = frow >> 1;
    int f_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) f_rd[kk] = frow * 128 + (((kk * 4 + fq) ^ fswz) << 4);
    f32x4 acc[2][4][2][4];                                           // [A part][m tile][B part][n tile]
    bf16x8 fa[2][4][2], fb[2][4][2];                                 // [register set][tile][k half]
    auto load_a = [&](auto SET_, int par, int part) PPN_G4_INL {
        constexpr int SET = decltype(SET_)::value;
        const unsigned char* base = lds + par * BUF_BYTES + (wr * 128 + part * 64) * 128;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fa[SET][mt][kk] = *reinterpret_cast<const bf16x8*>(base + f_rd[kk] + mt * 16 * 128);
    };
    auto load_b = [&](auto SET_, int par, int part) PPN_G4_INL {
        constexpr int SET = decltype(SET_)::value;
        const unsigned char* base = lds + par * BUF_BYTES + TILE_BYTES + (wc * 128 + part * 64) * 128;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) fb[SET][nt][kk] = *reinterpret_cast<const bf16x8*>(base + f_rd[kk] + nt * 16 * 128);
    };
    // One MFMA of a step, as assembly with the accumulator constrained to the accumulation registers ("a") and the fragments to the
    // vector registers ("v"): left to itself hipcc spreads accumulators and fragments over both halves of the file and moves them
    // with v_accvgpr_* between the MFMAs (964 moves for 512 MFMAs in the first build).  idx = kk * 16 + mt * 4 + nt: an accumulator
    // is touched once per k-half, 16 MFMAs apart (no dependent back-to-back pair).  Hazards: ds_read -> MFMA operand is covered by
    // the step's lgkmcnt(0); MFMA -> MFMA on the same accumulator needs no wait states; the epilogue's reads are padded there.
    auto mfma1 = [&](auto AP_, auto BP_, auto SA_, auto SB_, auto ZERO_, auto IDX_) PPN_G4_INL {
        constexpr int AP = decltype(AP_)::value, BP = decltype(BP_)::value, SA = decltype(SA_)::value, SB = decltype(SB_)::value;
        constexpr int idx = decltype(IDX_)::value, kk = idx >> 4, mt = (idx >> 2) & 3, nt = idx & 3;
        f32x4& c = acc[AP][mt][BP][nt];                               // (named here: clang does not capture a variable that only an asm operand inside `if constexpr` mentions)
        const bf16x8& x = fb[SB][nt][kk];
        const bf16x8& y = fa[SA][mt][kk];
        if constexpr (decltype(ZERO_)::value != 0 && kk == 0)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(x), "v"(y));
        else
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(x), "v"(y));
    };
    // one fragment read of a part: i = tile * 2 + k half
    auto read_a1 = [&](auto SET_, int par, int part, auto I_) PPN_G4_INL {
        constexpr int SET = decltype(SET_)::value, i = decltype(I_)::value, mt = i >> 1, kk = i & 1;
        fa[SET][mt][kk] = *reinterpret_cast<const bf16x8*>(lds + par * BUF_BYTES + (wr * 128 + part * 64 + mt * 16) * 128 + f_rd[kk]);
    };
    auto read_b1 = [&](auto SET_, int par, int part, auto I_) PPN_G4_INL {
        constexpr int SET = decltype(SET_)::value, i = decltype(I_)::value, nt = i >> 1, kk = i & 1;
        fb[SET][nt][kk] = *reinterpret_cast<const bf16x8*>(lds + par * BUF_BYTES + TILE_BYTES + (wc * 128 + part * 64 + nt * 16) * 128 + f_rd[kk]);
    };

    // ---- this workgroup's stream of k-tiles: g = 0 .. total - 1 over its tiles
    const int G = gridDim.x;
    const int my_tiles = (nblk - (int)blockIdx.x + G - 1) / G;
    const int total = my_tiles * nk;
    TileSrc cur, nxt, nx2;
    tile_src(blockIdx.x, cur);
    nxt = cur; nx2 = cur;
    if (my_tiles > 1) tile_src(blockIdx.x + G, nxt);
    if (my_tiles > 2) tile_src(blockIdx.x + 2 * G, nx2);
    int g = 0;
    int mark[8];                                                      // `issued` right after the copy of stream unit i (slot i & 7)
    // unit (op, u) of stream k-tile g + d, seen from k-tile kt of tile `cur` (d <= 3, nk >= 2: at most two tiles ahead)
    struct Ahead { TileSrc t; int k; bool on; };
    auto ahead = [&](int kt, int d) PPN_G4_INL {
        Ahead a; a.on = g + d < total; a.t = cur; a.k = kt + d;
        if (a.k >= nk) { a.k -= nk; a.t = nxt; if (a.k >= nk) { a.k -= nk; a.t = nx2; } }
        return a;
    };

    // prologue: k-tiles 0 and 1 whole, in consumption order A0 B0 B1 A1 | A0 B0 B1 A1 (stream units 0..7)
    stage(0, 0, cur, 0, 0); mark[0] = issued;
    stage(1, 0, cur, 0, 0); mark[1] = issued;
    stage(1, 1, cur, 0, 0); mark[2] = issued;
    stage(0, 1, cur, 0, 0); mark[3] = issued;
    stage(0, 0, cur, 1, 1); mark[4] = issued;
    stage(1, 0, cur, 1, 1); mark[5] = issued;
    stage(1, 1, cur, 1, 1); mark[6] = issued;
    stage(0, 1, cur, 1, 1); mark[7] = issued;
    // step -2: A0(0) -> fa[0]
    wait_vm(issued - mark[0]);
    __builtin_amdgcn_s_barrier();
    load_a(I<0>{}, 0, 0);
    // step -1: copy A0(2) over it, B0(0) -> fb[0]
    wait_vm(issued - mark[1]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (2 < total) { if (2 < nk) stage(0, 0, cur, 2, 0); else stage(0, 0, nxt, 2 - nk, 0); }
    mark[0] = issued;
    load_b(I<0>{}, 0, 0);

#define PPN_G4_SYNC(slot) do {                                     \
        wait_vm(issued - mark[slot]);                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
        __builtin_amdgcn_s_barrier();                              \
        __builtin_amdgcn_sched_barrier(0);                         \
    } while (0)

    // One step: 32 MFMAs in 16 pairs; behind pairs 0-3 one copy each of the unit being restaged, behind pairs 4-11 one fragment
    // read each of the part the next step needs — issued in the shadow of the matrix pipe, in exactly this order (sched_barrier).
    auto step = [&](auto AP_, auto BP_, auto SA_, auto SB_, auto ZERO_, const Ahead& ah, int op, int u, int dpar, auto&& read1, bool do_read) PPN_G4_INL {
        auto pair = [&](auto P_) PPN_G4_INL {
            constexpr int P = decltype(P_)::value;
            mfma1(AP_, BP_, SA_, SB_, ZERO_, I<2 * P>{});
            mfma1(AP_, BP_, SA_, SB_, ZERO_, I<2 * P + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (P < 4) { if (ah.on) stage1(op, u, ah.t, ah.k, dpar, P); }
            else if constexpr (P < 12) { if (do_read) read1(I<P - 4>{}); }
            __builtin_amdgcn_sched_barrier(0);
        };
        pair(I<0>{}); pair(I<1>{}); pair(I<2>{}); pair(I<3>{}); pair(I<4>{}); pair(I<5>{}); pair(I<6>{}); pair(I<7>{});
        pair(I<8>{}); pair(I<9>{}); pair(I<10>{}); pair(I<11>{}); pair(I<12>{}); pair(I<13>{}); pair(I<14>{}); pair(I<15>{});
    };
    // One k-tile.  PAR = g & 1: its LDS buffer, and the register set of its B0 fragments.  ZERO: the tile's first k-tile (zero C operand).
    auto ktile = [&](auto PAR_, auto ZERO_, int kt) PPN_G4_INL {
        constexpr int PAR = decltype(PAR_)::value;
        // stream unit indices: A0(g) = 4g, B0(g) = 4g + 1, B1(g) = 4g + 2, A1(g) = 4g + 3 -> mark slots (4 PAR + i) & 7
        const bool more = g + 1 < total;
        // ---- step 0: quadrant (0,0); read B1(g); copy B0(g+2)
        PPN_G4_SYNC((4 * PAR + 2) & 7);
        { const Ahead ah = ahead(kt, 2);
          step(I<0>{}, I<0>{}, I<0>{}, I<PAR>{}, ZERO_, ah, 1, 0, PAR, [&](auto i_) PPN_G4_INL { read_b1(I<PAR ^ 1>{}, PAR, 1, i_); }, true); }
        mark[(4 * PAR + 1) & 7] = issued;
        // ---- step 1: quadrant (0,1); read A1(g); copy B1(g+2)
        PPN_G4_SYNC((4 * PAR + 3) & 7);
        { const Ahead ah = ahead(kt, 2);
          step(I<0>{}, I<1>{}, I<0>{}, I<PAR ^ 1>{}, ZERO_, ah, 1, 1, PAR, [&](auto i_) PPN_G4_INL { read_a1(I<1>{}, PAR, 1, i_); }, true); }
        mark[(4 * PAR + 2) & 7] = issued;
        // ---- step 2: quadrant (1,1); read A0(g+1); copy A1(g+2)
        PPN_G4_SYNC((4 * PAR + 4) & 7);
        { const Ahead ah = ahead(kt, 2);
          step(I<1>{}, I<1>{}, I<1>{}, I<PAR ^ 1>{}, ZERO_, ah, 0, 1, PAR, [&](auto i_) PPN_G4_INL { read_a1(I<0>{}, PAR ^ 1, 0, i_); }, more); }
        mark[(4 * PAR + 3) & 7] = issued;
        // ---- step 3: quadrant (1,0); read B0(g+1); copy A0(g+3)
        PPN_G4_SYNC((4 * PAR + 5) & 7);
        { const Ahead ah = ahead(kt, 3);
          step(I<1>{}, I<0>{}, I<1>{}, I<PAR>{}, ZERO_, ah, 0, 0, PAR ^ 1, [&](auto i_) PPN_G4_INL { read_b1(I<PAR ^ 1>{}, PAR ^ 1, 0, i_); }, more); }
        mark[(4 * PAR + 4) & 7] = issued;
        ++g;
    };

    // ---- epilogue of one tile (not overlapped in this form): a lane holds 4 consecutive n of row frow in each 16-wide column tile;
    // v_permlane16_swap between neighbouring lane rows leaves 8 consecutive n of ONE tile per lane: 16-byte stores
    const int qcol = (fq & 1) * 16 + (fq >> 1) * 8;
    auto epilogue = [&](int m0, int n0) PPN_G4_INL {
        // the last MFMAs are assembly: hipcc pads nothing behind them (MFMA result -> any reader: 12+ wait states)
        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        float bq[2][2][8];
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int n = n0 + wc * 128 + bp * 64 + pr * 32 + qcol;
                if (EPI != EPI_ACCUM) {
                    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
                    bq[bp][pr][0] = b0.x; bq[bp][pr][1] = b0.y; bq[bp][pr][2] = b0.z; bq[bp][pr][3] = b0.w;
                    bq[bp][pr][4] = b1.x; bq[bp][pr][5] = b1.y; bq[bp][pr][6] = b1.z; bq[bp][pr][7] = b1.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bq[bp][pr][e] = 0.f;
                }
            }
#pragma unroll
        for (int ap = 0; ap < 2; ++ap)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int m = m0 + wr * 128 + ap * 64 + mt * 16 + frow;
                __bf16* crow = p.C + (size_t)m * p.ldc + n0 + wc * 128 + qcol;
                uint4 prev[2][2];
                if (EPI == EPI_ACCUM) {
#pragma unroll
                    for (int bp = 0; bp < 2; ++bp)
#pragma unroll
                        for (int pr = 0; pr < 2; ++pr) prev[bp][pr] = *reinterpret_cast<const uint4*>(crow + bp * 64 + pr * 32);
                }
#pragma unroll
                for (int bp = 0; bp < 2; ++bp)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        f32x4 x = acc[ap][mt][bp][2 * pr], y = acc[ap][mt][bp][2 * pr + 1];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[r]), __float_as_uint(y[r]), false, false);
                            x[r] = __uint_as_float(t[0]); y[r] = __uint_as_float(t[1]);
                        }
                        float o[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] += bq[bp][pr][e];
                        if (EPI == EPI_ACCUM) {
                            const bf16x8 s8 = __builtin_bit_cast(bf16x8, prev[bp][pr]);
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] += (float)s8[e];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            if (EPI == EPI_BIAS_GELU) o[e] = gelu_erf(o[e]);
                            if (EPI == EPI_BIAS_RELU) o[e] = fmaxf(o[e], 0.f);
                        }
                        const bf16x8 w = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3], (__bf16)o[4], (__bf16)o[5], (__bf16)o[6], (__bf16)o[7]};
                        *reinterpret_cast<bf16x8*>(crow + bp * 64 + pr * 32) = w;
                    }
            }
        issued += 32;
    };

    // Straight-line control flow around the accumulators: a branch that selects between two k-tile bodies joins 256 accumulator
    // phis and hipcc keeps both copies alive (1 KB of scratch per lane in the first build).  nk is even (host), so a tile starts
    // on an even stream k-tile and the two parities alternate without a test.
    for (int it = 0; it < my_tiles; ++it) {
        ktile(I<0>{}, I<1>{}, 0);
        ktile(I<1>{}, I<0>{}, 1);
        for (int kt = 2; kt < nk; kt += 2) {
            ktile(I<0>{}, I<0>{}, kt);
            ktile(I<1>{}, I<0>{}, kt + 1);
        }
        epilogue(cur.m0, cur.n0);
        cur = nxt; nxt = nx2;
        if (it + 3 < my_tiles) tile_src(blockIdx.x + (it + 3) * G, nx2);
    }
#undef PPN_G4_SYNC
}

}  // namespace g4
}  // namespace ppn
