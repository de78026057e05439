// na2d_halo16.hip — the 16 x 16-query form of the matrix-core neighbourhood attention (na2d_mfma.hip: same arithmetic, block for
// block) as a PERSISTENT kernel whose halo staging never passes through registers.
//
// Why: na2d_mfma_kernel<16> stages a 22 x 22 key halo per workgroup through VGPRs (load, wait, ds_write, barrier) and only then
// computes; 80 KB of LDS hold two workgroups per CU, their staging phases collide, and the staging's address arithmetic is a third
// of the kernel's vector instructions — it ran at 0.3 of the HBM roofline with the matrix and vector pipes each a third busy.
// Here one workgroup of 16 waves owns a CU for the whole launch and walks its share of the (tile, head) items:
//   stage(i + 1)   66 KB of K and V rows, copied HBM -> LDS by global_load_lds (LDS-DMA: 16 B per lane, lane-linear in LDS, no
//                  VGPRs), 2 - 3 pairs of wave instructions per wave, into the OTHER of two LDS buffers
//   compute(i)     wave w = block (w / 4, w % 4) of the tile: 10 + 15 MFMAs, the softmax of 16 queries (na2d_mfma.hip's block)
// with ONE barrier per tile (after it every wave has finished tile i - 1, so its buffer may be overwritten; the stage of tile i was
// issued a whole tile earlier).  The per-lane part of a piece's address (halo row t, column sc, chunk) does not depend on the tile;
// what does comes from a 64-byte descriptor per tile (byte offsets of its halo, query and output origins, its geometry) that a
// preparation kernel writes ONCE PER GEOMETRY — the launcher keeps the tables — so a tile costs one scalar load instead of nine integer
// divisions, a piece of the halo two compares (its class: stored token / padded token / zeros) before a DMA from a uniform base, and a
// launch neither allocates nor prepares anything (a captured HIP graph holds the kernel alone).
//
// LDS (138,752 B): 2 x { K image 22 x 24 slots x 64 B | V image (both with the 32-byte halves of a row swapped on every other
// group of 4 slots: conflict-free fragment reads) | 4 slots } | BT, BTM 2 x 16 x 24 f32 of the workgroup's head.  The column pitch is 24 slots for 22 loaded: a block reads 16 slots from
// column co <= 12, so its last 4 run into the next halo row (the V image / the zero slots behind the last row) — finite values
// whose logits carry the window mask, i.e. probability exactly 0.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include "ppn_device.h"
#include "ppn_kernels.h"

namespace ppn {

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int MK = 7, MN = 3, MHD = 32;
constexpr int KB = 64;                 // bytes per staged key row (32 bf16)
constexpr int RT = 16, HROWS = RT + 6, PC = 24;
constexpr int SLOTS = HROWS * PC;      // 528 = 33 x 16
constexpr int IMG = SLOTS * KB;
constexpr int TAIL = 4 * KB;
constexpr int BUF = 2 * IMG + TAIL;                  // a tile: K image | V image | 4 zero slots
constexpr int BT_ROWS = 16;
// The 16 queries of an MFMA block are TQY x TQX of the tile: 4 x 4 (the union of their windows is 10 halo rows x 10 columns: 10 key
// tiles of 16 column slots, 100 of 160 slots hold a key of some query) or 2 x 8 (8 rows x 14 columns: 8 key tiles, 112 of 128 — a fifth
// less of everything that is per key slot: logit MFMAs, exponentials, maxima, conversions, K / V / bias reads).
template <int TQY_, int TQX_>
struct Blk {
    static constexpr int TQY = TQY_, TQX = TQX_;
    static constexpr int HR = TQY + MK - 1;              // key tiles (halo rows) per block
    static constexpr int NBY = RT / TQY, NBX = RT / TQX; // blocks of a tile
    // Bias tables, [row][column].  A lane reads row (r0 - ti0 + 6) + t - jy, columns (c0 - tj0 + 6) + 4g + r - jx, r = 0 .. 3: the
    // (t, r) part is an immediate (consecutive r = consecutive words: ds_read2_b32 fills an accumulator's register pair in place),
    // the (jy, jx, g) part the lane's own.  A 32-bit LDS read is banked (address / 4) mod 32 over the lane groups {0-31}, {32-63}
    // (g = 0, 1 / 2, 3), so inside a group the lanes' parts must fall into different banks unless they are equal:
    //   4 x 4: pitch 24 — 4 g - jx takes 8 values, the 4 query rows sit 24 = -8 banks apart;
    //   2 x 8: pitch 48 — 4 g - jx takes 12 values, the 2 query rows sit 48 = 16 banks apart.  (Pitch 32 put both query rows of a
    //          block into the same banks: 16.8 M conflict cycles per launch at 64 x 64, a third of the kernel's LDS time; a
    //          transposed table at pitch 17 is conflict-free too, but hipcc pairs neighbouring words into ds_read2_b32 and then
    //          moves 29 registers per block into place.)  Columns start BT_COL0 = 8 in: a lane's column index runs from -7.
    static constexpr int BT_PITCH = TQX == 4 ? 24 : 48, BT_COL0 = TQX == 4 ? 0 : 8;
    static constexpr int BT_SIZE = BT_ROWS * BT_PITCH;                   // floats per table
    static constexpr int TBL = 2 * BT_SIZE * 4;
    static __device__ __forceinline__ constexpr int bt_at(int a, int b) { return a * BT_PITCH + b + BT_COL0; }   // element (row a, column b)
    static constexpr int BT_DROW = BT_PITCH, BT_DCOL = 1;
    static constexpr int LDS_TOTAL = 2 * BUF + TBL;      // two tiles | the head's two bias tables
    static_assert(TQY * TQX == 16 && HR % 2 == 0 && NBY * NBX == 16 && RT - TQX + 16 <= PC + 4 && TBL % 16 == 0, "block shape");
};
constexpr int NPAIR = SLOTS / 16;      // wave instructions per image
constexpr int NW = 16, NTHR = NW * 64;
constexpr int PPW = (NPAIR + NW - 1) / NW;
constexpr int DESC = 32;               // ints per tile descriptor
static_assert(SLOTS % 16 == 0 && (PC & 7) == 0, "staging layout");
static_assert(PPW * 10 <= 32 && HROWS + 2 <= 32 && PC <= 32, "packed staging coordinates");

// descriptor of a tile (what of it does not depend on the head): byte offsets of the halo origin's k row (head 0), of the tile
// origin's q row and of its output row; the halo extent and its real part (virtual padding), packed; the tile's geometry
enum { D_KV = 0, D_Q = 2, D_O = 4, D_EXT = 6, D_TY0, D_TX0, D_HS, D_WS, D_HQ, D_WQ, D_R0, D_C0, D_VALID, D_ROW = 16, D_COL = 24 };
// D_ROW + by / D_COL + bx: the geometry of block row by / block column bx of the tile, so that a wave derives its block's from two
// words instead of a dozen clamps and compares: window origin r0 (16 bits) | its place in the halo ro << 16 (4) | bias-table row
// r0 - ti0 + 6 << 20 (4) | the block row holds queries << 24 | the image ends inside it << 25 | no query's window is clamped << 26
enum { B_LIVE = 1 << 24, B_CUT = 1 << 25, B_INT = 1 << 26 };

__device__ __forceinline__ int clampm(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// LDS-DMA of 16 bytes per lane, base + off -> lds_uniform + 16 * lane.  Written as assembly on purpose: the compiler orders LDS reads
// it cannot tell apart from a __builtin_amdgcn_global_load_lds in flight behind that copy (s_waitcnt vmcnt(0) — before the transposed
// V reads here), which would make every tile wait for the NEXT tile's staging; what it does not see it does not wait for, and its own
// counted waits only become stricter by operations it does not know of (vmcnt retires in order).  M0 is otherwise unused here.
__device__ __forceinline__ void dma16(const void* base_uniform, unsigned off, unsigned lds_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base_uniform), "s"(lds_uniform) : "memory");
}
__device__ __forceinline__ float max3(float a, float b, float c) {        // (fmaxf would canonicalise every MFMA result first)
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// maximum of N registers as a tree of v_max3 (40 -> 14 -> 5 -> 2 -> 1, 32 -> 11 -> 4 -> 2 -> 1: four deep)
template <int N>
__device__ __forceinline__ float max_tree(const float (&v)[N]) {
    if constexpr (N == 1) return v[0];
    else if constexpr (N == 2) return max3(v[0], v[1], v[1]);
    else {
        constexpr int M = (N + 2) / 3;
        float r[M];
#pragma unroll
        for (int i = 0; i < N / 3; ++i) r[i] = max3(v[3 * i], v[3 * i + 1], v[3 * i + 2]);
        if constexpr (N % 3 == 1) r[M - 1] = v[N - 1];
        if constexpr (N % 3 == 2) r[M - 1] = max3(v[N - 2], v[N - 1], v[N - 1]);
        return max_tree<M>(r);
    }
}
}  // namespace

// One descriptor per tile: it depends on the launch's geometry only (not on the data), so a launcher-side cache keeps it per shape.
__global__ __launch_bounds__(256) void na2d_halo16_prep_kernel(int* __restrict__ desc, int heads, int H, int W, int Hr, int Wr, int dil, int tiles_y, int tiles_x,
                                                               int total_tiles, int padded, int tqy, int tqx) {
    const int gtile = (int)blockIdx.x * 256 + threadIdx.x;
    if (gtile >= total_tiles) return;
    const int ntiles = tiles_y * tiles_x;
    const int bz = gtile / ntiles, tile_id = gtile - bz * ntiles;
    const int b = bz / (dil * dil), g2 = bz % (dil * dil);
    const int gi = g2 / dil, gj = g2 % dil;
    const int hs = (H - gi + dil - 1) / dil, ws = (W - gj + dil - 1) / dil;                     // key sub-image of this dilation group
    const int hq = gi < Hr ? (Hr - gi + dil - 1) / dil : 0, wq = gj < Wr ? (Wr - gj + dil - 1) / dil : 0;   // its real part: the queries
    const int ty0 = (tile_id / tiles_x) * RT, tx0 = (tile_id % tiles_x) * RT;
    const int valid = (ty0 < hq && tx0 < wq) ? 1 : 0;                                           // groups differ by one row / column
    int* d = desc + (size_t)gtile * DESC;
    if (!valid) {
        for (int i = 0; i < DESC; ++i) d[i] = 0;
        return;
    }
    const int Hs = padded ? Hr : H, Ws = padded ? Wr : W;                                       // stored token grid
    const int R0 = clampm(ty0 - MN, 0, hs - MK), C0 = clampm(tx0 - MN, 0, ws - MK);            // halo origin
    const int ymax = min(ty0 + RT - 1, hq - 1), xmax = min(tx0 + RT - 1, wq - 1);
    const int NR = clampm(ymax - MN, 0, hs - MK) + MK - R0, NC = clampm(xmax - MN, 0, ws - MK) + MK - C0;   // halo extent
    const int NRr = padded ? min(NR, hq - R0) : NR, NCr = padded ? min(NC, wq - C0) : NC;       // ... that holds stored tokens
    const unsigned long long tokb = (unsigned long long)3 * heads * MHD * 2;
    const unsigned long long kv = ((unsigned long long)(b * Hs + gi + R0 * dil) * Ws + gj + C0 * dil) * tokb + (unsigned long long)heads * MHD * 2;
    const unsigned long long qo = ((unsigned long long)(b * Hs + gi + ty0 * dil) * Ws + gj + tx0 * dil) * tokb;
    const unsigned long long oo = ((unsigned long long)(b * Hr + gi + ty0 * dil) * Wr + gj + tx0 * dil) * ((unsigned long long)heads * MHD * 2);
    d[D_KV] = (int)(unsigned)kv; d[D_KV + 1] = (int)(unsigned)(kv >> 32);
    d[D_Q] = (int)(unsigned)qo; d[D_Q + 1] = (int)(unsigned)(qo >> 32);
    d[D_O] = (int)(unsigned)oo; d[D_O + 1] = (int)(unsigned)(oo >> 32);
    d[D_EXT] = NR | (NC << 8) | (NRr << 16) | (NCr << 24);
    d[D_TY0] = ty0; d[D_TX0] = tx0; d[D_HS] = hs; d[D_WS] = ws; d[D_HQ] = hq; d[D_WQ] = wq; d[D_R0] = R0; d[D_C0] = C0; d[D_VALID] = 1;
    for (int k = D_ROW; k < DESC; ++k) d[k] = 0;
    for (int k = 0; k < RT / tqy; ++k) {
        const int ti0 = ty0 + k * tqy;
        const int r0 = clampm(ti0 - MN, 0, hs - MK);                                             // the block row's window origin
        if (ti0 < hq)
            d[D_ROW + k] = r0 | ((r0 - R0) << 16) | ((r0 - ti0 + MK - 1) << 20) | B_LIVE | (ti0 + tqy > hq ? B_CUT : 0) |
                           (ti0 >= MN && ti0 + tqy - 1 + MN <= hs - 1 ? B_INT : 0);
    }
    for (int k = 0; k < RT / tqx; ++k) {
        const int tj0 = tx0 + k * tqx;
        const int c0 = clampm(tj0 - MN, 0, ws - MK);
        if (tj0 < wq)
            d[D_COL + k] = c0 | ((c0 - C0) << 16) | ((c0 - tj0 + MK - 1) << 20) | B_LIVE | (tj0 + tqx > wq ? B_CUT : 0) |
                           (tj0 >= MN && tj0 + tqx - 1 + MN <= ws - 1 ? B_INT : 0);
    }
}

struct Tile {
    unsigned long long kv, q, o;
    int ext, ty0, tx0, hs, ws, hq, wq, R0, C0, valid, h, row, col;       // row / col: D_ROW / D_COL of the loading wave's block
};

__device__ __forceinline__ Tile load_tile(const int* __restrict__ desc, int gtile, int h, int by, int bx) {
    const int* d = desc + (size_t)gtile * DESC;
    Tile t;
    t.h = h;
    t.kv = (unsigned long long)(unsigned)d[D_KV] | ((unsigned long long)(unsigned)d[D_KV + 1] << 32);
    t.q = (unsigned long long)(unsigned)d[D_Q] | ((unsigned long long)(unsigned)d[D_Q + 1] << 32);
    t.o = (unsigned long long)(unsigned)d[D_O] | ((unsigned long long)(unsigned)d[D_O + 1] << 32);
    t.ext = d[D_EXT]; t.ty0 = d[D_TY0]; t.tx0 = d[D_TX0]; t.hs = d[D_HS]; t.ws = d[D_WS]; t.hq = d[D_HQ]; t.wq = d[D_WQ];
    t.R0 = d[D_R0]; t.C0 = d[D_C0]; t.valid = d[D_VALID];
    t.row = d[D_ROW + by]; t.col = d[D_COL + bx];
    return t;
}

template <int TQY, int TQX>
__global__ __launch_bounds__(NTHR) void na2d_halo16_kernel(const __bf16* __restrict__ qkv, const __bf16* __restrict__ pad_kv, __bf16* __restrict__ out,
                                                           const float* __restrict__ rpb, const int* __restrict__ desc, int Wr, int Ws, int heads,
                                                           int dil, float scale, int n_items) {
    typedef Blk<TQY, TQX> S;
    constexpr int HR = S::HR;
    extern __shared__ __attribute__((aligned(16))) unsigned char nl[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;                                // MFMA column (query) and lane quarter
    const unsigned tokb = 3u * heads * MHD * 2;                            // bytes of a token's qkv row

    // this workgroup's items (item = tile * heads + head): XCD x (workgroups x, x + 8, ...) owns a contiguous run with the head
    // fastest — the heads of a token share its 128-byte lines, neighbouring tiles share halos — and its workgroups take consecutive
    // items of it at the same time
    const int nx = gridDim.x >> 3, xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
    const int per = (n_items + 7) >> 3;
    const int end = min((xcd + 1) * per, n_items);
    int item = xcd * per + kx;
    if (item >= end) return;                                               // workgroup-uniform
    int gtile = item / heads, h = item - gtile * heads;
    const int step_t = nx / heads, step_h = nx - step_t * heads;           // item + nx without a division per tile

    // Both tile buffers start as zeros, once: a slot that no piece of a tile covers — the 2 padding columns of the 24-slot pitch, the
    // rows and columns beyond a border tile's halo extent, the 4 slots behind each V image — keeps what an EARLIER tile left there
    // (or these zeros): finite K and V values whose logits carry the window mask, i.e. probability exactly 0.  Re-zeroing them per
    // tile was a second pair of DMA instructions per piece (each ~60+ issue cycles) for lanes that are idle in the first.
    for (int i = threadIdx.x; i < 2 * BUF / 16; i += NTHR) reinterpret_cast<uint4*>(nl)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();                                                       // ... before the first piece lands

    // ---- the tile-independent part of this lane's staging pieces.  Pair k = wave + 16 i covers slots 16 k .. 16 k + 15 of both
    // images, this lane the 16 bytes at position lane % 4 of slot 16 k + lane / 4 = halo row t, column sc.  Both images are stored with
    // the 32-byte halves of a row swapped on every other group of 4 slots (the K fragment reads take 16 bytes per lane and the
    // transposed V reads 8 from 16 slots at a 64-byte pitch: slots s and s + 4 would meet in the same banks); the DMA lands
    // lane-linear, so the lane FETCHES the other chunk.  16 | slots per pair and 8 | PC: the group parity is the lane's own.
    const unsigned c16 = (unsigned)((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16;
    unsigned st_pack = 0, st_off[PPW];
    {
        const unsigned rowstep = (unsigned)(Ws * dil) * tokb, colstep = (unsigned)dil * tokb;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int slot = (wave + NW * i) * 16 + (lane >> 2);
            const int t = slot / PC, sc = slot - t * PC;
            st_pack |= ((unsigned)t | ((unsigned)sc << 5)) << (10 * i);
            st_off[i] = (unsigned)t * rowstep + (unsigned)sc * colstep + c16;      // from the halo origin's k row, bytes
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)nl;
    const char* qkvb = reinterpret_cast<const char*>(qkv);
    const unsigned vofs = (unsigned)heads * MHD * 2;

    // A slot inside the tile's halo extent has a source — its token's row or the padded token (virtual padding: k / v = the qkv
    // bias) — and each of the two classes is one DMA under its lanes' mask from a uniform base with a 32-bit lane offset: two compares
    // per pair are all the vector work the staging of a tile costs.  Slots outside the extent are not written (see above).
    auto stage = [&](const Tile& T, unsigned buf) __attribute__((always_inline)) {
        if (!T.valid) return;
        const char* kbase = qkvb + T.kv + (unsigned)T.h * (MHD * 2);
        const char* vbase = kbase + vofs;
        const int NR = T.ext & 255, NC = (T.ext >> 8) & 255, NRr = (T.ext >> 16) & 255, NCr = (T.ext >> 24) & 255;
        unsigned pk2 = st_pack;
        asm volatile("" : "+v"(pk2));                                      // (unpacked per tile: one register across the loop, not six)
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int k = wave + NW * i;
            if (k < NPAIR) {                                               // wave-uniform
                const int t = (pk2 >> (10 * i)) & 31, sc = (pk2 >> (10 * i + 5)) & 31;
                const bool real = t < NRr && sc < NCr, inside = t < NR && sc < NC;
                const unsigned kl = buf + k * 1024, vl = buf + IMG + k * 1024;
                if (real) {
                    dma16(kbase, st_off[i], kl);
                    dma16(vbase, st_off[i], vl);
                } else if (inside) {                                       // only with pad_kv
                    const char* pk = reinterpret_cast<const char*>(pad_kv) + (size_t)(heads + T.h) * (MHD * 2);
                    dma16(pk, c16, kl);
                    dma16(pk + vofs, c16, vl);
                }
            }
        }
    };

    // ---- this wave's block of the tile: rows TQY by .., columns TQX bx ..; lane (j, g): query (jy, jx) of it, quarter g
    const int by = wave / S::NBX, bx = wave % S::NBX;
    const int jy = j / TQX, jx = j % TQX;
    const int ry = by * TQY + jy, rx = bx * TQX + jx;                      // the query's place in the tile
    const unsigned qoff_full = (unsigned)(ry * Ws + rx) * (unsigned)dil * tokb + 16u * g;             // B operand: channels 8g .. 8g+7 of query j
    const unsigned ooff_full = (unsigned)(ry * Wr + rx) * (unsigned)dil * (unsigned)(heads * MHD * 2) + 8u * g;
    const int btl_full = S::bt_at(-jy, -jx + 4 * g) * 4;
    // dead queries (a tile the image cuts) shadow a live one and are never stored
    auto q_request = [&](const Tile& T) __attribute__((always_inline)) -> bf16x8 {
        const int rmax = T.hq - 1 - T.ty0, cmax = T.wq - 1 - T.tx0;
        unsigned qoff = qoff_full;
        if (rmax < RT - 1 || cmax < RT - 1)                                // wave-uniform
            qoff = (unsigned)(clampm(ry, 0, max(rmax, 0)) * Ws + clampm(rx, 0, max(cmax, 0))) * (unsigned)dil * tokb + 16u * g;
        return *reinterpret_cast<const bf16x8*>(qkvb + T.q + (unsigned)T.h * (MHD * 2) + qoff);
    };

    // The two bias tables of a head, in the units of the raw product ([2][16][24] f32: rpb[h] / scale zero-padded; the same inside the
    // centred 7 x 7 window and -1e30 elsewhere = bias AND window mask of a query whose window the border does not clamp), built by the
    // workgroup itself: its items are 32 apart (a multiple of every power-of-two head count), so the head changes rarely or never
    float* const BTL = reinterpret_cast<float*>(nl + 2 * BUF);
    auto build_tables = [&](int hh) __attribute__((always_inline)) {
        for (int i = threadIdx.x; i < 2 * S::BT_SIZE; i += NTHR) {
            const int which = i >= S::BT_SIZE, t = i - which * S::BT_SIZE;
            const int a = t / S::BT_PITCH, b = t % S::BT_PITCH - S::BT_COL0;
            const float v = (a < 13 && b >= 0 && b < 13) ? rpb[(size_t)hh * 169 + a * 13 + b] / scale : 0.f;
            BTL[i] = (which == 0 || (a >= MN && a <= 3 * MN && b >= MN && b <= 3 * MN)) ? v : -1.0e30f;
        }
    };
    int h_tab = h;
    build_tables(h_tab);
    Tile cur = load_tile(desc, gtile, h, by, bx);
    stage(cur, lds0);
    bf16x8 q_next = q_request(cur);
    int bsel = 0;
    bool stored = false;                                                   // wave-uniform: the tile before left two stores in flight
    const int q4 = j >> 2, p4 = j & 3;                                     // transposed read: this lane addresses row q4, columns 4 p4 ..
    const float NEG = -1.0e30f;
    const float sl2 = scale * 1.4426950408889634f;

    while (true) {
        // this wave's pieces of this tile's stage and its q fragment have landed — NOT the last tile's two stores, which are younger
        // than both (vmcnt retires in order; a live block issues exactly two store instructions behind everything it requested, a dead
        // one none): waiting for them too was a store's round trip per tile with all 16 waves idle.  Behind the barrier every other
        // wave's pieces have landed as well, and every wave has left the tile before (whose buffer the next stage fills)
        if (stored) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bf16x8 qf = q_next;
        asm volatile("" : "+v"(qf));
        __syncthreads();
        if (cur.h != h_tab) {                                              // workgroup-uniform, rare: every wave has left the old tables
            h_tab = cur.h;
            build_tables(h_tab);
            __syncthreads();
        }
        const unsigned char* buf = nl + bsel * BUF;
        const int nitem = item + nx;
        const bool more = nitem < end;                                     // workgroup-uniform
        Tile nxt = cur;
        if (more) {
            gtile += step_t;
            h += step_h;
            if (h >= heads) { h -= heads; ++gtile; }
            nxt = load_tile(desc, gtile, h, by, bx);
            stage(nxt, lds0 + (bsel ^ 1) * BUF);
            q_next = q_request(nxt);
        }

        const Tile& T = cur;
        stored = (T.row & T.col & B_LIVE) != 0;
        if (stored) {                                      // wave-uniform: the block holds queries (an invalid tile's words are 0)
            const unsigned char* Kimg = buf;
            const unsigned char* Vimg = buf + IMG;
            const unsigned char* BT = nl + 2 * BUF;
            const int ti0 = T.ty0 + by * TQY, tj0 = T.tx0 + bx * TQX;
            const int hs = T.hs, ws = T.ws, hq = T.hq, wq = T.wq;
            const bool cut = ((T.row | T.col) & B_CUT) != 0;               // wave-uniform: the image ends inside this block
            const int jyc = cut ? min(jy, hq - 1 - ti0) : jy, jxc = cut ? min(jx, wq - 1 - tj0) : jx;
            const bool qvalid = jyc == jy && jxc == jx;
            const int r0 = T.row & 0xffff, c0 = T.col & 0xffff;            // this block's window origin
            const int ro = (T.row >> 16) & 15, co = (T.col >> 16) & 15;    // ... inside the staged halo: ro + 9 < HROWS, co <= 12

            // all LDS reads of the logit phase are issued before the first use (K fragments: slot j of halo row ro + t, channels
            // 8g .. 8g+7; the 40 bias values of this lane), then S^T: key tile t = halo row ro + t, slots co .. co + 15
            f32x4 sacc[HR];
            bf16x8 kf[HR];
            const int ks0 = ro * PC + co + j;
            const unsigned char* kb = Kimg + ks0 * KB + ((g ^ (((ks0 >> 2) & 1) << 1)) * 16);
            // The relative position bias enters as the MFMA's INITIAL ACCUMULATOR, in the units of the raw product (rpb / scale; the
            // logit is (q.k + b) * scale * log2 e).  A block none of whose 16 queries has its window clamped by the border
            // (wave-uniform) takes the table whose entries outside the centred 7 x 7 window are -1e30: bias and window mask are then
            // one function of the key's offset from the query (na2d_mfma.hip)
            const bool interior = (T.row & T.col & B_INT) != 0 && !cut;
            const int bts = ((interior ? S::BT_SIZE : 0) + S::bt_at((T.row >> 20) & 15, (int)((T.col >> 20) & 15) - S::BT_COL0)) * 4;   // uniform
            const int btl = cut ? S::bt_at(-jyc, -jxc + 4 * g) * 4 : btl_full;
            const float* bt = reinterpret_cast<const float*>(BT + bts + btl);
#pragma unroll
            for (int t = 0; t < HR; ++t) kf[t] = *reinterpret_cast<const bf16x8*>(kb + t * PC * KB);
#pragma unroll
            for (int t = 0; t < HR; ++t)
                sacc[t] = f32x4{bt[t * S::BT_DROW], bt[t * S::BT_DROW + S::BT_DCOL], bt[t * S::BT_DROW + 2 * S::BT_DCOL], bt[t * S::BT_DROW + 3 * S::BT_DCOL]};
#ifndef PPN_HALO_MASK_AFTER
            if (!interior) {
                // Window mask, applied to the INITIAL accumulator like the interior table's (-1e30 + q.k is -1e30 in float32): one
                // select per logit in front of the MFMA instead of two operations behind it.  Lane (j, g) holds, per tile t, keys
                // (row r0 + t, column c0 + 4g + r), r = 0 .. 3, of query j.  A query's window starts 0 .. TQY - 1 rows below r0 (the
                // clamp is monotone with slope <= 1), so halo rows TQY - 1 .. 6 of the block are in EVERY query's window: only
                // the TQY - 1 rows at either end need the row test.
                const int wi = clampm(ti0 + jyc - MN, 0, hs - MK), wj = clampm(tj0 + jxc - MN, 0, ws - MK);
                bool cv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int cx = c0 + 4 * g + r; cv[r] = cx >= wj && cx <= wj + MK - 1; }
#pragma unroll
                for (int t = 0; t < HR; ++t) {
                    const bool edge = t < TQY - 1 || t > MK - 1;          // compile-time
                    const bool rv = !edge || (r0 + t >= wi && r0 + t <= wi + MK - 1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[t][r] = (rv && cv[r]) ? sacc[t][r] : NEG;
                }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < HR; ++t) sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t], qf, sacc[t], 0, 0, 0);
#ifdef PPN_HALO_MASK_AFTER                                                  // (A/B build: the mask behind the MFMAs, two operations per logit)
            if (!interior) {
                const int wi = clampm(ti0 + jyc - MN, 0, hs - MK), wj = clampm(tj0 + jxc - MN, 0, ws - MK);
#pragma unroll
                for (int t = 0; t < HR; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cx = c0 + 4 * g + r;
                        const bool ok = r0 + t >= wi && r0 + t <= wi + MK - 1 && cx >= wj && cx <= wj + MK - 1;
                        sacc[t][r] = ok ? sacc[t][r] : NEG;
                    }
            }
#endif
            // The maxima are inline assembly, and the compiler's hazard recogniser does not look inside it: a vector instruction that
            // reads an MFMA's result fewer than 11 wait states after the MFMA's issue gets the register's OLD content (no hardware
            // interlock) — hipcc had scheduled the v_max3 of tile t right behind the MFMA of tile t + 1, and the "maximum" was one of
            // K-fragment bit patterns (harmless to the softmax while the logits are small, wrong all the same).  Every logit passes
            // through this statement, which is the wait.
            if constexpr (HR == 10)
                asm volatile("s_nop 7\n\ts_nop 4" : "+v"(sacc[0]), "+v"(sacc[1]), "+v"(sacc[2]), "+v"(sacc[3]), "+v"(sacc[4]), "+v"(sacc[5]), "+v"(sacc[6]),
                             "+v"(sacc[7]), "+v"(sacc[8]), "+v"(sacc[9]));
            else
                asm volatile("s_nop 7\n\ts_nop 4" : "+v"(sacc[0]), "+v"(sacc[1]), "+v"(sacc[2]), "+v"(sacc[3]), "+v"(sacc[4]), "+v"(sacc[5]), "+v"(sacc[6]),
                             "+v"(sacc[7]));
            static_assert(HR == 10 || HR == 8, "the wait above names every logit tile");
            float lg[4 * HR];
#pragma unroll
            for (int i = 0; i < 4 * HR; ++i) lg[i] = sacc[i >> 2][i & 3];
            float mx = max_tree<4 * HR>(lg);
            {   // the query's other three lane quarters (lane ^ 16, lane ^ 32) by register swaps: v_permlane16_swap / v_permlane32_swap
                // of the value with itself leave it and its partner's — no trip through the LDS crossbar (ds_bpermute) in the
                // chain every exponential waits for
                auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
                t = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
            }
            const float nm = -mx * sl2;                                    // p = 2^((S' - max) * scale * log2 e): one multiply-add per logit
#pragma unroll
            for (int t = 0; t < HR; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[t][r], sl2, nm));

            // O^T = V^T . P^T over 5 k-steps of 32 slots = key tiles (2ks, 2ks+1); a third product per k-step with an all-ones A
            // operand sums the (bfloat16-rounded, as the numerator uses them) probabilities into the softmax denominator
            f32x4 oacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, lacc = {0.f, 0.f, 0.f, 0.f};
            const bf16x8 ones = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};
            const int vslot = (ro * PC) + co + 4 * g + q4;
            const int vsw = (vslot >> 2) & 1;                              // this lane's slots are stored half-swapped (every row: 8 | PC)
            const unsigned char* vb = Vimg + vslot * KB + 8 * p4;
            const unsigned char* vbc[2] = {vb + vsw * 32, vb + (vsw ^ 1) * 32};
            bf16x4 vlo[HR / 2][2], vhi[HR / 2][2];
#pragma unroll
            for (int ks = 0; ks < HR / 2; ++ks)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    vlo[ks][cb] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vbc[cb] + (2 * ks) * PC * KB));
                    vhi[ks][cb] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(vbc[cb] + (2 * ks + 1) * PC * KB));
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < HR / 2; ++ks) {
                const f32x4 pa = sacc[2 * ks], pb = sacc[2 * ks + 1];
                const bf16x8 pf = {(__bf16)pa[0], (__bf16)pa[1], (__bf16)pa[2], (__bf16)pa[3], (__bf16)pb[0], (__bf16)pb[1], (__bf16)pb[2], (__bf16)pb[3]};
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const bf16x4 lo = vlo[ks][cb], hi = vhi[ks][cb];
                    const bf16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    oacc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vv, pf, oacc[cb], 0, 0, 0);
                }
                lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf, lacc, 0, 0, 0);
            }
            if (qvalid) {
                const float inv = __builtin_amdgcn_rcpf(lacc[0]);     // (1 ulp; the full division is 10 instructions for a value rounded to bfloat16 next)
                char* dst = reinterpret_cast<char*>(out) + T.o + (unsigned)T.h * (MHD * 2) + ooff_full;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const uint2 w = make_uint2(pack_bf16x2(oacc[cb][0] * inv, oacc[cb][1] * inv), pack_bf16x2(oacc[cb][2] * inv, oacc[cb][3] * inv));
                    *reinterpret_cast<uint2*>(dst + cb * 32) = w;           // channels cb*16 + 4g .. +3 of query j
                }
            }
        }
        if (!more) break;
        cur = nxt;
        item = nitem;
        bsel ^= 1;
    }
}

namespace {
// The descriptor tables, one per launch geometry, kept for the life of the process (a PPNet / DiNAT forward has a handful of
// geometries; 64 bytes per tile).  A table is built on first use with hipMalloc + the preparation kernel + a stream synchronise —
// never inside a stream capture (hipMalloc is illegal there): a geometry first met while capturing is declined (-1: the caller's
// per-tile kernel runs), the warm-up pass in front of a capture has normally met it already.  Nothing is allocated or freed per
// launch, so a captured graph holds no allocation nodes of this kernel.
struct DescKey {
    int dev, B, H, W, Hr, Wr, heads, dil, padded, tqy;
    bool operator<(const DescKey& o) const {
        const int a[10] = {dev, B, H, W, Hr, Wr, heads, dil, padded, tqy}, b[10] = {o.dev, o.B, o.H, o.W, o.Hr, o.Wr, o.heads, o.dil, o.padded, o.tqy};
        for (int i = 0; i < 10; ++i)
            if (a[i] != b[i]) return a[i] < b[i];
        return false;
    }
};
std::mutex g_desc_mutex;
std::map<DescKey, int*> g_desc;

const int* descriptors(const DescKey& k, int tiles_y, int tiles_x, long long total, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_desc_mutex);
    const auto it = g_desc.find(k);
    if (it != g_desc.end()) return it->second;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    if (g_desc.size() >= 512) return nullptr;                               // (a process that meets this many geometries keeps the per-tile kernel for the rest)
    int* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)total * DESC * 4) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(na2d_halo16_prep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, d, k.heads, k.H, k.W, k.Hr, k.Wr, k.dil, tiles_y,
                       tiles_x, (int)total, k.padded, k.tqy, 16 / k.tqy);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { (void)hipFree(d); return nullptr; }
    g_desc[k] = d;
    return d;
}
}  // namespace

// 0 = launched; -1 = shape outside this kernel (the caller falls back to na2d_mfma_kernel<16>)
int na2d_halo16_launch(const void* qkv, const void* pad_kv, const float* rpb, void* out, int B, int H, int W, int Hr, int Wr, int heads, int dil,
                       float scale, const void* zero, hipStream_t stream) {
    const int hq = (Hr + dil - 1) / dil, wq = (Wr + dil - 1) / dil;          // largest query sub-image
    const int tiles_y = (hq + RT - 1) / RT, tiles_x = (wq + RT - 1) / RT;
    const long long total = (long long)tiles_y * tiles_x * B * dil * dil;
    const long long items = total * heads;
    const int Ws = pad_kv ? Wr : W;                                          // stored token grid
    // a piece's offset from its tile's halo origin, and a query's from its tile's, are 32-bit (bytes)
    if (items >= (1LL << 30) || heads > 255 || (long long)(HROWS + 2) * dil * Ws * 3 * heads * MHD * 2 >= (1LL << 31)) return -1;
    // the query block of a wave: 2 x 8 (8 key tiles) unless PPNET_NA_HALO_BLOCK=4x4 (10 key tiles; the round-3/4 form, kept for A/B)
    const char* const blk_env = std::getenv("PPNET_NA_HALO_BLOCK");         // read per launch: tools alternate it inside one process
    const bool blk44 = blk_env && blk_env[0] == '4';
    static DeviceOnce attr44, attr28;                                        // per device, like the descriptor tables (DescKey.dev)
    const int dev = current_device();
    if (dev < 0) return (int)hipErrorInvalidDevice;
    const int n_cu = device_cu_count() & ~7;
    const void* fn = blk44 ? (const void*)na2d_halo16_kernel<4, 4> : (const void*)na2d_halo16_kernel<2, 8>;
    const int lds = blk44 ? Blk<4, 4>::LDS_TOTAL : Blk<2, 8>::LDS_TOTAL;
    if (n_cu < 8 || dynamic_lds_once(blk44 ? attr44 : attr28, fn, lds) != 0) return -1;
    const int* desc = descriptors(DescKey{dev, B, H, W, Hr, Wr, heads, dil, pad_kv ? 1 : 0, blk44 ? 4 : 2}, tiles_y, tiles_x, total, stream);
    if (!desc) return -1;
    // whole XCD rows of workgroups, no more than the items of an XCD's run
    const long long per = (items + 7) / 8;
    int grid = n_cu;
    if (per * 8 < grid) grid = (int)per * 8;
    if (blk44)
        hipLaunchKernelGGL((na2d_halo16_kernel<4, 4>), dim3(grid), dim3(NTHR), lds, stream, (const __bf16*)qkv, (const __bf16*)pad_kv, (__bf16*)out, rpb, desc,
                           Wr, Ws, heads, dil, scale, (int)items);
    else
        hipLaunchKernelGGL((na2d_halo16_kernel<2, 8>), dim3(grid), dim3(NTHR), lds, stream, (const __bf16*)qkv, (const __bf16*)pad_kv, (__bf16*)out, rpb, desc,
                           Wr, Ws, heads, dil, scale, (int)items);
    return (int)hipGetLastError();
}

}  // namespace ppn
