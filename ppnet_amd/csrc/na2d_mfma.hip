// na2d_mfma.hip — fused neighbourhood attention forward on the matrix cores (bfloat16), the op behind
// natten.NeighborhoodAttention2D as the reference uses it (SegNet/nat.py:14,111-120,144); same semantics as na2d.hip and
// oracle/na_np.py (kernel 7, head dim 32, dilation groups, window start clamp(u - 3, 0, n - 7), relative position bias).
//
// Why: the VALU kernel (na2d.hip) spends 1 680 v_dot2_f32_bf16 per query and that instruction issues at a quarter of the
// plain VALU rate on gfx950 (measured: 8-10 queries/ns = 10 cycles per dot2), so it sits at its own instruction roofline at
// 0.25 of the HBM one.  Here one WAVE owns a 4 x 4 block of queries of one (image, head, dilation group):
//   S^T = K_halo . Q^T     10 x v_mfma_f32_16x16x32_bf16: key tile t = halo row t (16 slots = columns c0 .. c0+15, 10 used),
//                          columns = the 16 queries; a lane holds 4 keys x 1 query per tile (40 logits)
//   mask + bias            validity is separable (row t in the query's window) & (column in its window): a 10-bit row mask and
//                          four column masks per lane, one v_bfi per logit; the bias is one LDS read with an immediate offset
//                          from a zero-padded 16 x 22 copy of rpb[h] (index = lane base + 22 t + r)
//   softmax                over a query's 160 slots = 40 registers x the 4 lanes {l, l^16, l^32, l^48}
//   O^T = V^T . P^T        10 MFMAs; P^T is the S^T accumulator converted in place (k slot 8g+j <-> key 4g+(j&3) of tile
//                          2ks + (j>>2): no lane movement), V^T comes from the row-major V halo by ds_read_b64_tr_b16
// i.e. ~450 VALU instructions per 16 queries instead of 1 680 quarter-rate ones per query.  Padded positions of DiNAT's
// dilated layers ("virtual padding": k / v = the qkv bias) are handled in the staging loads; no special compute path.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "ppn_device.h"
#include "ppn_kernels.h"

namespace ppn {

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int MK = 7, MN = 3, MHD = 32;
constexpr int TQ = 4;                  // 4 x 4 queries per MFMA block
constexpr int HR = 10;                 // key tiles (halo rows) per block
constexpr int KB = 64;                 // bytes per staged key row (32 bf16)
constexpr int BT_ROWS = 16, BT_COLS = 22;

__device__ __forceinline__ int clampm(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
}  // namespace

// RT x RT queries per halo tile.  RT = 16: one tile per workgroup, its 16 blocks over the 4 waves (halo 22 x 22 keys, 1.9 x the
// queries);  RT = 8: one tile, one block per wave;  RT = 4: four independent tiles per workgroup, one per wave (DiNAT's small
// dilation groups).  The halo is staged once per tile for K and V (rows of 64 B, HROWS x PC slots; the PC - HCOLS slack columns
// and everything outside the image are zero) and every block reads its 10 x 16 window of it.
template <int RT>
__global__ __launch_bounds__(RT == 16 ? 512 : 256, RT == 16 ? 4 : 2) void na2d_mfma_kernel(const __bf16* __restrict__ qkv, const __bf16* __restrict__ pad_kv,
                                                           const float* __restrict__ rpb, __bf16* __restrict__ out, int B, int H, int W, int Hr,
                                                           int Wr, int heads, int dil, float scale, int tiles_y, int tiles_x, int total_tiles,
                                                           const __bf16* __restrict__ zero) {
    constexpr int NTHR = RT == 16 ? 512 : 256, NWAVES = NTHR / 64;
    constexpr int TPW = RT == 4 ? 4 : 1;                                   // tiles per workgroup
    constexpr int NT = NTHR / TPW;                                         // threads staging one tile
    constexpr int HROWS = RT + 6, PC = RT + 12;                            // halo rows, column pitch in slots (RT + 6 columns are loaded)
    constexpr int IMG = HROWS * PC * KB;                                   // bytes of one staged image (K or V)
    constexpr int PIECES = HROWS * PC * 4;
    constexpr int ITER = (PIECES + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char nl[];
    float* BT = reinterpret_cast<float*>(nl);                              // [16][22]: rpb[h] / scale, zero outside the 13 x 13 table
    // [16][22]: the same inside the centred 7 x 7 window, -1e30 elsewhere — bias AND window mask of a query whose window is not
    // clamped by the image border.  Both in the units of the raw product: the logit is (q.k + table) * scale * log2 e
    float* BTM = BT + BT_ROWS * BT_COLS;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sub = TPW == 4 ? wave : 0;                                   // which tile of the workgroup
    const int tid = TPW == 4 ? lane : (int)threadIdx.x;                    // thread within the tile's staging group
    unsigned char* Kimg = nl + 2 * BT_ROWS * BT_COLS * 4 + sub * 2 * IMG;
    unsigned char* Vimg = Kimg + IMG;
    // XCD-aware order: workgroups w and w + 8 share an L2; give each XCD a contiguous run of (tile, head) pairs with the head
    // fastest — the heads of a token share its 128-byte lines (two heads per line of q, k and v), neighbouring tiles share halos
    int wg = blockIdx.x;
    {
        const int n = gridDim.x, q = n >> 3, r = n & 7, x = wg & 7;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (wg >> 3);
    }
    const int h = wg % heads;
    wg /= heads;
    for (int t = threadIdx.x; t < BT_ROWS * BT_COLS; t += NTHR) {
        const int a = t / BT_COLS, b = t - a * BT_COLS;
        BT[t] = (a < 13 && b < 13) ? rpb[(size_t)h * 169 + a * 13 + b] / scale : 0.f;
        BTM[t] = (a >= MN && a <= 3 * MN && b >= MN && b <= 3 * MN) ? rpb[(size_t)h * 169 + a * 13 + b] / scale : -1.0e30f;
    }
    const int gtile = min(wg * TPW + sub, total_tiles - 1);                // surplus waves redo the last tile and store nothing
    const bool live = wg * TPW + sub < total_tiles;
    const int ntiles = tiles_y * tiles_x;
    const int bz = gtile / ntiles;
    const int tile_id = gtile - bz * ntiles;
    const int b = bz / (dil * dil), g2 = bz % (dil * dil);
    const int gi = g2 / dil, gj = g2 % dil;
    const int hs = (H - gi + dil - 1) / dil, ws = (W - gj + dil - 1) / dil;                     // key sub-image of this dilation group
    const int hq = gi < Hr ? (Hr - gi + dil - 1) / dil : 0, wq = gj < Wr ? (Wr - gj + dil - 1) / dil : 0;   // its real part: the queries
    const int ty0 = (tile_id / tiles_x) * RT, tx0 = (tile_id % tiles_x) * RT;
    const bool tile_in = live && ty0 < hq && tx0 < wq;                     // groups differ by one row / column
    const int Hs = pad_kv ? Hr : H, Ws = pad_kv ? Wr : W;                  // stored token grid
    const size_t tok = (size_t)3 * heads * MHD;
    const int R0 = clampm(ty0 - MN, 0, hs - MK), C0 = clampm(tx0 - MN, 0, ws - MK);            // halo origin
    const int ymax = min(ty0 + RT - 1, hq - 1), xmax = min(tx0 + RT - 1, wq - 1);
    const int NR = clampm(ymax - MN, 0, hs - MK) + MK - R0, NC = clampm(xmax - MN, 0, ws - MK) + MK - C0;   // halo extent

    // ---- the q fragments of this wave's blocks are requested first: their latency hides behind the staging (B operand: channels
    // 8g .. 8g+7 of query j; dead queries shadow a live one and are never stored)
    constexpr int NBLK = (RT / TQ) * (RT / TQ), BPW = RT == 4 ? 1 : NBLK / NWAVES;   // blocks per tile, blocks per wave
    const int j = lane & 15, g = lane >> 4;                                // MFMA column (query) and lane quarter
    bf16x8 qfs[BPW];
#pragma unroll
    for (int bw = 0; bw < BPW; ++bw) {
        const int blk = RT == 4 ? 0 : wave * BPW + bw;
        const int ti0 = ty0 + (blk / (RT / TQ)) * TQ, tj0 = tx0 + (blk % (RT / TQ)) * TQ;
        const int uq = clampm(ti0 + (j >> 2), 0, max(hq - 1, 0)), vq_ = clampm(tj0 + (j & 3), 0, max(wq - 1, 0));
        const int qy = gi + uq * dil, qx = gj + vq_ * dil;
        qfs[bw] = *reinterpret_cast<const bf16x8*>(qkv + ((size_t)(b * Hs + min(qy, Hs - 1)) * Ws + min(qx, Ws - 1)) * tok + (size_t)h * MHD + 8 * g);
    }

    // ---- stage K and V: every piece has an address — its token's row, the padded token (virtual padding: k / v = the qkv
    // bias) or a line of zeros (slack columns, rows beyond the halo; V must stay finite) — so the loads are unconditional
    if (tile_in) {
        const int chunk = tid & 3;
        uint4 kq[ITER], vq[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {                                 // all loads of the tile in flight before the first LDS write
            const int p = min(it * NT + tid, PIECES - 1);
            const int slot = p >> 2;
            const int t = slot / PC, sc = slot - t * PC;
            const int y = gi + (R0 + t) * dil, x = gj + (C0 + sc) * dil;
            const bool inside = t < NR && sc < NC;
            const bool real = inside && (!pad_kv || (y < Hr && x < Wr));
            const __bf16* pr = qkv + ((size_t)(b * Hs + (real ? y : 0)) * Ws + (real ? x : 0)) * tok + ((size_t)heads + h) * MHD;
            const __bf16* pp = pad_kv ? pad_kv + ((size_t)heads + h) * MHD : zero;
            const __bf16* src = (real ? pr : (inside ? pp : zero)) + 8 * chunk;
            const bool z = !inside || (!real && !pad_kv);
            kq[it] = *reinterpret_cast<const uint4*>(src);
            vq[it] = *reinterpret_cast<const uint4*>(src + (z ? 0 : heads * MHD));
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int p = it * NT + tid;
            if (p < PIECES) {
                *reinterpret_cast<uint4*>(Kimg + (p >> 2) * KB + chunk * 16) = kq[it];
                // V rows are stored with their 32-byte halves swapped on every other group of 4 slots: the transposed reads below
                // take 8 bytes per lane from 16 slots at a 64-byte pitch, and slots s and s + 4 would meet in the same banks
                *reinterpret_cast<uint4*>(Vimg + (p >> 2) * KB + ((chunk ^ (((p >> 4) & 1) << 1)) * 16)) = vq[it];
            }
        }
    }
    __syncthreads();
    if (!tile_in) return;

    const int q4 = j >> 2, p4 = j & 3;                                     // transposed read: this lane addresses row q4, columns 4 p4 ..
    const float NEG = -1.0e30f;
    const float sl2 = scale * 1.4426950408889634f;
#pragma unroll
    for (int bw = 0; bw < BPW; ++bw) {
        const int blk = RT == 4 ? 0 : wave * BPW + bw;
        const int ti0 = ty0 + (blk / (RT / TQ)) * TQ, tj0 = tx0 + (blk % (RT / TQ)) * TQ;
        if (ti0 >= hq || tj0 >= wq) continue;                              // wave-uniform
        const int u = ti0 + (j >> 2), v = tj0 + (j & 3);
        const bool qvalid = u < hq && v < wq;
        const int uc = u < hq ? u : hq - 1, vc = v < wq ? v : wq - 1;      // dead queries shadow a live one (never stored)
        const int wi = clampm(uc - MN, 0, hs - MK), wj = clampm(vc - MN, 0, ws - MK);
        const int r0 = clampm(ti0 - MN, 0, hs - MK), c0 = clampm(tj0 - MN, 0, ws - MK);        // this block's window origin
        const int ro = r0 - R0, co = c0 - C0;                              // ... inside the staged halo: ro + 9 < HROWS, co + 15 < PC

        const bf16x8 qf = qfs[bw];

        // all LDS reads of the logit phase are issued before the first use (K fragments: slot j of halo row ro + t, channels
        // 8g .. 8g+7; the 40 bias values of this lane), then S^T: key tile t = halo row ro + t, slots co .. co + 15
        f32x4 sacc[HR];
        bf16x8 kf[HR];
        const unsigned char* kb = Kimg + ((ro * PC) + co + j) * KB + g * 16;
        // The relative position bias enters as the MFMA's INITIAL ACCUMULATOR, in the units of the raw product (rpb / scale; the logit
        // is (q.k + b) * scale * log2 e): S' = K Q^T + b comes out of the matrix pipe, no multiply-add per logit.  A block none of whose
        // 16 queries has its window clamped by the border (wave-uniform; 3/4 of the blocks of a 64 x 64 map) takes the table whose
        // entries outside the centred 7 x 7 window are -1e30: bias and window mask are then one function of the key's offset from the
        // query and the row / column window tests and selects — a quarter of this kernel's VALU work — do not exist for it.
        const bool interior = ti0 >= MN && ti0 + TQ - 1 + MN <= hs - 1 && tj0 >= MN && tj0 + TQ - 1 + MN <= ws - 1 &&
                              ti0 + TQ <= hq && tj0 + TQ <= wq;
        const float* bt = (interior ? BTM : BT) + (r0 - uc + MK - 1) * BT_COLS + (c0 + 4 * g - vc + MK - 1);
#pragma unroll
        for (int t = 0; t < HR; ++t) kf[t] = *reinterpret_cast<const bf16x8*>(kb + t * PC * KB);
#pragma unroll
        for (int t = 0; t < HR; ++t) sacc[t] = f32x4{bt[t * BT_COLS], bt[t * BT_COLS + 1], bt[t * BT_COLS + 2], bt[t * BT_COLS + 3]};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < HR; ++t) sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t], qf, sacc[t], 0, 0, 0);
        if (!interior) {
            // window mask.  Lane (j, g) holds, per tile t, keys (row r0 + t, column c0 + 4g + r), r = 0 .. 3, of query j
            uint32_t rowmask = 0;
#pragma unroll
            for (int t = 0; t < HR; ++t) rowmask |= (uint32_t)((r0 + t >= wi) && (r0 + t <= wi + MK - 1)) << t;
            uint32_t csel[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int cx = c0 + 4 * g + r; csel[r] = (cx >= wj && cx <= wj + MK - 1) ? 0xffffffffu : 0u; }
#pragma unroll
            for (int t = 0; t < HR; ++t) {
                const uint32_t rsel = 0u - ((rowmask >> t) & 1u);          // all ones when halo row t is in this query's window
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t m = rsel & csel[r];
                    sacc[t][r] = __uint_as_float((__float_as_uint(sacc[t][r]) & m) | (__float_as_uint(NEG) & ~m));
                }
            }
        }
        float mx = NEG;
#pragma unroll
        for (int t = 0; t < HR; ++t) mx = fmaxf(mx, fmaxf(fmaxf(sacc[t][0], sacc[t][1]), fmaxf(sacc[t][2], sacc[t][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float nm = -mx * sl2;                                        // p = 2^((S' - max) * scale * log2 e): one multiply-add per logit
#pragma unroll
        for (int t = 0; t < HR; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[t][r], sl2, nm));

        // O^T = V^T . P^T over 5 k-steps of 32 slots = key tiles (2ks, 2ks+1)
        // The softmax denominator comes out of the matrix pipe too: a third product per k-step with an all-ones A operand sums the
        // (bfloat16-rounded, as the numerator uses them) probabilities of each query over the step's 32 key slots into every row of
        // its column — 5 MFMAs on an idle pipe instead of 40 adds and two cross-lane exchanges on the VALU, which is this kernel's limit.
        f32x4 oacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, lacc = {0.f, 0.f, 0.f, 0.f};
        const bf16x8 ones = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};
        const int vslot = (ro * PC) + co + 4 * g + q4;
        const int vsw = (vslot >> 2) & 1;                                  // this lane's slots of the even halo rows are stored half-swapped
        constexpr int VFLIP = (PC >> 2) & 1;                               // ... and those of the odd rows the other way round (PC / 4 odd)
        const unsigned char* vb = Vimg + vslot * KB + 8 * p4;
        const unsigned char* vbc[2] = {vb + vsw * 32, vb + (vsw ^ 1) * 32};
        s16x4 vlo[HR / 2][2], vhi[HR / 2][2];
#pragma unroll
        for (int ks = 0; ks < HR / 2; ++ks)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                vlo[ks][cb] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vbc[cb] + (2 * ks) * PC * KB));
                vhi[ks][cb] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vbc[cb ^ VFLIP] + (2 * ks + 1) * PC * KB));
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < HR / 2; ++ks) {
            const f32x4 pa = sacc[2 * ks], pb = sacc[2 * ks + 1];
            const bf16x8 pf = {(__bf16)pa[0], (__bf16)pa[1], (__bf16)pa[2], (__bf16)pa[3], (__bf16)pb[0], (__bf16)pb[1], (__bf16)pb[2], (__bf16)pb[3]};
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const s16x4 lo = vlo[ks][cb], hi = vhi[ks][cb];
                const s16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                oacc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vv), pf, oacc[cb], 0, 0, 0);
            }
            lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf, lacc, 0, 0, 0);
        }
        if (qvalid) {
            const float inv = __builtin_amdgcn_rcpf(lacc[0]);         // (1 ulp; as na2d_halo16.hip: the two kernels stay bit-equal)
            const int y = gi + u * dil, x = gj + v * dil;
            __bf16* dst = out + ((size_t)(b * Hr + y) * Wr + x) * ((size_t)heads * MHD) + (size_t)h * MHD;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const uint2 w = make_uint2(pack_bf16x2(oacc[cb][0] * inv, oacc[cb][1] * inv), pack_bf16x2(oacc[cb][2] * inv, oacc[cb][3] * inv));
                *reinterpret_cast<uint2*>(dst + cb * 16 + 4 * g) = w;       // channels cb*16 + 4g .. +3 of query j
            }
        }
    }
}

template <int RT>
static int launch_rt(const void* qkv, const void* pad_kv, const float* rpb, void* out, int B, int H, int W, int Hr, int Wr, int heads, int dil,
                     float scale, const __bf16* zero, hipStream_t stream) {
    constexpr int TPW = RT == 4 ? 4 : 1;
    const int hq = (Hr + dil - 1) / dil, wq = (Wr + dil - 1) / dil;          // largest query sub-image
    const int tiles_y = (hq + RT - 1) / RT, tiles_x = (wq + RT - 1) / RT;
    const long long total = (long long)tiles_y * tiles_x * B * dil * dil;
    if (total >= (1LL << 31) - 8) return -1;
    const size_t lds = (size_t)2 * BT_ROWS * BT_COLS * 4 + (size_t)TPW * 2 * (RT + 6) * (RT + 12) * KB;
    static DeviceOnce attr;
    if (const int e = dynamic_lds_once(attr, (const void*)na2d_mfma_kernel<RT>, (int)lds)) return e;
    const long long wgs = (total + TPW - 1) / TPW * heads;
    if (wgs >= (1LL << 31)) return -1;
    const dim3 grid((unsigned)wgs, 1, 1);
    hipLaunchKernelGGL(na2d_mfma_kernel<RT>, grid, dim3(RT == 16 ? 512 : 256), lds, stream, (const __bf16*)qkv, (const __bf16*)pad_kv, rpb, (__bf16*)out, B, H, W, Hr,
                       Wr, heads, dil, scale, tiles_y, tiles_x, (int)total, zero);
    return (int)hipGetLastError();
}

int na2d_mfma_launch(const void* qkv, const void* pad_kv, const float* rpb, void* out, int B, int H, int W, int Hr, int Wr, int heads, int dil,
                     float scale, hipStream_t stream) {
    const __bf16* zero = (const __bf16*)zero_line();
    if (!zero) return (int)hipErrorOutOfMemory;
    const int hq = (Hr + dil - 1) / dil, wq = (Wr + dil - 1) / dil;
    int best = na_region_size(hq, wq);                                      // the cheapest cover of the query sub-image (ppn_kernels.h)
    if (const char* f = getenv("PPNET_NA_RT")) { const int v = atoi(f); if (v == 4 || v == 8 || v == 16) best = v; }   // A/B: force the region size
    if (best == 4) return launch_rt<4>(qkv, pad_kv, rpb, out, B, H, W, Hr, Wr, heads, dil, scale, zero, stream);
    if (best == 8) return launch_rt<8>(qkv, pad_kv, rpb, out, B, H, W, Hr, Wr, heads, dil, scale, zero, stream);
    // 16 x 16 regions: the persistent LDS-DMA kernel (na2d_halo16.hip); this file's kernel where that one declines (or PPNET_NA_HALO16=0: A/B)
    const char* ab = getenv("PPNET_NA_HALO16");
    if (!(ab && ab[0] == '0')) {
        const int r = na2d_halo16_launch(qkv, pad_kv, rpb, out, B, H, W, Hr, Wr, heads, dil, scale, zero, stream);
        if (r != -1) return r;
    }
    return launch_rt<16>(qkv, pad_kv, rpb, out, B, H, W, Hr, Wr, heads, dil, scale, zero, stream);
}

}  // namespace ppn
