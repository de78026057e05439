// na2d.hip — fused 2-D neighbourhood attention forward for gfx950:  QK^T + relative position bias ->
// softmax over the k x k (dilated) neighbourhood -> AV, in one kernel, reading q/k/v straight out of the
// qkv projection's output layout and writing the layout the output projection consumes.
//
// Replaces the two NATTEN CUDA kernels the reference calls through natten.NeighborhoodAttention2D
// (SegNet/nat.py:111-120,144): natten2dqkrpb + softmax + natten2dav.  NATTEN's source is not part of the
// reference; the semantics implemented here are stated in oracle/na_np.py ("parity unpinned"):
// a query at position i of an axis of length L belongs to dilation group i mod d; inside the group (positions
// g, g+d, ...; n of them) its window is the k consecutive group members starting at clamp(i/d - k/2, 0, n-k),
// and the bias index of window member t is (start + t - i/d) + (k-1).
//
// Dilated attention is plain (d = 1) attention on each of the d*d sub-sampled images, so one workgroup
// handles a 16x16 tile of queries of one (batch, head, dilation group): the (16+k-1)^2 halo of K rows is
// staged in LDS (rows padded to a conflict-free stride for ds_read_b128), each lane keeps its 49 logits in
// registers, then the same LDS buffer is refilled with V.  HBM traffic is q, k, v read once (+ halo overlap)
// and out written once; head_dim = 32 (every NAT/DiNAT level), kernel 7.
//
// The kernel is VALU-bound (SQ_ACTIVE_INST_VALU ~ 80 %), so the bf16 path keeps both products on v_dot2c_f32_bf16:
// QK over channel pairs, AV over NEIGHBOUR pairs (V staged as row pairs, probabilities rounded to bf16 pairs, float32
// accumulation): 1680 dot2 per query instead of 784 dot2 + 1.6 k unpack + 1.6 k FMA — 0.85 -> 0.69 ms on the 64x64
// level at batch 256.  The float32 path keeps float32 probabilities and FMAs.
#include <hip/hip_bf16.h>
#include <cstdlib>
#include "ppn_device.h"
#include "ppn_kernels.h"

namespace ppn {

namespace {
constexpr int KS = 7, NS = 3, HD = 32;

template <typename T> struct Row;
template <> struct Row<float> { static constexpr int STRIDE = 36; };          // 144 B: 16 lanes x b128 hit 64 distinct banks
template <> struct Row<__hip_bfloat16> { static constexpr int STRIDE = 40; }; // 80 B

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__hip_bfloat16 v) { return __bfloat162float(v); }
__device__ __forceinline__ void store_out(float* p, float v) { *p = v; }
__device__ __forceinline__ void store_out(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// one 32-element row (LDS or global) -> 32 floats, with 16-byte accesses
__device__ __forceinline__ void load_row(const float* row, float (&r)[HD]) {
#pragma unroll
    for (int p = 0; p < HD / 4; ++p) {
        const float4 v = *reinterpret_cast<const float4*>(row + 4 * p);
        r[4 * p] = v.x; r[4 * p + 1] = v.y; r[4 * p + 2] = v.z; r[4 * p + 3] = v.w;
    }
}
__device__ __forceinline__ void load_row(const __hip_bfloat16* row, float (&r)[HD]) {
#pragma unroll
    for (int p = 0; p < HD / 8; ++p) {
        const uint4 v = *reinterpret_cast<const uint4*>(row + 8 * p);       // 8 bf16; f32 bits = bf16 bits << 16
        r[8 * p] = __uint_as_float(v.x << 16);     r[8 * p + 1] = __uint_as_float(v.x & 0xffff0000u);
        r[8 * p + 2] = __uint_as_float(v.y << 16); r[8 * p + 3] = __uint_as_float(v.y & 0xffff0000u);
        r[8 * p + 4] = __uint_as_float(v.z << 16); r[8 * p + 5] = __uint_as_float(v.z & 0xffff0000u);
        r[8 * p + 6] = __uint_as_float(v.w << 16); r[8 * p + 7] = __uint_as_float(v.w & 0xffff0000u);
    }
}
}  // namespace

// TILE x TILE queries per tile, NT threads per tile (= TILE*TILE), THREADS/NT tiles per workgroup.  TILE = 16: one
// tile per 256-thread workgroup (large sub-images); TILE = 8: four independent 8x8 tiles, one per wave; TILE = 4:
// eight 4x4 tiles per 128-thread workgroup.  DiNAT's dilations leave 7x7 or 8x8 sub-images on most layers (a 16x16
// tile would idle 75-80 % of its lanes), and on the PADDED layers only the Hr x Wr real tokens are queries: their
// 4x4 real corner of each 7x7 group fills a 4x4 tile exactly.
//
// qkv is laid out for the padded H x W token grid (the module pads before the qkv projection, like NATTEN);
// out is the unpadded [B][Hr][Wr][heads*32] tensor — padded positions are keys/values only, never queries.
// Register budget: the plain bf16 variants fit 128 VGPRs without spills (4 waves per SIMD: +10 % on the 8x8-tile layers); the
// pad-aware 8x8 variant is held to 168 (3 waves: 0.24 -> 0.19 ms on the d = 3 layers, 2 spills); the float32 and the other
// pad-aware variants need 145-171 and are left uncapped (capping them measured slower).
template <typename T, int TILE, int THREADS, bool VPAD>
__global__ __launch_bounds__(THREADS, (sizeof(T) == 4 ? 1 : (VPAD ? (TILE == 8 ? 3 : 1) : 4))) void na2d_fwd_kernel(const T* __restrict__ qkv, const T* __restrict__ pad_kv, const float* __restrict__ rpb,
                                                           T* __restrict__ out, int B, int H, int W, int Hr, int Wr, int heads,
                                                           int dil, float scale, int total_tiles, int halo_r, int halo_c, int tile_bytes, int kpitch, int vpitch) {
    constexpr int STRIDE = Row<T>::STRIDE;
    constexpr int NT = TILE * TILE, TPB = THREADS / NT;                      // tiles per workgroup
    // halo_r x halo_c = min(TILE + KS - 1, sub-image extent): the LDS footprint (and so the occupancy) follows the
    // sub-image — DiNAT's padded dilated layers have 7x7 groups, 49 rows per tile instead of 196
    // kpitch: elements between staged rows of K (or V rows, f32 path); vpitch: dwords between staged row pairs of V
    extern __shared__ unsigned char na_lds[];
    const int sub = threadIdx.x / NT;                                        // which tile of this workgroup
    T* tile = reinterpret_cast<T*>(na_lds + (size_t)sub * tile_bytes);
    float* bias = reinterpret_cast<float*>(na_lds + (size_t)TPB * tile_bytes);   // [13][13]

    const int tid = threadIdx.x % NT, ty = tid / TILE, tx = tid % TILE;
    const int h = blockIdx.y;
    const int tiles_x = (((Wr + dil - 1) / dil) + TILE - 1) / TILE, tiles_y = (((Hr + dil - 1) / dil) + TILE - 1) / TILE;
    const int ntiles = tiles_x * tiles_y;
    const int gtile = min((int)blockIdx.x * TPB + sub, total_tiles - 1);      // surplus sub-tiles redo the last tile's loads, store nothing
    const bool live = (int)blockIdx.x * TPB + sub < total_tiles;
    const int bz = gtile / ntiles, tile_id = gtile - bz * ntiles;             // tiles * B * d*d folded into grid.x
    const int b = bz / (dil * dil), g = bz % (dil * dil);
    const int gi = g / dil, gj = g % dil;
    const int hs = (H - gi + dil - 1) / dil, ws = (W - gj + dil - 1) / dil;   // sub-image of this dilation group (keys)
    const int hq = gi < Hr ? (Hr - gi + dil - 1) / dil : 0, wq = gj < Wr ? (Wr - gj + dil - 1) / dil : 0;   // ... its real part (queries)
    const int ti0 = (tile_id / tiles_x) * TILE, tj0 = (tile_id % tiles_x) * TILE;
    // a tile outside this group's queries (groups differ by one row/column) idles through the barriers
    const bool tile_in = live && ti0 < hq && tj0 < wq;

    const int u = ti0 + ty, v = tj0 + tx;                                   // query in sub-image coordinates
    const bool valid = tile_in && u < hq && v < wq;
    const int umax = min(ti0 + TILE - 1, hq - 1), vmax = min(tj0 + TILE - 1, wq - 1);
    const int r0 = clampi(ti0 - NS, 0, hs - KS), c0 = clampi(tj0 - NS, 0, ws - KS);
    const int nr = clampi(umax - NS, 0, hs - KS) + KS - r0, nc = clampi(vmax - NS, 0, ws - KS) + KS - c0;
    const int wi = clampi(u - NS, 0, hs - KS), wj = clampi(v - NS, 0, ws - KS);   // window start of this query
    const size_t tok = (size_t)3 * heads * HD;                              // elements per token in qkv
    // pad_kv == null: qkv is stored for the whole H x W grid.  pad_kv != null ("virtual padding"): qkv holds only the
    // Hr x Wr real tokens and every padded position has the same k / v, pad_kv[3][heads][32] (the qkv bias: NATTEN's module
    // zero-pads BEFORE its projection) — the projection and this kernel then never touch the 1.7-3x larger padded grid.
    const int Hs = pad_kv ? Hr : H, Ws = pad_kv ? Wr : W;
    // VPAD (bf16 launches with pad_kv): the real keys of a dilation group are the top-left hq x wq corner of its sub-image
    // and every other key is the same vector, so only real rows are staged, a padded neighbour's logit is the one dot
    // product q . k_pad (+ its own position bias), and the padded neighbours' probabilities are summed into one
    // p_pad * v_pad term — DiNAT's 7x7 groups with a 4x4 real corner do a third of the staging, QK and AV work.
    constexpr bool SKIP = VPAD && sizeof(T) == 2;
    const int rreal = SKIP ? clampi(hq - r0, 0, nr) : nr, creal = SKIP ? clampi(wq - c0, 0, nc) : nc;   // real part of the halo
    auto kv_src = [&](int which, int y, int x) -> const T* {
        return (!pad_kv || (y < Hr && x < Wr)) ? qkv + ((size_t)(b * Hs + y) * Ws + x) * tok + ((size_t)which * heads + h) * HD
                                                : pad_kv + ((size_t)which * heads + h) * HD;
    };

    for (int t = threadIdx.x; t < 13 * 13; t += THREADS) bias[t] = rpb[(size_t)h * 169 + t];

    auto load_tile = [&](int which) {                                       // which: 1 = K, 2 = V
        // one 16-byte piece per lane: HD*sizeof(T)/16 pieces per row
        constexpr int PIECES = HD * (int)sizeof(T) / 16, EPP = 16 / (int)sizeof(T);
        if (!tile_in) return;
        for (int p = tid; p < rreal * creal * PIECES; p += NT) {
            const int row = p / PIECES, piece = p - row * PIECES;
            const int rr = row / creal, cc = row - rr * creal;
            const int y = gi + (r0 + rr) * dil, x = gj + (c0 + cc) * dil;
            const T* src = kv_src(which, y, x) + piece * EPP;
            *reinterpret_cast<uint4*>(tile + (size_t)rr * kpitch + (size_t)cc * STRIDE + piece * EPP) = *reinterpret_cast<const uint4*>(src);
        }
    };

    constexpr bool BF16 = sizeof(T) == 2;
    // bf16 AV runs on v_dot2 over PAIRS of neighbours: V is staged as row pairs, entry (rp, c) = 32 dwords
    // {V[2rp][c][ch], V[2rp+1][c][ch]} (+4 pad: 144 B pitch, conflict-free ds_read_b128), rows >= nr zero.  A window that
    // starts on an odd halo row uses the aligned pairs around it with probability 0 on the extra row, so one copy serves
    // both parities: 28 pairs x 32 dot2 instead of 49 rows x (32 unpack + 32 FMA).
    constexpr int VP = 36;                                                  // dwords per (row pair, column) entry
    auto load_tile_vpairs = [&]() {
        if constexpr (BF16) {
            if (!tile_in) return;
            uint32_t* vp = reinterpret_cast<uint32_t*>(tile);
            const int nrp = SKIP ? (rreal + 1) >> 1 : (nr + 2) >> 1;        // pairs covering rows 0 .. nr (the real rows when SKIP)
            for (int p = tid; p < nrp * creal * 4; p += NT) {
                const int e = p >> 2, piece = p & 3;
                const int rp = e / creal, cc = e - rp * creal;
                const int ra = 2 * rp, rb = ra + 1;
                const int x = gj + (c0 + cc) * dil;
                uint4 a = make_uint4(0u, 0u, 0u, 0u), bb = a;
                if (ra < rreal) a = *reinterpret_cast<const uint4*>(kv_src(2, gi + (r0 + ra) * dil, x) + piece * 8);
                if (rb < rreal) bb = *reinterpret_cast<const uint4*>(kv_src(2, gi + (r0 + rb) * dil, x) + piece * 8);
                uint32_t* dst = vp + (size_t)rp * vpitch + (size_t)cc * VP + piece * 8;
                // v_perm_b32: bytes 0-3 = second operand, 4-7 = first
                *reinterpret_cast<uint4*>(dst) = make_uint4(__builtin_amdgcn_perm(bb.x, a.x, 0x05040100u), __builtin_amdgcn_perm(bb.x, a.x, 0x07060302u),
                                                            __builtin_amdgcn_perm(bb.y, a.y, 0x05040100u), __builtin_amdgcn_perm(bb.y, a.y, 0x07060302u));
                *reinterpret_cast<uint4*>(dst + 4) = make_uint4(__builtin_amdgcn_perm(bb.z, a.z, 0x05040100u), __builtin_amdgcn_perm(bb.z, a.z, 0x07060302u),
                                                                __builtin_amdgcn_perm(bb.w, a.w, 0x05040100u), __builtin_amdgcn_perm(bb.w, a.w, 0x07060302u));
            }
        }
    };
    float q[BF16 ? 1 : HD];
    uint32_t qp[BF16 ? HD / 2 : 1];                                         // bf16: q stays packed, two channels per register
    if (valid) {
        const int y = gi + u * dil, x = gj + v * dil;
        const T* src = qkv + ((size_t)(b * Hs + y) * Ws + x) * tok + (size_t)h * HD;   // queries are real tokens
        if constexpr (BF16) {
#pragma unroll
            for (int p4 = 0; p4 < HD / 8; ++p4) {
                const uint4 w4 = *reinterpret_cast<const uint4*>(src + 8 * p4);
                qp[4 * p4] = w4.x; qp[4 * p4 + 1] = w4.y; qp[4 * p4 + 2] = w4.z; qp[4 * p4 + 3] = w4.w;
            }
        } else {
            load_row(src, q);
#pragma unroll
            for (int c = 0; c < HD; ++c) q[c] = q[c] * scale;               // q = q * scale before QK (NATTEN module)
        }
    }
    load_tile(1);
    __syncthreads();

    float logit[KS * KS];
    float mx = -3.0e38f;
    if (valid) {
        const int pbi = wi - u + (KS - 1), pbj = wj - v + (KS - 1);         // bias index of window member (0,0)
        float qk_pad = 0.0f;
        if constexpr (SKIP) {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
            const T* kp = pad_kv + ((size_t)heads + h) * HD;
#pragma unroll
            for (int p4 = 0; p4 < HD / 8; ++p4) {
                const uint4 w4 = *reinterpret_cast<const uint4*>(kp + 8 * p4);
                qk_pad = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, qp[4 * p4]), __builtin_bit_cast(bf2, w4.x), qk_pad, false);
                qk_pad = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, qp[4 * p4 + 1]), __builtin_bit_cast(bf2, w4.y), qk_pad, false);
                qk_pad = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, qp[4 * p4 + 2]), __builtin_bit_cast(bf2, w4.z), qk_pad, false);
                qk_pad = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, qp[4 * p4 + 3]), __builtin_bit_cast(bf2, w4.w), qk_pad, false);
            }
            qk_pad *= scale;
        }
#pragma unroll
        for (int ki = 0; ki < KS; ++ki) {
#pragma unroll
            for (int kj = 0; kj < KS; ++kj) {
                const T* krow = tile + (size_t)(wi - r0 + ki) * kpitch + (size_t)(wj - c0 + kj) * STRIDE;
                float acc = 0.0f;
                const bool real = !SKIP || ((wi - r0 + ki) < rreal && (wj - c0 + kj) < creal);
                if (!real) {
                    acc = qk_pad;
                } else if constexpr (BF16) {
                    // v_dot2c_f32_bf16: two channels per instruction, float32 accumulate; the bf16 q cannot carry the
                    // scale without another rounding, so the scale multiplies the finished dot product instead
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
#pragma unroll
                    for (int p4 = 0; p4 < HD / 8; ++p4) {
                        const uint4 w4 = *reinterpret_cast<const uint4*>(krow + 8 * p4);
                        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, qp[4 * p4]), __builtin_bit_cast(bf2, w4.x), acc, false);
                        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, qp[4 * p4 + 1]), __builtin_bit_cast(bf2, w4.y), acc, false);
                        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, qp[4 * p4 + 2]), __builtin_bit_cast(bf2, w4.z), acc, false);
                        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, qp[4 * p4 + 3]), __builtin_bit_cast(bf2, w4.w), acc, false);
                    }
                    acc *= scale;
                } else {
                    float kr[HD];
                    load_row(krow, kr);
#pragma unroll
                    for (int c = 0; c < HD; ++c) acc = fmaf(q[c], kr[c], acc);
                }
                acc += bias[(pbi + ki) * 13 + pbj + kj];
                logit[ki * KS + kj] = acc;
                mx = fmaxf(mx, acc);
                asm volatile("" ::: "memory");                              // keep one neighbour's row in flight, not 49
            }
        }
    }
    __syncthreads();
    if constexpr (BF16) load_tile_vpairs(); else load_tile(2);
    __syncthreads();
    if (valid) {
        float sum = 0.0f;
#pragma unroll
        // exp(x) = 2^(x * log2 e) on the hardware v_exp_f32 (1 ulp; arguments are <= 0, underflow flushes to 0): expf()'s
        // range handling is ~8 more instructions per neighbour in a VALU-bound kernel
        for (int t = 0; t < KS * KS; ++t) { logit[t] = __builtin_amdgcn_exp2f((logit[t] - mx) * 1.4426950408889634f); sum += logit[t]; }
        float o[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) o[c] = 0.0f;
        if constexpr (BF16) {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
            const uint32_t* vp = reinterpret_cast<const uint32_t*>(tile);
            const int hr = wi - r0;                                         // window start in halo rows
            const bool odd = hr & 1;
            const int rp0 = hr >> 1;
            float p_pad = 0.0f;                                             // SKIP: total probability of the padded neighbours
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int kj = 0; kj < KS; ++kj) {
                    // rows 2(rp0+t), 2(rp0+t)+1 are window rows (2t, 2t+1) for an even start, (2t-1, 2t) for an odd one
                    const float pe_a = logit[(2 * t) * KS + kj], pe_b = (2 * t + 1 < KS) ? logit[(2 * t + 1) * KS + kj] : 0.0f;
                    const float po_a = (t > 0) ? logit[(2 * t - 1) * KS + kj] : 0.0f, po_b = logit[(2 * t) * KS + kj];
                    float pa = odd ? po_a : pe_a, pb = odd ? po_b : pe_b;
                    if constexpr (SKIP) {                                   // real rows are a prefix: row b real => row a real
                        const bool c_real = (wj - c0 + kj) < creal;
                        const bool a_real = c_real && 2 * (rp0 + t) < rreal, b_real = c_real && 2 * (rp0 + t) + 1 < rreal;
                        p_pad += (a_real ? 0.0f : pa) + (b_real ? 0.0f : pb);
                        pa = a_real ? pa : 0.0f; pb = b_real ? pb : 0.0f;
                        if (!a_real) continue;                              // nothing staged here
                    }
                    const bf2 pp = {(__bf16)pa, (__bf16)pb};
                    const uint32_t* ent = vp + (size_t)(rp0 + t) * vpitch + (size_t)(wj - c0 + kj) * VP;
#pragma unroll
                    for (int p4 = 0; p4 < HD / 4; ++p4) {
                        const uint4 w4 = *reinterpret_cast<const uint4*>(ent + 4 * p4);
                        o[4 * p4]     = __builtin_amdgcn_fdot2_f32_bf16(pp, __builtin_bit_cast(bf2, w4.x), o[4 * p4], false);
                        o[4 * p4 + 1] = __builtin_amdgcn_fdot2_f32_bf16(pp, __builtin_bit_cast(bf2, w4.y), o[4 * p4 + 1], false);
                        o[4 * p4 + 2] = __builtin_amdgcn_fdot2_f32_bf16(pp, __builtin_bit_cast(bf2, w4.z), o[4 * p4 + 2], false);
                        o[4 * p4 + 3] = __builtin_amdgcn_fdot2_f32_bf16(pp, __builtin_bit_cast(bf2, w4.w), o[4 * p4 + 3], false);
                    }
                    asm volatile("" ::: "memory");
                }
            }
            if constexpr (SKIP) {
                float vr[HD];
                load_row(pad_kv + ((size_t)2 * heads + h) * HD, vr);
#pragma unroll
                for (int c = 0; c < HD; ++c) o[c] = fmaf(p_pad, vr[c], o[c]);
            }
        } else {
#pragma unroll
            for (int ki = 0; ki < KS; ++ki) {
#pragma unroll
                for (int kj = 0; kj < KS; ++kj) {
                    float vr[HD];
                    load_row(tile + (size_t)(wi - r0 + ki) * kpitch + (size_t)(wj - c0 + kj) * STRIDE, vr);
                    const float p = logit[ki * KS + kj];
#pragma unroll
                    for (int c = 0; c < HD; ++c) o[c] = fmaf(p, vr[c], o[c]);
                    asm volatile("" ::: "memory");
                }
            }
        }
        const float inv = 1.0f / sum;
        const int y = gi + u * dil, x = gj + v * dil;
        T* dst = out + ((size_t)(b * Hr + y) * Wr + x) * ((size_t)heads * HD) + (size_t)h * HD;
#pragma unroll
        for (int c = 0; c < HD; ++c) store_out(dst + c, o[c] * inv);
    }
}

template <typename T, int TILE, int THREADS, bool VPAD>
static int launch_variant_v(const void* qkv, const void* pad_kv, const float* rpb, void* out, int B, int H, int W, int Hr, int Wr, int heads, int dil,
                          float scale, hipStream_t stream) {
    constexpr int HALO = TILE + KS - 1, TPB = THREADS / (TILE * TILE);
    const int hs = (H + dil - 1) / dil, ws = (W + dil - 1) / dil;            // largest key sub-image
    const int hq = (Hr + dil - 1) / dil, wq = (Wr + dil - 1) / dil;          // largest query sub-image
    const long long total = (long long)((hq + TILE - 1) / TILE) * ((wq + TILE - 1) / TILE) * B * dil * dil;
    const dim3 grid((unsigned)((total + TPB - 1) / TPB), heads, 1);
    const int halo_r = hs < HALO ? hs : HALO, halo_c = ws < HALO ? ws : HALO;
    // row pitches: padding them to 256 B (so that the two tile rows of a 16-lane LDS group start on the same bank) was
    // measured and does not pay (0.701 vs 0.695 ms on the 64x64 level)
    const size_t kpitch_b = (size_t)halo_c * Row<T>::STRIDE * sizeof(T), vpitch_b = (size_t)halo_c * 144;
    size_t tile_bytes = (size_t)halo_r * kpitch_b;
    if (sizeof(T) == 2) {                                                    // bf16: V is staged as row pairs, 144 B per (pair, column)
        const size_t vpairs = (size_t)((halo_r + 2) / 2) * vpitch_b;
        if (vpairs > tile_bytes) tile_bytes = vpairs;
    }
    tile_bytes = (tile_bytes + 15) & ~(size_t)15;
    const size_t lds = (size_t)TPB * tile_bytes + 169 * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void*)na2d_fwd_kernel<T, TILE, THREADS, VPAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((na2d_fwd_kernel<T, TILE, THREADS, VPAD>), grid, dim3(THREADS), lds, stream, (const T*)qkv, (const T*)pad_kv, rpb, (T*)out, B, H, W, Hr,
                       Wr, heads, dil, scale, (int)total, halo_r, halo_c, (int)tile_bytes, (int)(kpitch_b / sizeof(T)), (int)(vpitch_b / 4));
    return (int)hipGetLastError();
}

template <typename T, int TILE, int THREADS>
static int launch_variant(const void* qkv, const void* pad_kv, const float* rpb, void* out, int B, int H, int W, int Hr, int Wr, int heads, int dil,
                          float scale, hipStream_t stream) {
    return pad_kv ? launch_variant_v<T, TILE, THREADS, true>(qkv, pad_kv, rpb, out, B, H, W, Hr, Wr, heads, dil, scale, stream)
                  : launch_variant_v<T, TILE, THREADS, false>(qkv, pad_kv, rpb, out, B, H, W, Hr, Wr, heads, dil, scale, stream);
}

template <typename T>
static int launch_typed(const void* qkv, const void* pad_kv, const float* rpb, void* out, int B, int H, int W, int Hr, int Wr, int heads, int dil,
                        float scale, hipStream_t stream) {
    const int hq = (Hr + dil - 1) / dil, wq = (Wr + dil - 1) / dil;
    // lane utilisation of each tiling on the query sub-image
    auto util = [&](int t) { return (double)(hq * wq) / ((double)((hq + t - 1) / t * t) * ((wq + t - 1) / t * t)); };
    int best = 16;
    if (util(8) > util(best) + 0.05) best = 8;
    if (util(4) > util(best) + 0.05) best = 4;
    if (best == 4) return launch_variant<T, 4, 128>(qkv, pad_kv, rpb, out, B, H, W, Hr, Wr, heads, dil, scale, stream);
    if (best == 8) return launch_variant<T, 8, 256>(qkv, pad_kv, rpb, out, B, H, W, Hr, Wr, heads, dil, scale, stream);
    return launch_variant<T, 16, 256>(qkv, pad_kv, rpb, out, B, H, W, Hr, Wr, heads, dil, scale, stream);
}

int na2d_launch(const void* qkv, const void* pad_kv, const float* rpb, void* out, int B, int H, int W, int Hr, int Wr, int heads, int dil, float scale,
                int dtype, hipStream_t stream) {
    // bfloat16 runs on the matrix cores (na2d_mfma.hip); PPNET_NA_VALU=1 keeps the v_dot2 kernel of this file (A/B runs)
    static const bool valu = getenv("PPNET_NA_VALU") != nullptr;
    const long long elems = (long long)B * (pad_kv ? Hr : H) * (pad_kv ? Wr : W) * 3 * heads * HD;       // 32-bit offsets inside
    // The MFMA kernel takes the layers whose dilation groups fill 16 x 16 query regions (measured on MI355X, batch 256, tools/
    // na_timing.py: 0.52 vs 0.66 ms at 64 x 64, 0.22 vs 0.26 at 32 x 32, 0.10 vs 0.11 at 16 x 16); on the 8 x 8 and 4 x 4 groups of
    // the dilated layers this file's kernel is the faster one (it skips the padded keys; 0.23 vs 0.27, 0.056 vs 0.060 ms).
    // PPNET_NA_MFMA=1 sends every bfloat16 launch to the MFMA kernel.
    static const bool all_mfma = getenv("PPNET_NA_MFMA") != nullptr;
    // launches whose dilation groups are exactly 7 x 7 (the grids padded to kernel * dilation): a dense 49-key attention per group,
    // one wave per (image, group, head) on the matrix cores (na2d_dense7.hip).  PPNET_NA_NO_DENSE7=1 switches it off (A/B runs).
    static const bool no_dense7 = getenv("PPNET_NA_NO_DENSE7") != nullptr;
    if (dtype == 1 && !valu && !no_dense7) {
        const int rc = na2d_dense7_launch(qkv, pad_kv, rpb, out, B, H, W, Hr, Wr, heads, dil, scale, stream);
        if (rc != -2) return rc;
    }
    // every other bfloat16 launch runs on the matrix cores, in the region size that covers its dilation groups cheapest
    // (na_region_size: rounds 2-3 sent groups that fill a 16 x 16 region poorly to this file's v_dot2 kernel — 2.23 ms against
    // 0.52 ms on the 11 x 11 groups of the dilation-3 layers at 512 x 512, and the 56 x 56 level of a 224 x 224 input likewise)
    (void)all_mfma;
    if (dtype == 1 && !valu && elems < 0xffffffffLL) return na2d_mfma_launch(qkv, pad_kv, rpb, out, B, H, W, Hr, Wr, heads, dil, scale, stream);
    return dtype == 0 ? launch_typed<float>(qkv, pad_kv, rpb, out, B, H, W, Hr, Wr, heads, dil, scale, stream)
                      : launch_typed<__hip_bfloat16>(qkv, pad_kv, rpb, out, B, H, W, Hr, Wr, heads, dil, scale, stream);
}

}  // namespace ppn
