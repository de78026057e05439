// na2d_dense7.hip — neighbourhood attention for launches whose dilation groups are exactly 7 x 7 or 8 x 8.  7 x 7 (H = W = 7 * dilation, the
// grids NATTEN zero-pads to kernel * dilation: DiNAT-B at 256 x 256 runs 10 of its 30 attention layers this way — dilation 16
// at 64 x 64, 8 at 32 x 32, 3 and 4 at 16 x 16, 2 at 8 x 8; reference SegNet/nat.py:111-120, dinat.py).  With 7 keys per axis the
// clamped window of EVERY query of a group is the whole group: the op is a dense 49-key attention per (image, group, head) with a
// bias that depends only on the (query, key) positions inside the group — no halo, no sharing between groups.  8 x 8 groups (dilation
// 4 at 32 x 32, 2 at 16 x 16, 1 at 8 x 8) are the same thing with 64 keys and each query's 49-key window applied through the table.
//
// One WAVE per (image, group, head); a workgroup's four waves share only their head's bias table (staged once, no barrier after):
//   K rows straight from global memory as the MFMA A operand (a padded key reads the qkv bias vector: "virtual padding",
//   ppn_na2d_fwd_vpad; slots 49..63 read a zero line), V rows by global_load_lds into the wave's private 4 KB of LDS and back
//   transposed (ds_read_b64_tr_b16), Q as the B operand (only the REAL queries: 16 per group at dilation 16 / 8 / 4 / 2),
//   S^T = K . Q^T (4 MFMAs per 16 queries), logits = S^T * scale * log2 e + T with T[head][query position][key slot] a small
//   float32 table (rpb gathered by relative position, -1e30 outside the window and on slots 49..63) that the workgroup builds for
//   its head in LDS at start (no per-launch table kernel or allocation), exact softmax over the 64 slots, O^T = V^T . P^T (4 MFMAs).  ~100 VALU instructions per 16
//   queries; one memory round trip per item, so the kernel runs at the latency x occupancy product: 3.3-4.2 TB/s of algorithmic bytes.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <cstdlib>
#include <algorithm>
#include "ppn_device.h"
#include "ppn_kernels.h"

namespace ppn {

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int HD7 = 32, D7_WAVES = 8;
}  // namespace

namespace {
__device__ __forceinline__ float d7_max3(float a, float b, float c) {     // (fmaxf would canonicalise every MFMA result first)
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
}  // namespace

template <int G7>
__global__ __launch_bounds__(64 * D7_WAVES, 4) void na2d_dense7_kernel(const __bf16* __restrict__ qkv, const __bf16* __restrict__ pad_kv,
                                                                    const float* __restrict__ rpb, __bf16* __restrict__ out, int B, int Hr, int Wr,
                                                                    int heads, int dil, float scale, int n_items, const __bf16* __restrict__ zero) {
    constexpr int NK = G7 * G7;                                            // 49 or 64 key slots in use
    constexpr int TP = 68;                                                 // table row pitch in floats: 16 consecutive rows start on 16 different bank groups
    constexpr bool FULL = G7 == 8;                                         // 8 x 8 groups: every slot is a stored token of every group
    __shared__ __attribute__((aligned(16))) unsigned char vimg_all[D7_WAVES][64 * 64];      // per wave: 64 key slots x 32 bf16 of V
    __shared__ __attribute__((aligned(16))) float tl[NK * TP];             // this workgroup's head of the bias / window table
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned char* vimg = vimg_all[wave];
    const int j = lane & 15, g = lane >> 4, q4 = j >> 2, p4 = j & 3;
    const uint32_t tokb = 3u * heads * HD7 * 2u;                             // bytes per token row; every byte offset below fits 32 bits (checked by the launcher)
    const float sl2 = scale * 1.4426950408889634f;
    const unsigned char* qkvb = reinterpret_cast<const unsigned char*>(qkv);
    const unsigned char* padb = reinterpret_cast<const unsigned char*>(pad_kv);
    const unsigned char* zerob = reinterpret_cast<const unsigned char*>(zero);
    // A workgroup keeps ONE head (its table rows live in LDS, read 4 x 16 bytes per query tile instead of 4 KB per item from
    // L2) and walks groups eight at a time, one per wave.  Workgroups w and w + 8 sit on the same XCD (round-robin dispatch) and
    // take the same groups with neighbouring heads, so the two heads of a 128-byte line still meet in one L2.
    const int h = (int)((blockIdx.x >> 3) % heads);
    const int slot0 = (int)(blockIdx.x / (8 * heads)) * 8 + (blockIdx.x & 7), nslots = (int)(gridDim.x / (8 * heads)) * 8;
    // T[u * G + v][slot] = rpb[h][kr - u + 6][kc - v + 6] / scale (the units of the raw product: the table is the logits' initial
    // accumulator) for key slot = kr * G + kc inside the query's window (start clamp(u - 3, 0, G - 7) per axis: the whole group for
    // G = 7), -1e30 outside it and on slots >= G * G.  Built by the workgroup from its head's 169 values: nothing is prepared or
    // allocated per launch (a captured graph holds this kernel alone)
    {
        const float inv_scale = 1.0f / scale;
        for (int i = threadIdx.x; i < NK * 64; i += 64 * D7_WAVES) {
            const int slot = i & 63, qp = i >> 6;
            float v = -1.0e30f;
            if (slot < NK) {
                const int u = qp / G7, w = qp - u * G7, kr = slot / G7, kc = slot - kr * G7;
                const int wu = min(max(u - 3, 0), G7 - 7), ww = min(max(w - 3, 0), G7 - 7);
                if (kr >= wu && kr < wu + 7 && kc >= ww && kc < ww + 7) v = rpb[(size_t)h * 169 + (kr - u + 6) * 13 + (kc - w + 6)] * inv_scale;
            }
            tl[qp * TP + slot] = v;
        }
    }
    __syncthreads();

    // ---- what of an item does not depend on the item (round 5: the counters said 308 vector and 171 scalar instructions per item,
    // two thirds of them this bookkeeping recomputed per item — three 64-bit divisions for the item's (image, group row, group
    // column), two 32-bit ones for its extent, eight slot decompositions with 64-bit pointer selects — and the kernel's time
    // followed its instruction count, not its occupancy: 1, 2 or 3 resident workgroups per CU ran within 20 %).
    // This lane's key slots: slot 16 t + j is its MFMA A row of key tile t, slot (64 it + lane) / 4 the V piece it copies in round it.
    const uint32_t rowb = (uint32_t)dil * Wr * tokb, colb = (uint32_t)dil * tokb;
    const uint32_t kh = (uint32_t)(heads + h) * (HD7 * 2), vh = (uint32_t)(2 * heads + h) * (HD7 * 2);
    uint32_t offK[4], offV[4], geoK = 0, geoV = 0;                         // byte offset from the group's first token; (row | column << 4) per slot
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int kk = 16 * t + j, kr = kk < NK ? kk / G7 : 15, kc = kk < NK ? kk % G7 : 0;
        offK[t] = (uint32_t)kr * rowb + (uint32_t)kc * colb + 16u * g;
        geoK |= (uint32_t)(kr | (kc << 4)) << (8 * t);
        const int vk = (t * 64 + lane) >> 2, vr = vk < NK ? vk / G7 : 15, vc = vk < NK ? vk % G7 : 0;
        offV[t] = (uint32_t)vr * rowb + (uint32_t)vc * colb + 16u * (lane & 3);
        geoV |= (uint32_t)(vr | (vc << 4)) << (8 * t);
    }
    // the padded token's K fragment (virtual padding: the qkv bias) — one per head, not one load per padded slot and item
    bf16x8 kpad = {};
    if (!FULL && padb) kpad = *reinterpret_cast<const bf16x8*>(padb + kh + 16 * g);
    // items of this wave: grp = (slot0 + n * nslots) * 8 + wave; (image, group row, group column) advance by a constant step with
    // carries instead of being divided out of grp per item
    const int dd = dil * dil, stride = nslots * D7_WAVES;
    int grp = slot0 * D7_WAVES + wave;
    int b = grp / dd, gi = (grp - b * dd) / dil, gj = grp - b * dd - gi * dil;
    const int sb = stride / dd, si = (stride - sb * dd) / dil, sj = stride - sb * dd - si * dil;
    const int hq_base = Hr / dil, hq_rem = Hr - hq_base * dil, wq_base = Wr / dil, wq_rem = Wr - wq_base * dil;
    int geo_hq = -1, geo_wq = -1;                                          // the extent the masks below and the V image's padding were made for
    uint32_t realm = 0, qmagic = 0;                                        // bit t: K slot t is a stored token; bit 4 + it: V piece it

    for (; grp < n_items; grp += stride) {
        const int hq = FULL ? G7 : hq_base + (gi < hq_rem), wq = FULL ? G7 : wq_base + (gj < wq_rem);   // stored rows / columns of the group
        const int nq = hq * wq;
        const uint32_t g0 = ((uint32_t)(b * Hr + gi) * Wr + gj) * tokb;      // the group's first token (wave-uniform)
        gj += sj;
        { const int c = gj >= dil; gj -= c ? dil : 0; gi += si + c; }
        { const int c = gi >= dil; gi -= c ? dil : 0; b += sb + c; }
        if (nq == 0) continue;                                               // a group of padding only: no queries (wave-uniform)
        const bool regeo = !FULL && (hq != geo_hq || wq != geo_wq);          // wave-uniform; false from the second item on unless the image
        if (regeo) {                                                         // cuts groups unequally (Hr, Wr not multiples of the dilation)
            geo_hq = hq; geo_wq = wq;
            realm = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                realm |= (uint32_t)((int)((geoK >> (8 * t)) & 15) < hq && (int)((geoK >> (8 * t + 4)) & 15) < wq) << t;
                realm |= (uint32_t)((int)((geoV >> (8 * t)) & 15) < hq && (int)((geoV >> (8 * t + 4)) & 15) < wq) << (4 + t);
            }
            qmagic = 65536u / (uint32_t)wq + 1u;                             // q / wq == (q * qmagic) >> 16 for q < 64
        }
        if (FULL) { realm = 0xffu; qmagic = 65536u / G7 + 1u; }
        // K fragments: stored token, padded token (the qkv bias: virtual padding) or zeros (slots 49..63)
        const unsigned char* gbase = qkvb + g0;                              // uniform base + 32-bit lane offset
        bf16x8 kf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (FULL) {
                kf[t] = *reinterpret_cast<const bf16x8*>(gbase + kh + offK[t]);
            } else {
                const bool real = (realm >> t) & 1u;
                const bf16x8 raw = *reinterpret_cast<const bf16x8*>(gbase + kh + (real ? offK[t] : 16u * g));     // (a lane without a token re-reads the first)
                const bool slot = 16 * t + j < NK;
                const bf16x8 other = slot ? kpad : bf16x8{};
                kf[t] = real ? raw : other;
            }
        }
        // V rows -> LDS (piece p = slot * 4 + chunk lives at byte 16 p): the stored tokens' pieces by LDS-DMA under their lanes' mask.
        // The pieces of padded tokens and of slots 49..63 do not depend on the item while its extent is the previous item's: they are
        // written when the extent changes (always for the first item) and left alone otherwise.
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const bool real = (realm >> (4 + it)) & 1u;
            if (FULL || real)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + vh + offV[it]),
                                                 (__attribute__((address_space(3))) void*)(vimg + it * 1024), 16, 0, 0);
            else if (regeo) {
                const bool slot = ((it * 64 + lane) >> 2) < NK;
                const unsigned char* src = (slot && padb) ? padb + vh : zerob;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * (lane & 3)),
                                                 (__attribute__((address_space(3))) void*)(vimg + it * 1024), 16, 0, 0);
            }
        }
        for (int qt = 0; qt * 16 < nq; ++qt) {                               // 16 real queries at a time (one tile unless dilation 3)
            const int qq = qt * 16 + j;
            const bool qvalid = qq < nq;
            const int qc = qvalid ? qq : nq - 1;                             // dead columns shadow the last real query, never stored
            const int u = (int)(((uint32_t)qc * qmagic) >> 16), v = qc - u * wq;
            const uint32_t trow = g0 + u * rowb + v * colb;                  // byte offset of the query's token row
            const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qkvb + (trow + (uint32_t)h * (HD7 * 2) + 16 * g));
            // bias / window rows of this query's position in the group, from LDS
            const float* tb = tl + (u * G7 + v) * TP + 4 * g;
            // bias + window mask are the MFMA's initial accumulator (raw-product units): S' = K Q^T + T out of the matrix pipe
            f32x4 s[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) s[t] = *reinterpret_cast<const f32x4*>(tb + 16 * t);
#pragma unroll
            for (int t = 0; t < 4; ++t) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t], qf, s[t], 0, 0, 0);
            // the maxima are inline assembly (no canonicalisation of 16 MFMA results), which the compiler's hazard recogniser does not
            // look into: an MFMA result read by a vector instruction fewer than 11 wait states behind the MFMA's issue is the
            // register's OLD content.  Every logit passes through this statement, which is the wait (na2d_halo16.hip).
            asm volatile("s_nop 7\n\ts_nop 4" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]));
            const float m0 = d7_max3(s[0][0], s[0][1], s[0][2]), m1 = d7_max3(s[0][3], s[1][0], s[1][1]), m2 = d7_max3(s[1][2], s[1][3], s[2][0]);
            const float m3 = d7_max3(s[2][1], s[2][2], s[2][3]), m4 = d7_max3(s[3][0], s[3][1], s[3][2]);
            float mx = d7_max3(d7_max3(m0, m1, m2), d7_max3(m3, m4, s[3][3]), m4);
            {   // the query's other three lane quarters (lane ^ 16, lane ^ 32) by register swaps: v_permlane16_swap / v_permlane32_swap
                // of the value with itself leave it and its partner's — no trip through the LDS crossbar (ds_bpermute) in the
                // chain every exponential waits for
                auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
                t = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
            }
            const float nm = -mx * sl2;                                      // p = 2^((S' - max) * scale * log2 e)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[t][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][e], sl2, nm));
            if (qt == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the V DMAs of this item have landed (same wave: no barrier)
            // O^T = V^T . P^T: k slot (g, e) of step ks is key 4g + (e & 3) of tile 2 ks + (e >> 2) — the S^T registers in place
            // the denominator: an all-ones A operand sums each query's (bfloat16-rounded) probabilities in the matrix pipe
            f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, lsum = {0.f, 0.f, 0.f, 0.f};
            const bf16x8 ones = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f32x4 pa = s[2 * ks], pb = s[2 * ks + 1];
                const bf16x8 pf = {(__bf16)pa[0], (__bf16)pa[1], (__bf16)pa[2], (__bf16)pa[3], (__bf16)pb[0], (__bf16)pb[1], (__bf16)pb[2], (__bf16)pb[3]};
                lsum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf, lsum, 0, 0, 0);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const unsigned char* va = vimg + (32 * ks + 4 * g + q4) * 64 + 8 * p4 + cb * 32;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va + 16 * 64));
                    const s16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vv), pf, o[cb], 0, 0, 0);
                }
            }
            if (qvalid) {
                const float inv = __builtin_amdgcn_rcpf(lsum[0]);   // (1 ulp; the result is rounded to bfloat16 next)
                // the output row of a token is a third of its qkv row: byte offset trow / 3
                unsigned char* dst = reinterpret_cast<unsigned char*>(out) + (trow / 3u + (uint32_t)h * (HD7 * 2) + 8 * g);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    *reinterpret_cast<uint2*>(dst + cb * 32) = make_uint2(pack_bf16x2(o[cb][0] * inv, o[cb][1] * inv), pack_bf16x2(o[cb][2] * inv, o[cb][3] * inv));
            }
        }
        // the next item's DMAs overwrite this wave's V image: its transposed reads must have returned (wave-local ordering)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

template <int G>
static int launch_dense_groups(const void* qkv, const void* pad_kv, const float* rpb, void* out, int B, int Hr, int Wr, int heads, int dil, float scale,
                               const __bf16* zero, hipStream_t stream) {
    const long long groups = (long long)B * dil * dil;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    // workgroups come in sets of 8 * heads (one head each, see the kernel): as many sets as fill the 3 workgroups (24 waves) a CU holds, at least one
    const long long per_set = 8LL * heads, want_sets = ((groups + D7_WAVES - 1) / D7_WAVES + 7) / 8;
    // as many sets as fill the workgroups a CU really holds (registers: 2 of the 7 x 7 form, 3 of the 8 x 8 one): a grid sized for
    // more leaves a last, half-empty round of workgroups
    static const int resident = [] {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)na2d_dense7_kernel<G>, 64 * D7_WAVES, 0) != hipSuccess || n < 1) n = 2;
        return n > 3 ? 3 : n;
    }();
    const char* occ_env = std::getenv("PPNET_D7_WG_PER_CU");                 // experiment: fewer resident workgroups per CU
    const int wg_per_cu = occ_env && occ_env[0] >= '1' && occ_env[0] <= '3' ? std::min(resident, occ_env[0] - '0') : resident;
    long long sets = std::max<long long>(1, std::min<long long>(want_sets, ((long long)cus * wg_per_cu) / per_set));
    hipLaunchKernelGGL(na2d_dense7_kernel<G>, dim3((unsigned)(sets * per_set)), dim3(64 * D7_WAVES), 0, stream, (const __bf16*)qkv, (const __bf16*)pad_kv,
                       rpb, (__bf16*)out, B, Hr, Wr, heads, dil, scale, (int)groups, zero);
    return (int)hipGetLastError();
}

// Serves a launch when every dilation group is 7 x 7 (H == W == 7 * dil: the padded grid of the vpad entry point, or a real grid of
// exactly that size) or 8 x 8 (H == W == 8 * dil, all tokens real: 64 keys per group, each query masked to its 49 by the table).
// Hr, Wr: the stored (real) token grid.  Returns -2 when the launch is not of this form.
int na2d_dense7_launch(const void* qkv, const void* pad_kv, const float* rpb, void* out, int B, int H, int W, int Hr, int Wr, int heads, int dil,
                       float scale, hipStream_t stream) {
    const bool g7 = H == 7 * dil && W == 7 * dil && (pad_kv || (Hr == H && Wr == W));
    const bool g8 = H == 8 * dil && W == 8 * dil && Hr == H && Wr == W;
    if (!g7 && !g8) return -2;
    const __bf16* zero = (const __bf16*)zero_line();
    if (!zero) return (int)hipErrorOutOfMemory;
    if ((long long)B * dil * dil <= 0 || (long long)B * dil * dil >= (1LL << 30) || dil > 4096 ||
        (long long)B * Hr * Wr * 3 * heads * HD7 * 2 >= (1LL << 32)) return -2;                   // 32-bit byte offsets and item numbers inside
    return g7 ? launch_dense_groups<7>(qkv, pad_kv, rpb, out, B, Hr, Wr, heads, dil, scale, zero, stream)
              : launch_dense_groups<8>(qkv, pad_kv, rpb, out, B, Hr, Wr, heads, dil, scale, zero, stream);
}

}  // namespace ppn
