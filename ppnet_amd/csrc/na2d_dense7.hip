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

template <int G7>
__global__ __launch_bounds__(64 * D7_WAVES, 6) void na2d_dense7_kernel(const __bf16* __restrict__ qkv, const __bf16* __restrict__ pad_kv,
                                                                    const float* __restrict__ rpb, __bf16* __restrict__ out, int B, int Hr, int Wr,
                                                                    int heads, int dil, float scale, long long n_items, const __bf16* __restrict__ zero) {
    constexpr int NK = G7 * G7;                                            // 49 or 64 key slots in use
    constexpr int TP = 68;                                                 // table row pitch in floats: 16 consecutive rows start on 16 different bank groups
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
    // this lane's key slots (the same for every item): slot 16 t + j as the MFMA A row, slot p >> 2 for the V pieces
    // A workgroup keeps ONE head (its table rows live in LDS, read 4 x 16 bytes per query tile instead of 4 KB per item from
    // L2) and walks groups four at a time, one per wave.  Workgroups w and w + 8 sit on the same XCD (round-robin dispatch) and
    // take the same groups with neighbouring heads, so the two heads of a 128-byte line still meet in one L2.
    const int h = (int)((blockIdx.x >> 3) % heads);
    const long long slot0 = (long long)(blockIdx.x / (8 * heads)) * 8 + (blockIdx.x & 7), nslots = (long long)(gridDim.x / (8 * heads)) * 8;
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
    for (long long gq = slot0; gq * D7_WAVES < n_items; gq += nslots) {
        const long long grp = gq * D7_WAVES + wave;                          // n_items = number of (image, group) pairs
        if (grp >= n_items) break;
        const int gj = (int)(grp % dil);
        const int gi = (int)((grp / dil) % dil);
        const int b = (int)(grp / ((long long)dil * dil));
        const int hq = gi < Hr ? (Hr - gi + dil - 1) / dil : 0, wq = gj < Wr ? (Wr - gj + dil - 1) / dil : 0;   // real rows / columns of the group
        const int nq = hq * wq;
        if (nq == 0) continue;                                               // a group of padding only: no queries (wave-uniform)
        const uint32_t g0 = ((uint32_t)(b * Hr + gi) * Wr + gj) * tokb;      // the group's first token (wave-uniform)
        const uint32_t rowb = (uint32_t)dil * Wr * tokb, colb = (uint32_t)dil * tokb;
        const uint32_t kh = (uint32_t)(heads + h) * (HD7 * 2), vh = (uint32_t)(2 * heads + h) * (HD7 * 2);

        // this lane's key slots, recomputed per item from an opaque copy of the lane id: as loop invariants they would pin eight
        // registers, and this kernel's throughput is its occupancy (80 VGPRs = 6 waves per SIMD; a spill would put scratch traffic
        // into the same in-order vmcnt queue as the loads below)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int j_o = lane_o & 15;
        // K fragments: real token, padded token (the qkv bias: virtual padding) or a zero line (slots 49..63)
        bf16x8 kf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kk = 16 * t + j_o, kr = kk < NK ? kk / G7 : 99, kc = kk % G7;       // row 99: beyond the group (slots 49..63)
            const bool slot = kr < G7, real = kr < hq && kc < wq;
            const unsigned char* src = real ? qkvb + (g0 + kr * rowb + kc * colb + kh) : ((slot && padb) ? padb + kh : zerob);
            kf[t] = *reinterpret_cast<const bf16x8*>(src + 16 * g);
        }
        // V rows -> LDS (piece p = slot * 4 + chunk lives at byte 16 p): four 1 KiB DMAs per wave
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int vk = (it * 64 + lane_o) >> 2, kr = vk < NK ? vk / G7 : 99, kc = vk % G7;
            const bool slot = kr < G7, real = kr < hq && kc < wq;
            const unsigned char* src = real ? qkvb + (g0 + kr * rowb + kc * colb + vh) : ((slot && padb) ? padb + vh : zerob);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * (lane & 3)),
                                             (__attribute__((address_space(3))) void*)(vimg + it * 1024), 16, 0, 0);
        }
        for (int qt = 0; qt * 16 < nq; ++qt) {                               // 16 real queries at a time (one tile unless dilation 3)
            const int qq = qt * 16 + j;
            const bool qvalid = qq < nq;
            const int qc = qvalid ? qq : nq - 1;                             // dead columns shadow the last real query, never stored
            const int u = qc / wq, v = qc - u * wq;
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
            float mx = -3.0e38f;
#pragma unroll
            for (int t = 0; t < 4; ++t) mx = fmaxf(mx, fmaxf(fmaxf(s[t][0], s[t][1]), fmaxf(s[t][2], s[t][3])));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
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
                const float inv = 1.0f / lsum[0];
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
    long long sets = std::max<long long>(1, std::min<long long>(want_sets, ((long long)cus * 3) / per_set));
    hipLaunchKernelGGL(na2d_dense7_kernel<G>, dim3((unsigned)(sets * per_set)), dim3(64 * D7_WAVES), 0, stream, (const __bf16*)qkv, (const __bf16*)pad_kv,
                       rpb, (__bf16*)out, B, Hr, Wr, heads, dil, scale, groups, zero);
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
    if ((long long)B * dil * dil <= 0 || (long long)B * Hr * Wr * 3 * heads * HD7 * 2 >= (1LL << 32)) return -2;     // 32-bit byte offsets inside
    return g7 ? launch_dense_groups<7>(qkv, pad_kv, rpb, out, B, Hr, Wr, heads, dil, scale, zero, stream)
              : launch_dense_groups<8>(qkv, pad_kv, rpb, out, B, Hr, Wr, heads, dil, scale, zero, stream);
}

}  // namespace ppn
