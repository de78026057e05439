// tokenizer_codes.hip — the first tokenizer convolution of SegNet (reference SegNet/nat.py:24-40: Conv2d(3, 64, 3, stride 2,
// padding 1)) evaluated straight from the planner's occupancy codes.  The network's input image is a three-colour palette
// (planning_seg.py:12-41 / process_map.py:120,128: free = white, start/goal marker = red, everything else = black), so a 3x3x3
// patch is fully described by nine palette indices and
//
//     conv(x)[co] = b[co] + sum_tap  L[co][tap][colour(tap)],      L[co][tap][c] = sum_ci w[co][ci][tap] * image_ci(c)
//
// is a product of the 64 x 28 table L (27 (tap, colour) columns + the bias column) with a one-hot column per output pixel —
// one MFMA K-step.  L is split hi + lo into two bfloat16 tables (the reference accumulates the 27 bf16 products in float32;
// hi + lo keeps 16 mantissa bits of each table entry), so a group of 16 output pixels x 64 channels costs eight
// v_mfma_f32_16x16x32_bf16 and the kernel is bound by its 128-byte-per-pixel output stream.  The normalised 3-channel image
// (ppn_grid_to_image) and the library convolution's separate bias pass are never materialised.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "../../include/ppnet_hip.h"
#include "ppn_device.h"
#include "ppn_kernels.h"

namespace ppn {

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

// output channel of (tile t, row i): a lane (pixel, group g) holds rows 4g..4g+3 of the four tiles = 16 consecutive channels
__device__ __forceinline__ int tok_channel(int t, int i) { return 16 * (i >> 2) + 4 * t + (i & 3); }

__device__ __forceinline__ uint32_t colour_bit(const uint8_t* __restrict__ g, int y, int x, int H, int W, int tap) {
    if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) return 0u;             // zero padding: no table column
    const uint32_t code = g[(size_t)y * W + x];
    const uint32_t c = code == PPN_GRID_FREE ? 0u : (code == PPN_GRID_MARK ? 1u : 2u);
    return 1u << (3 * tap + c);
}
}  // namespace

// lut: [2][64][32] bfloat16 (hi table, lo table); row = output channel, column k = 3 * (ky * 3 + kx) + colour, k = 27 the bias
// (hi table only), 28..31 zero.  out: [B][H/2][W/2][64] bfloat16.  One wave per 16 consecutive output pixels of a row.
__global__ __launch_bounds__(256) void tokenizer_codes_kernel(const uint8_t* __restrict__ grid, const __hip_bfloat16* __restrict__ lut,
                                                              __hip_bfloat16* __restrict__ out, int B, int H, int W, long long groups) {
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4;
    const int Ho = H / 2, Wo = W / 2, gpr = Wo / 16;
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int co = tok_channel(t, p);
        ah[t] = *reinterpret_cast<const bf16x8*>(lut + (size_t)co * 32 + 8 * g);
        al[t] = *reinterpret_cast<const bf16x8*>(lut + (size_t)(64 + co) * 32 + 8 * g);
    }
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    for (long long grp = wave0; grp < groups; grp += nwaves) {
        const int xg = (int)(grp % gpr), oy = (int)((grp / gpr) % Ho), b = (int)(grp / ((long long)gpr * Ho));
        const int ox = xg * 16 + p;
        const uint8_t* gb = grid + (size_t)b * H * W;
        uint32_t m = 1u << 27;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) m |= colour_bit(gb, 2 * oy - 1 + ky, 2 * ox - 1 + kx, H, W, ky * 3 + kx);
        const uint32_t bits = (m >> (8 * g)) & 0xffu;
        u32x4_t hot;                                                       // eight bf16 ones / zeros: this lane's slice of the one-hot column
#pragma unroll
        for (int q = 0; q < 4; ++q) hot[q] = ((bits >> (2 * q)) & 1u) * 0x3F80u + ((bits >> (2 * q + 1)) & 1u) * 0x3F800000u;
        const bf16x8 bv = __builtin_bit_cast(bf16x8, hot);
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t], bv, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[t], bv, acc[t], 0, 0, 0);
        }
        uint4* dst = reinterpret_cast<uint4*>(out + (((size_t)b * Ho + oy) * Wo + ox) * 64 + 16 * g);
        dst[0] = make_uint4(pack_bf16x2(acc[0][0], acc[0][1]), pack_bf16x2(acc[0][2], acc[0][3]), pack_bf16x2(acc[1][0], acc[1][1]),
                            pack_bf16x2(acc[1][2], acc[1][3]));
        dst[1] = make_uint4(pack_bf16x2(acc[2][0], acc[2][1]), pack_bf16x2(acc[2][2], acc[2][3]), pack_bf16x2(acc[3][0], acc[3][1]),
                            pack_bf16x2(acc[3][2], acc[3][3]));
    }
}

// The whole tokenizer — both convolutions and the LayerNorm (SegNet/nat.py:17-46) — from the occupancy codes in ONE kernel.
// Per 16 tokens (output pixels of the second, stride-2 convolution) and per tap of that convolution, the first convolution's
// output at the tap's position is the palette-table product above (8 MFMAs for its 64 channels), and its accumulator layout — lane
// = pixel, four consecutive channels per register group of each channel tile — is taken AS the second convolution's B operand: the
// k order of that product is free, so (tap, channel-tile pair) is one k-step and the weights are packed to match (147 KB, resident
// in LDS for the kernel's lifetime; the lanes of a fragment read touch one contiguous KiB).  16 MFMAs per tap for the 128 output
// channels; then bias, LayerNorm over the token's 128 channels (32 per lane, 4 lanes per token) and 32-byte stores.  The 64-channel
// half-resolution tensor (537 MB at batch 256) is never written, and no library convolution is left in SegNet's path.
//   lut: [2][64][32] bfloat16 as tokenizer_codes_kernel but rows in natural channel order;
//   w2p: [8 tiles][9 taps][2][16 rows][32 k-slots] bfloat16, row (tile nt, i) = output channel (nt>>2)*64 + 16*(i>>2) + 4*(nt&3) + (i&3),
//        slot 8g + e of step s = input channel (2s + (e>>2)) * 16 + 4g + (e&3);   vec: [3][128] float32 = conv2 bias, LN weight, LN bias.
constexpr int TOKF_THREADS = 512;
constexpr int TOKF_W2_BYTES = 8 * 9 * 2 * 16 * 32 * 2;
constexpr int TOKF_LDS = TOKF_W2_BYTES + 3 * 128 * 4;

__global__ __launch_bounds__(TOKF_THREADS, 1) void tokenizer_fused_kernel(const uint8_t* __restrict__ grid, const __hip_bfloat16* __restrict__ lut,
                                                                          const __hip_bfloat16* __restrict__ w2p, const float* __restrict__ vec,
                                                                          __hip_bfloat16* __restrict__ out, int B, int H, int W, long long groups, float eps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tl[];
    float* vl = reinterpret_cast<float*>(tl + TOKF_W2_BYTES);
    // A fragment (16 rows x 64 bytes) is laid out [lane quarter g][row p] in LDS, not [p][g] as in memory: lane (p, g) reads its 16
    // bytes at g * 256 + p * 16, and the 16-lane groups a ds_read_b128 is served in — rows {0-3, 12-15} of one quarter with rows 4-11
    // of the next — then fall on 16 different bank groups (row-major they met two by two: SQ_LDS_BANK_CONFLICT was 45 % of the
    // kernel's LDS cycles, and it reads 144 fragments per 16 tokens)
    for (int i = threadIdx.x; i < TOKF_W2_BYTES / 16; i += TOKF_THREADS)
        reinterpret_cast<uint4*>(tl)[(i & ~63) + (i & 3) * 16 + ((i >> 2) & 15)] = reinterpret_cast<const uint4*>(w2p)[i];
    for (int i = threadIdx.x; i < 3 * 128; i += TOKF_THREADS) vl[i] = vec[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4;
    const int Ht = H / 4, Wt = W / 4, gpr = Wt / 16;
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        ah[t] = *reinterpret_cast<const bf16x8*>(lut + (size_t)(t * 16 + p) * 32 + 8 * g);
        al[t] = *reinterpret_cast<const bf16x8*>(lut + (size_t)(64 + t * 16 + p) * 32 + 8 * g);
    }
    const unsigned char* wrow = tl + g * 256 + p * 16;                      // this lane's 16 bytes of every weight fragment
    const long long wave0 = (long long)blockIdx.x * (TOKF_THREADS / 64) + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * (TOKF_THREADS / 64);
    for (long long grp = wave0; grp < groups; grp += nwaves) {
        const uint32_t g32 = (uint32_t)grp, bimg = g32 / (uint32_t)(gpr * Ht), rem = g32 - bimg * (uint32_t)(gpr * Ht);
        const int ty = (int)(rem / (uint32_t)gpr), tx = (int)(rem - (uint32_t)ty * gpr) * 16 + p;
        const uint8_t* gb = grid + (size_t)bimg * H * W;
        // the 7 x 8 code patch of this token (rows 4ty-3 .. 4ty+3, columns 4tx-4 .. 4tx+3) as one-hot colours, 3 bits per code:
        // 1 free, 2 marker, 4 anything else, 0 outside the image (zero padding of the first convolution: no table column)
        uint32_t crow[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int y = 4 * ty - 3 + r;
            const bool rin = y >= 0 && y < H;
            const int cy = min(max(y, 0), H - 1);
            uint32_t bits = 0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int x = 4 * tx - 4 + 4 * k;
                const bool in = rin && x >= 0 && x < W;
                const uint32_t d = *reinterpret_cast<const uint32_t*>(gb + (size_t)cy * W + min(max(x, 0), W - 4));   // unconditional, clamped
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t code = (d >> (8 * q)) & 0xffu;
                    const uint32_t hot = code == PPN_GRID_FREE ? 1u : (code == PPN_GRID_MARK ? 2u : 4u);
                    bits |= (in ? hot : 0u) << (3 * (4 * k + q));
                }
            }
            crow[r] = bits;
        }
        f32x4 acc[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t2 = 0; t2 < 9; ++t2) {
            const int ky = t2 / 3, kx = t2 % 3;
            // first-convolution position (2ty + ky - 1, 2tx + kx - 1) of the half-resolution grid; its 3 x 3 code window is patch rows
            // 2ky .. 2ky+2, patch columns 2kx+1 .. 2kx+3: nine consecutive one-hot bits per row = the table's column order
            const int py = 2 * ty + ky - 1, px = 2 * tx + kx - 1;
            const bool inside = py >= 0 && py < H / 2 && px >= 0 && px < W / 2;
            const uint32_t m = ((crow[2 * ky] >> (3 * (2 * kx + 1))) & 0x1ffu) | (((crow[2 * ky + 1] >> (3 * (2 * kx + 1))) & 0x1ffu) << 9) |
                               (((crow[2 * ky + 2] >> (3 * (2 * kx + 1))) & 0x1ffu) << 18) | (1u << 27);
            const uint32_t bits = (m >> (8 * g)) & 0xffu;
            u32x4_t hotv;
#pragma unroll
            for (int q = 0; q < 4; ++q) hotv[q] = ((bits >> (2 * q)) & 1u) * 0x3F80u + ((bits >> (2 * q + 1)) & 1u) * 0x3F800000u;
            const bf16x8 bv = __builtin_bit_cast(bf16x8, hotv);
            u32x4_t f[2];                                                   // the 64 channels of the position as two B-operand k-steps
#pragma unroll
            for (int sstep = 0; sstep < 2; ++sstep) {
                f32x4 ya = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[2 * sstep], bv, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                ya = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[2 * sstep], bv, ya, 0, 0, 0);
                f32x4 yb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[2 * sstep + 1], bv, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                yb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[2 * sstep + 1], bv, yb, 0, 0, 0);
                f[sstep] = u32x4_t{pack_bf16x2(ya[0], ya[1]), pack_bf16x2(ya[2], ya[3]), pack_bf16x2(yb[0], yb[1]), pack_bf16x2(yb[2], yb[3])};
                if (!inside) f[sstep] = u32x4_t{0u, 0u, 0u, 0u};            // the second convolution's zero padding
            }
            const bf16x8 f0 = __builtin_bit_cast(bf16x8, f[0]), f1 = __builtin_bit_cast(bf16x8, f[1]);
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) {
                const unsigned char* wp = wrow + ((nt * 9 + t2) * 2) * 1024;
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(wp), f0, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(wp + 1024), f1, acc[nt], 0, 0, 0);
                if (nt & 1) __builtin_amdgcn_sched_barrier(0);              // four fragment reads in flight are enough: hoisting all 144 spills
            }
        }
        // bias, LayerNorm over the token's 128 channels (this lane: channels qd * 64 + 16 g + [0, 16) of both quads), store
        float v[2][16];
        float sum = 0.f;
#pragma unroll
        for (int qd = 0; qd < 2; ++qd)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(vl + qd * 64 + 16 * g + 4 * tt);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[qd][4 * tt + e] = acc[4 * qd + tt][e] + bb[e]; sum += v[qd][4 * tt + e]; }
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / 128.0f);
        float sq = 0.f;
#pragma unroll
        for (int qd = 0; qd < 2; ++qd)
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float dlt = v[qd][e] - mean; sq = fmaf(dlt, dlt, sq); }
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);
        const float rstd = rsqrtf(sq * (1.0f / 128.0f) + eps);
        __hip_bfloat16* orow = out + (((size_t)bimg * Ht + ty) * Wt + tx) * 128 + 16 * g;
#pragma unroll
        for (int qd = 0; qd < 2; ++qd) {
            uint32_t o[8];
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const int c = qd * 64 + 16 * g + e;
                const float a = fmaf((v[qd][e] - mean) * rstd, vl[128 + c], vl[256 + c]);
                const float b = fmaf((v[qd][e + 1] - mean) * rstd, vl[128 + c + 1], vl[256 + c + 1]);
                o[e >> 1] = pack_bf16x2(a, b);
            }
            uint4* dst = reinterpret_cast<uint4*>(orow + qd * 64);
            dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
            dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
        }
    }
}

int tokenizer_fused_launch(const uint8_t* grid, const void* lut, const void* w2p, const float* vec, void* out, int B, int H, int W, float eps,
                           hipStream_t stream) {
    static DeviceOnce attr;
    if (const int e = dynamic_lds_once(attr, (const void*)tokenizer_fused_kernel, TOKF_LDS)) return e;
    const long long groups = (long long)B * (H / 4) * (W / 4 / 16);
    if (groups >= (1LL << 31)) return (int)hipErrorInvalidValue;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const long long want = (groups + TOKF_THREADS / 64 - 1) / (TOKF_THREADS / 64);
    hipLaunchKernelGGL(tokenizer_fused_kernel, dim3((unsigned)(want < cus ? want : cus)), dim3(TOKF_THREADS), TOKF_LDS, stream, grid, (const __hip_bfloat16*)lut,
                       (const __hip_bfloat16*)w2p, vec, (__hip_bfloat16*)out, B, H, W, groups, eps);
    return (int)hipGetLastError();
}

int tokenizer_codes_launch(const uint8_t* grid, const void* lut, void* out, int B, int H, int W, hipStream_t stream) {
    const long long groups = (long long)B * (H / 2) * (W / 2 / 16);
    long long blocks = (groups + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(tokenizer_codes_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, grid, (const __hip_bfloat16*)lut, (__hip_bfloat16*)out, B, H, W,
                       groups);
    return (int)hipGetLastError();
}

}  // namespace ppn
