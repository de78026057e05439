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

int tokenizer_codes_launch(const uint8_t* grid, const void* lut, void* out, int B, int H, int W, hipStream_t stream) {
    const long long groups = (long long)B * (H / 2) * (W / 2 / 16);
    long long blocks = (groups + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(tokenizer_codes_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, grid, (const __hip_bfloat16*)lut, (__hip_bfloat16*)out, B, H, W,
                       groups);
    return (int)hipGetLastError();
}

}  // namespace ppn
