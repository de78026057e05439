// gennet_conv.hip — GenNet's 24-channel stride-2 stages on MFMA (reference GenNet/networks/ae_vit.py:24-36 encoder
// Conv2d(dim, dim, 3, 2, 1) + BN + LeakyReLU, :44-55 decoder ConvTranspose2d(dim, dim, 3, 2, 1, output_padding=1) + BN +
// LeakyReLU; dim = 24, predict.py:36).  NHWC bfloat16, BatchNorm folded into weight / bias by the caller.
//
// With 24 channels these layers are bandwidth-bound by a wide margin once the multiply-adds sit on the matrix pipe
// (v_mfma_f32_16x16x32_bf16): 9 * 24 = 216 (conv) or 4 * 24 = 96 (transposed conv: the 2x2 input block under a 2x2 output block)
// reduction elements per output, padded to 224 / 96.  No LDS: the weights live in registers for the whole kernel (the MFMA's A
// operand, rows = output channel), the activation fragments (B operand, columns = 16 consecutive pixels) are 16-byte loads
// straight from the NHWC image — a chunk of 8 channels of one tap — served by L1 / L2 for the overlapping taps.  A lane ends
// up with 4 consecutive output channels of one pixel: 8-byte stores.
#include <hip/hip_runtime.h>
#include "ppn_device.h"
#include "ppn_kernels.h"

namespace ppn {

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ bf16x8 ld8(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 zero8() { return bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }

__device__ __forceinline__ void store4(__bf16* dst, const f32x4 v, const float4 b, float slope) {
    float o[4] = {v[0] + b.x, v[1] + b.y, v[2] + b.z, v[3] + b.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = o[r] > 0.f ? o[r] : o[r] * slope;
    *reinterpret_cast<bf16x4*>(dst) = bf16x4{(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
}
}  // namespace

constexpr int GC = 24;                 // channels
constexpr int GROUPS = 2;              // 16-pixel groups in flight per wave iteration

// Encoder stage: y[b][oy][ox][co] = lrelu(sum_{ky,kx,ci} x[b][2oy+ky-1][2ox+kx-1][ci] * w[co][(ky*3+kx)*24+ci] + bias[co]).
// wk: [32][224] bf16, rows co (24..31 zero), k = tap*24 + ci (216..223 zero).  bias: [32] float32.
__global__ __launch_bounds__(256) void gennet_enc_conv_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ wk,
                                                              const float* __restrict__ bias, __bf16* __restrict__ y, int B, int H, int W,
                                                              float slope) {
    const int lane = threadIdx.x & 63, pl = lane & 15, g = lane >> 4;
    const int Ho = H >> 1, Wo = W >> 1;
    const long long total = (long long)B * Ho * Wo;
    bf16x8 wa[2][7];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) wa[nt][ks] = ld8(wk + (nt * 16 + pl) * 224 + ks * 32 + g * 8);
    const float4 b0 = *reinterpret_cast<const float4*>(bias + 4 * g), b1 = *reinterpret_cast<const float4*>(bias + 16 + 4 * g);
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    for (long long base = wave * (16 * GROUPS); base < total; base += nwaves * (16 * GROUPS)) {
        bf16x8 fb[GROUPS][7];
#pragma unroll
        for (int q = 0; q < GROUPS; ++q) {
            long long pix = base + q * 16 + pl;
            const bool live = pix < total;
            if (!live) pix = total - 1;
            const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho);
            const long long img = pix / ((long long)Wo * Ho);
            const __bf16* xb = x + img * H * W * GC;
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {
                const int c = ks * 4 + g;                        // chunk of 8 reduction elements: tap c / 3, channels (c % 3) * 8 ..
                const int tap = c / 3, ci0 = (c - tap * 3) * 8;
                const int iy = 2 * oy + tap / 3 - 1, ix = 2 * ox + tap % 3 - 1;
                const bool in = live && c < 27 && iy >= 0 && iy < H && ix >= 0 && ix < W;
                fb[q][ks] = in ? ld8(xb + ((long long)iy * W + ix) * GC + ci0) : zero8();
            }
        }
#pragma unroll
        for (int q = 0; q < GROUPS; ++q) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][ks], fb[q][ks], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[1][ks], fb[q][ks], a1, 0, 0, 0);
            }
            const long long pix = base + q * 16 + pl;
            if (pix < total) {
                __bf16* dst = y + pix * GC;
                store4(dst + 4 * g, a0, b0, slope);                       // co = 4g .. 4g+3
                if (g < 2) store4(dst + 16 + 4 * g, a1, b1, slope);       // co = 16 + 4g ..; 24..31 are padding
            }
        }
    }
}

// Decoder stage (ConvTranspose2d 3x3, stride 2, padding 1, output_padding 1): the 2x2 output block at (2iy+a, 2ix+b) reads the
// 2x2 input block (iy+dy, ix+dx):  class (a, b) uses tap ky = 1 (a = 0, dy = 0) or ky = 2 / 0 (a = 1, dy = 0 / 1), same in x.
// wt: [4 classes][32 co][96] bf16, k = (dy*2+dx)*24 + ci, zero where a class has no tap.  bias [32] float32.
__global__ __launch_bounds__(256) void gennet_dec_conv_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ wt,
                                                              const float* __restrict__ bias, __bf16* __restrict__ y, int B, int H, int W,
                                                              float slope) {
    const int lane = threadIdx.x & 63, pl = lane & 15, g = lane >> 4;
    const long long total = (long long)B * H * W;                 // input positions; each makes a 2x2 output block
    bf16x8 wa[4][2][3];
#pragma unroll
    for (int cl = 0; cl < 4; ++cl)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) wa[cl][nt][ks] = ld8(wt + ((cl * 32) + nt * 16 + pl) * 96 + ks * 32 + g * 8);
    const float4 b0 = *reinterpret_cast<const float4*>(bias + 4 * g), b1 = *reinterpret_cast<const float4*>(bias + 16 + 4 * g);
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const int Wo = 2 * W;
    for (long long base = wave * 16; base < total; base += nwaves * 16) {
        long long pos = base + pl;
        const bool live = pos < total;
        if (!live) pos = total - 1;
        const int ix = (int)(pos % W), iy = (int)((pos / W) % H);
        const long long img = pos / ((long long)W * H);
        const __bf16* xb = x + img * H * W * GC;
        bf16x8 fb[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int c = ks * 4 + g;                            // chunk: input pixel c / 3 of the 2x2 block, channels (c % 3) * 8 ..
            const int p = c / 3, ci0 = (c - p * 3) * 8;
            const int yy = iy + (p >> 1), xx = ix + (p & 1);
            const bool in = live && yy < H && xx < W;
            fb[ks] = in ? ld8(xb + ((long long)yy * W + xx) * GC + ci0) : zero8();
        }
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cl][0][ks], fb[ks], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cl][1][ks], fb[ks], a1, 0, 0, 0);
            }
            if (live) {
                __bf16* dst = y + ((img * 2 * H + (2 * iy + (cl >> 1))) * Wo + (2 * ix + (cl & 1))) * GC;
                store4(dst + 4 * g, a0, b0, slope);
                if (g < 2) store4(dst + 16 + 4 * g, a1, b1, slope);
            }
        }
    }
}

int gennet_enc_conv_launch(const void* x, const void* wk, const float* bias, void* y, int B, int H, int W, float slope, hipStream_t stream) {
    const long long total = (long long)B * (H / 2) * (W / 2);
    long long blocks = (total + 16 * GROUPS * 4 - 1) / (16 * GROUPS * 4);
    if (blocks > 256 * 16) blocks = 256 * 16;                    // grid-stride: the register-resident weights are loaded once per wave
    hipLaunchKernelGGL(gennet_enc_conv_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const __bf16*)x, (const __bf16*)wk, bias,
                       (__bf16*)y, B, H, W, slope);
    return (int)hipGetLastError();
}

int gennet_dec_conv_launch(const void* x, const void* wt, const float* bias, void* y, int B, int H, int W, float slope, hipStream_t stream) {
    const long long total = (long long)B * H * W;
    long long blocks = (total + 63) / 64;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(gennet_dec_conv_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const __bf16*)x, (const __bf16*)wt, bias,
                       (__bf16*)y, B, H, W, slope);
    return (int)hipGetLastError();
}

}  // namespace ppn
