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
// a0 / a1: rows 4g .. 4g+3 of the two output-channel tiles.  The rows of a tile are ASSIGNED to channels so that they are this
// lane's channels 8g .. 8g+3 and 8g+4 .. 8g+7 (row_channel below): one 16-byte store per lane quarter g < 3 instead of an 8-byte
// store per tile (the stages are store-bound: 805 MB out of the last decoder stage).
__device__ __forceinline__ void store8(__bf16* dst, const f32x4 a0, const f32x4 a1, const float4 b0, const float4 b1, float slope) {
    float o[8] = {a0[0] + b0.x, a0[1] + b0.y, a0[2] + b0.z, a0[3] + b0.w, a1[0] + b1.x, a1[1] + b1.y, a1[2] + b1.z, a1[3] + b1.w};
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = o[r] > 0.f ? o[r] : o[r] * slope;
    *reinterpret_cast<bf16x8*>(dst) = bf16x8{(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3], (__bf16)o[4], (__bf16)o[5], (__bf16)o[6], (__bf16)o[7]};
}
// output channel of row `row` (0 .. 15) of tile nt: 8 (row / 4) + 4 nt + row % 4   (24 .. 31: the zero rows of the padded weights)
__device__ __forceinline__ int row_channel(int nt, int row) { return 8 * (row >> 2) + 4 * nt + (row & 3); }
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
        for (int ks = 0; ks < 7; ++ks) wa[nt][ks] = ld8(wk + row_channel(nt, pl) * 224 + ks * 32 + g * 8);
    const float4 b0 = *reinterpret_cast<const float4*>(bias + 8 * g), b1 = *reinterpret_cast<const float4*>(bias + 8 * g + 4);
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    for (long long base = wave * (16 * GROUPS); base < total; base += nwaves * (16 * GROUPS)) {
        bf16x8 fb[GROUPS][7];
#pragma unroll
        for (int q = 0; q < GROUPS; ++q) {
            long long pix = base + q * 16 + pl;
            const bool live = pix < total;
            if (!live) pix = total - 1;
            // 32-bit divisions (the launcher checks the pixel count): the 64-bit form costs hundreds of instructions per group
            const uint32_t p32 = (uint32_t)pix, img = p32 / (uint32_t)(Wo * Ho), rem = p32 - img * (uint32_t)(Wo * Ho);
            const int oy = (int)(rem / (uint32_t)Wo), ox = (int)(rem - (uint32_t)oy * Wo);
            const __bf16* xb = x + (size_t)img * H * W * GC;
#pragma unroll
            for (int ks = 0; ks < 7; ++ks) {
                const int c = ks * 4 + g;                        // chunk of 8 reduction elements: tap c / 3, channels (c % 3) * 8 ..
                const int tap = c / 3, ci0 = (c - tap * 3) * 8;
                const int iy = 2 * oy + tap / 3 - 1, ix = 2 * ox + tap % 3 - 1;
                const bool in = live && c < 27 && iy >= 0 && iy < H && ix >= 0 && ix < W;
                fb[q][ks] = in ? ld8(xb + (uint32_t)((iy * W + ix) * GC + ci0)) : zero8();
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
            if (pix < total && g < 3) store8(y + pix * GC + 8 * g, a0, a1, b0, b1, slope);      // co = 8g .. 8g+7; g = 3 is padding
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
            for (int ks = 0; ks < 3; ++ks) wa[cl][nt][ks] = ld8(wt + ((cl * 32) + row_channel(nt, pl)) * 96 + ks * 32 + g * 8);
    const float4 b0 = *reinterpret_cast<const float4*>(bias + 8 * g), b1 = *reinterpret_cast<const float4*>(bias + 8 * g + 4);
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const int Wo = 2 * W;
    for (long long base = wave * 16; base < total; base += nwaves * 16) {
        long long pos = base + pl;
        const bool live = pos < total;
        if (!live) pos = total - 1;
        const uint32_t p32 = (uint32_t)pos, img = p32 / (uint32_t)(W * H), rem = p32 - img * (uint32_t)(W * H);   // 32-bit divisions, see the encoder
        const int iy = (int)(rem / (uint32_t)W), ix = (int)(rem - (uint32_t)iy * W);
        const __bf16* xb = x + (size_t)img * H * W * GC;
        bf16x8 fb[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int c = ks * 4 + g;                            // chunk: input pixel c / 3 of the 2x2 block, channels (c % 3) * 8 ..
            const int p = c / 3, ci0 = (c - p * 3) * 8;
            const int yy = iy + (p >> 1), xx = ix + (p & 1);
            const bool in = live && yy < H && xx < W;
            fb[ks] = in ? ld8(xb + (uint32_t)((yy * W + xx) * GC + ci0)) : zero8();
        }
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cl][0][ks], fb[ks], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cl][1][ks], fb[ks], a1, 0, 0, 0);
            }
            if (live && g < 3)
                store8(y + ((size_t)(img * 2 * H + (2 * iy + (cl >> 1))) * Wo + (2 * ix + (cl & 1))) * GC + 8 * g, a0, a1, b0, b1, slope);
        }
    }
}

// The LAST decoder stage fused with GenNet's final convolution (ae_vit.py:44-58: ConvTranspose2d + BN + LeakyReLU, then
// Conv2d(dim, 1, 3, 1, 1)): the 24-channel full-resolution tensor (805 MB per batch of 256 at 256 x 256) is neither written nor read.
// A workgroup owns a 32 x 16 tile of the final image.  Its 34 x 18 halo lies inside the 2 x 2 output blocks of 18 x 10 decoder
// input positions; for each group of 16 positions the decoder product runs as in gennet_dec_conv_kernel, and its accumulators —
// bias, LeakyReLU, rounded to bfloat16 exactly as the stored tensor was — ARE the B operand (pixel = column, lane quarter g =
// channels 8g .. 8g+7) of the final convolution's tap-response product of small_conv.hip (T[tap][pixel] = sum_c w1[tap][c] x[pixel][c],
// float32 weights as hi + lo bfloat16): two more MFMAs per class, T to LDS (zero for pixels outside the image: the convolution's
// padding), then y[i][j] = bias1 + sum of the nine shifted taps.  Same values in the same order as the two-kernel path: bit-identical.
namespace {
constexpr int DF_W = 32, DF_H = 16;                                        // final-resolution outputs per workgroup
constexpr int DF_PW = DF_W / 2 + 2, DF_PH = DF_H / 2 + 2, DF_POS = DF_PW * DF_PH;      // decoder input positions: 18 x 10
constexpr int DF_TW = 2 * DF_PW, DF_TH = 2 * DF_PH;                        // T region: 36 x 20 pixels, origin (i0 - 2, j0 - 2)
constexpr int DF_GROUPS = (DF_POS + 15) / 16, DF_GPW = (DF_GROUPS + 3) / 4;
}  // namespace

__global__ __launch_bounds__(256) void gennet_dec_final_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ wt, const float* __restrict__ bias,
                                                               float slope, const float* __restrict__ w1, float bias1, __bf16* __restrict__ y,
                                                               int B, int H, int W) {
    __shared__ __attribute__((aligned(16))) float Tl[DF_TW * DF_TH * 12];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pl = lane & 15, g = lane >> 4;
    const int Ho = 2 * H, Wo = 2 * W;
    const int tiles_x = (Wo + DF_W - 1) / DF_W, tiles_y = (Ho + DF_H - 1) / DF_H;
    const int total_tiles = B * tiles_x * tiles_y;
    bf16x8 wa[4][2][3];
#pragma unroll
    for (int cl = 0; cl < 4; ++cl)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) wa[cl][nt][ks] = ld8(wt + ((cl * 32) + row_channel(nt, pl)) * 96 + ks * 32 + g * 8);
    const float4 b0 = *reinterpret_cast<const float4*>(bias + 8 * g), b1 = *reinterpret_cast<const float4*>(bias + 8 * g + 4);
    // final convolution: row = tap (lane & 15), k = channel 8g .. 8g+7; w1 is [24][9] float32 (ci * 9 + tap)
    bf16x8 ahi, alo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float wv = (pl < 9 && g < 3) ? w1[(8 * g + e) * 9 + pl] : 0.0f;
        const __bf16 h = (__bf16)wv;
        ahi[e] = h; alo[e] = (__bf16)(wv - (float)h);
    }
    // persistent: a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... so the 26 weight fragments above are loaded once
    // per wave, not once per 512 outputs (they are 11x the bytes of a tile's own input)
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int b = tile / (tiles_x * tiles_y), tt = tile - b * tiles_x * tiles_y;
    const int i0 = (tt / tiles_x) * DF_H, j0 = (tt % tiles_x) * DF_W;
    const __bf16* xb = x + (size_t)b * H * W * GC;
    bf16x8 fb[DF_GPW][3];
#pragma unroll
    for (int q = 0; q < DF_GPW; ++q) {
        const int e = min((wave * DF_GPW + q) * 16 + pl, DF_POS - 1);
        const int py = e / DF_PW, px = e - py * DF_PW;
        const int iy = (i0 >> 1) - 1 + py, ix = (j0 >> 1) - 1 + px;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int c = ks * 4 + g;                            // chunk: input pixel c / 3 of the 2x2 block, channels (c % 3) * 8 ..
            const int p = c / 3, ci0 = (c - p * 3) * 8;
            const int yy = iy + (p >> 1), xx = ix + (p & 1);
            const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
            fb[q][ks] = in ? ld8(xb + (uint32_t)((yy * W + xx) * GC + ci0)) : zero8();
        }
    }
#pragma unroll
    for (int q = 0; q < DF_GPW; ++q) {
        const int grp = wave * DF_GPW + q;
        const int e = min(grp * 16 + pl, DF_POS - 1);
        const int py = e / DF_PW, px = e - py * DF_PW;
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cl][0][ks], fb[q][ks], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cl][1][ks], fb[q][ks], a1, 0, 0, 0);
            }
            // this lane's 8 channels of output pixel (ly, lx) of the T region, as the tensor the unfused path stores
            const int ly = 2 * py + (cl >> 1), lx = 2 * px + (cl & 1);
            const int oy = i0 - 2 + ly, ox = j0 - 2 + lx;
            const bool inside = oy >= 0 && oy < Ho && ox >= 0 && ox < Wo;
            float o[8] = {a0[0] + b0.x, a0[1] + b0.y, a0[2] + b0.z, a0[3] + b0.w, a1[0] + b1.x, a1[1] + b1.y, a1[2] + b1.z, a1[3] + b1.w};
            bf16x8 xv;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float v = o[r] > 0.f ? o[r] : o[r] * slope;
                xv[r] = (inside && g < 3) ? (__bf16)v : (__bf16)0.0f;
            }
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi, xv, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo, xv, t, 0, 0, 0);
            if (grp < DF_GROUPS && g < 3) *reinterpret_cast<float4*>(Tl + (ly * DF_TW + lx) * 12 + 4 * g) = make_float4(t[0], t[1], t[2], t[3]);
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty0 = threadIdx.x >> 5;               // two output rows per thread: ty0 and ty0 + 8
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int ty = ty0 + 8 * r;
        const int i = i0 + ty, jx = j0 + tx;
        float acc = bias1;
#pragma unroll
        for (int di = 0; di < 3; ++di)
#pragma unroll
            for (int dj = 0; dj < 3; ++dj) acc += Tl[((ty + 1 + di) * DF_TW + tx + 1 + dj) * 12 + di * 3 + dj];
        if (i < Ho && jx < Wo) y[((size_t)b * Ho + i) * Wo + jx] = (__bf16)acc;
    }
    __syncthreads();                                                        // T is rewritten by the next tile
    }
}

// GenNet's first convolution (1 -> 24, 3x3, stride 1, + BN + LeakyReLU, ae_vit.py:24-28) FUSED into the first encoder stage
// (24 -> 24, stride 2): the 24-channel full-resolution tensor between them (805 MB at batch 256, 256 x 256) is never written.
// Per 16 output pixels and per encoder tap t (9 of them), the first convolution's output at the tap's position is itself a
// matrix product — [24 channels] x [9 input taps + 1], weights split hi + lo bfloat16 into one K = 32 step (slots 0..8 hi, 16..24
// lo; the input values sit in both halves; slots 9 / 25 carry the bias against a constant 1) — whose accumulator layout (lane = pixel, 4 consecutive channels per register group of
// each of the two channel tiles) is taken AS the encoder product's B operand: the encoder's k order is free, so tap t is one
// k-step whose slot (g, e) is channel 4g + e (e < 4) or 16 + 4g + e - 4 (e >= 4, g < 2), and the weights are packed to match.
// 2 + 2 MFMAs per tap, LeakyReLU + zero-padding mask on 8 values per lane in between (slope1 must be <= 1); every lane keeps the 5 x 5 input
// patch of its pixel (15 dword loads) and cuts the nine 3 x 3 windows out of it.
//   x1: [B][H][W] bfloat16;  w1: [2 tiles][16 rows][32] bfloat16 (hi | lo as above incl. the bias columns, rows = channel);
//   wk2: [2 tiles][9 taps][16 rows co][32 k-slots] bfloat16;  bias2: [32] float32;  y: [B][H/2][W/2][24] bfloat16.
__global__ __launch_bounds__(256) void gennet_first_enc_kernel(const __bf16* __restrict__ x1, const __bf16* __restrict__ w1, const float* __restrict__ /*b1: inside w1*/,
                                                               const __bf16* __restrict__ wk2, const float* __restrict__ bias2, __bf16* __restrict__ y,
                                                               int B, int H, int W, float slope1, float slope2) {
    const int lane = threadIdx.x & 63, pl = lane & 15, g = lane >> 4;
    const int Ho = H >> 1, Wo = W >> 1;
    const long long total = (long long)B * Ho * Wo;
    bf16x8 wa1[2], wa2[2][9];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        wa1[nt] = ld8(w1 + (nt * 16 + pl) * 32 + g * 8);
#pragma unroll
        for (int t = 0; t < 9; ++t) wa2[nt][t] = ld8(wk2 + ((nt * 9 + t) * 16 + pl) * 32 + g * 8);
    }
    const float4 d0 = *reinterpret_cast<const float4*>(bias2 + 4 * g), d1 = *reinterpret_cast<const float4*>(bias2 + 16 + 4 * g);
    // 32-bit index arithmetic throughout (the launcher checks B * H * W < 2^31): 64-bit divisions cost hundreds of instructions per iteration
    const uint32_t total32 = (uint32_t)total, wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const uint32_t per_img = (uint32_t)Wo * Ho;
    for (uint32_t base = wave * 16; base < total32; base += nwaves * 16) {
        uint32_t pix = base + pl;
        const bool live = pix < total32;
        if (!live) pix = total32 - 1;
        const uint32_t img = pix / per_img, rem = pix - img * per_img;
        const int oy = (int)(rem / (uint32_t)Wo), ox = (int)(rem - (uint32_t)oy * Wo);
        const __bf16* xb = x1 + (size_t)img * H * W;                         // wave-uniform only when a group does not straddle images: keep per lane
        // the 5 x 5 input patch around (2 oy, 2 ox): rows 2oy-2 .. 2oy+2, columns 2ox-2 .. 2ox+3 as three dwords per row (W is even
        // and the first column is even: a dword is wholly inside or wholly outside the image)
        uint32_t patch[5][3];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int iy = 2 * oy - 2 + r;
            const int cy = min(max(iy, 0), H - 1);
            const bool rin = iy >= 0 && iy < H;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int ix = 2 * ox - 2 + 2 * k;
                const bool in = rin && ix >= 0 && ix < W;
                // unconditional load from a clamped address, then the select (a load inside a divergent branch would serialise)
                const int cx = min(max(ix, 0), W - 2);
                const uint32_t d = *reinterpret_cast<const uint32_t*>(xb + (uint32_t)(cy * W + cx));
                patch[r][k] = in ? d : 0u;
            }
        }
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        const bool odd = g & 1;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ty = t / 3, tx = t % 3;
            // window of the position (2oy + ty - 1, 2ox + tx - 1): patch rows ty .. ty+2, patch columns tx .. tx+2.  Its nine 16-bit
            // values are paired straight into operand dwords: element i of the window is patch value (ty + i / 3, tx + i % 3), and a
            // pair of values is one byte-permute of (at most) two patch dwords.
            auto val2 = [&](int i0, int i1) -> uint32_t {                  // dword holding window elements i0 (low half) and i1 (high half)
                const int r0 = ty + i0 / 3, c0_ = tx + i0 % 3, r1 = ty + i1 / 3, c1_ = tx + i1 % 3;
                const uint32_t da = patch[r0][c0_ >> 1], db = patch[r1][c1_ >> 1];
                // v_perm_b32(hi_src, lo_src, sel): byte k of the result = byte sel[k] of {hi_src : lo_src} (bytes 4..7 : 0..3)
                const uint32_t sel = ((c0_ & 1) ? 0x0302u : 0x0100u) | (((c1_ & 1) ? 0x0706u : 0x0504u) << 16);
                return __builtin_amdgcn_perm(db, da, sel);
            };
            // even lane quarters hold input taps 0..7, odd quarters tap 8 and the constant 1 that carries the bias column
            const uint32_t e0 = val2(0, 1), e1 = val2(2, 3), e2 = val2(4, 5), e3 = val2(6, 7);
            const uint32_t w8 = (patch[ty + 2][(tx + 2) >> 1] >> (((tx + 2) & 1) * 16)) & 0xffffu;
            typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_;
            const u32x4_ fw = {odd ? (w8 | 0x3F800000u) : e0, odd ? 0u : e1, odd ? 0u : e2, odd ? 0u : e3};
            const bf16x8 bw = __builtin_bit_cast(bf16x8, fw);
            f32x4 y0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa1[0], bw, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            f32x4 y1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa1[1], bw, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            // LeakyReLU (slope < 1: max(v, v * slope)); positions outside the image are the encoder's zero padding
            const int py = 2 * oy + ty - 1, px = 2 * ox + tx - 1;
            const bool inside = py >= 0 && py < H && px >= 0 && px < W;
            float v[8] = {y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * slope1);
            u32x4_ fv = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            if (!inside) fv = u32x4_{0u, 0u, 0u, 0u};
            const bf16x8 fb = __builtin_bit_cast(bf16x8, fv);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa2[0][t], fb, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa2[1][t], fb, a1, 0, 0, 0);
        }
        if (live) {
            __bf16* dst = y + (size_t)pix * GC;
            store4(dst + 4 * g, a0, d0, slope2);                             // co = 4g .. 4g+3
            if (g < 2) store4(dst + 16 + 4 * g, a1, d1, slope2);             // co = 16 + 4g ..; 24..31 are padding
        }
    }
}

int gennet_first_enc_launch(const void* x1, const void* w1, const float* b1, const void* wk2, const float* bias2, void* y, int B, int H, int W, float slope1,
                            float slope2, hipStream_t stream) {
    const long long total = (long long)B * (H / 2) * (W / 2);
    if ((long long)B * H * W >= (1LL << 31)) return (int)hipErrorInvalidValue;
    long long blocks = (total + 63) / 64;
    if (blocks > 256 * 12) blocks = 256 * 12;                    // grid-stride: the register-resident weights are loaded once per wave
    hipLaunchKernelGGL(gennet_first_enc_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const __bf16*)x1, (const __bf16*)w1, b1,
                       (const __bf16*)wk2, bias2, (__bf16*)y, B, H, W, slope1, slope2);
    return (int)hipGetLastError();
}

int gennet_enc_conv_launch(const void* x, const void* wk, const float* bias, void* y, int B, int H, int W, float slope, hipStream_t stream) {
    const long long total = (long long)B * (H / 2) * (W / 2);
    long long blocks = (total + 16 * GROUPS * 4 - 1) / (16 * GROUPS * 4);
    if (blocks > 256 * 16) blocks = 256 * 16;                    // grid-stride: the register-resident weights are loaded once per wave
    hipLaunchKernelGGL(gennet_enc_conv_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const __bf16*)x, (const __bf16*)wk, bias,
                       (__bf16*)y, B, H, W, slope);
    return (int)hipGetLastError();
}

int gennet_dec_final_launch(const void* x, const void* wt, const float* bias, float slope, const float* w1, float bias1, void* y, int B, int H, int W,
                            hipStream_t stream) {
    const long long tiles = (long long)B * ((2 * H + DF_H - 1) / DF_H) * ((2 * W + DF_W - 1) / DF_W);
    if (tiles >= (1LL << 31)) return (int)hipErrorInvalidValue;
    int cus = device_cu_count();
    if (!cus) cus = 256;
    const long long grid = tiles < 2LL * cus ? tiles : 2LL * cus;          // 2 workgroups per CU (212 VGPRs: two waves per SIMD)
    hipLaunchKernelGGL(gennet_dec_final_kernel, dim3((unsigned)grid), dim3(256), 0, stream, (const __bf16*)x, (const __bf16*)wt, bias, slope, w1, bias1,
                       (__bf16*)y, B, H, W);
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
