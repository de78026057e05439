// gemm_small.hip — the dense projection c = epilogue(a w^T) (ppn_gemm_bf16's contract; SegNet/nat.py:62-85,111-120) for FEW ROWS: a
// batch of 1-16 problems gives the NAT projections M = 49 .. 4 096 tokens, i.e. a handful of the 256 x 256 tiles the large-batch
// kernels are built around — a single workgroup would walk K = 4 096 alone.  Here a workgroup owns a 32-row x 64-column block of c,
// its four waves split K (64-wide slices round-robin; fragments straight from global memory — the operands of such a product live
// in L2 — and one LDS reduction at the end, in wave order), and the grid is (M / 32) x (N / 64) workgroups: every CU has work at
// batch 1 and the longest dependent chain is K / 256 trips.  The product is computed transposed (D^T = W A^T) so that a lane ends up with 4 consecutive columns of one row of c:
// 8-byte stores, 16-byte bias reads.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "ppn_device.h"
#include "ppn_kernels.h"

namespace ppn {

namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int SM = 32, SN = 64;        // a wave's block of c
}  // namespace

template <int EPI>
__global__ __launch_bounds__(256) void gemm_small_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ w, const float* __restrict__ bias,
                                                         __bf16* __restrict__ c, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float part[3][8][64][4];       // the partial blocks of waves 1 .. 3: [tile][lane] f32x4
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m0 = blockIdx.x * SM, n0 = blockIdx.y * SN;
    const int i = lane & 15, g = lane >> 4;
    // fragment rows: w rows n0 + 16 nt + i (A operand), a rows m0 + 16 mt + i (B operand; rows past M repeat the last one, never stored)
    const __bf16* wp[4];
    const __bf16* ap[2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wp[nt] = w + (size_t)(n0 + nt * 16 + i) * K + 8 * g;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) ap[mt] = a + (size_t)min(m0 + mt * 16 + i, M - 1) * K + 8 * g;
    f32x4 acc[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the four waves split K: wave s takes the 64-wide slices s, s + 4, ... (two k-steps per trip, their 12 loads in flight together)
    for (int k0 = wave * 64; k0 < K; k0 += 256) {
        bf16x8 wf[2][4], af[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[s][nt] = *reinterpret_cast<const bf16x8*>(wp[nt] + k0 + 32 * s);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[s][mt] = *reinterpret_cast<const bf16x8*>(ap[mt] + k0 + 32 * s);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][nt], af[s][mt], acc[nt][mt], 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) *reinterpret_cast<f32x4*>(part[wave - 1][nt * 2 + mt][lane]) = acc[nt][mt];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int s = 0; s < 3; ++s) acc[nt][mt] += *reinterpret_cast<const f32x4*>(part[s][nt * 2 + mt][lane]);     // in wave order: reproducible
    // D^T tile (nt, mt): lane (i, g) holds columns n0 + 16 nt + 4 g + 0..3 of row m0 + 16 mt + i
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = m0 + mt * 16 + i;
        if (row >= M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int col = n0 + nt * 16 + 4 * g;
            __bf16* dst = c + (size_t)row * N + col;
            f32x4 v = acc[nt][mt];
            if (EPI == 2) {
                const uint2 o = *reinterpret_cast<const uint2*>(dst);
                v[0] += __uint_as_float(o.x << 16); v[1] += __uint_as_float(o.x & 0xffff0000u);
                v[2] += __uint_as_float(o.y << 16); v[3] += __uint_as_float(o.y & 0xffff0000u);
            } else {
                const float4 b = *reinterpret_cast<const float4*>(bias + col);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                if (EPI == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752f));
                }
                if (EPI == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
            }
            *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
    }
}

// the shapes this kernel takes from ppn_gemm_bf16: fewer than 64 of the large kernel's tiles, whole 64-column blocks
bool gemm_small_wanted(long long M, int N, int K) {
    static const int tiles = getenv("PPNET_SMALL_GEMM_TILES") ? atoi(getenv("PPNET_SMALL_GEMM_TILES")) : 64;     // A/B: where the large kernels take over
    return (N % SN) == 0 && (K % 64) == 0 && ((M + 255) / 256) * ((N + 255) / 256) < tiles;
}

int gemm_small_launch(const void* a, const void* w, const float* bias, void* c, long long M, int N, int K, int epilogue, hipStream_t stream) {
    const dim3 grid((unsigned)((M + SM - 1) / SM), (unsigned)(N / SN));
    const __bf16* A = (const __bf16*)a; const __bf16* W = (const __bf16*)w; __bf16* C = (__bf16*)c;
    switch (epilogue) {
        case 0: hipLaunchKernelGGL(gemm_small_kernel<0>, grid, dim3(256), 0, stream, A, W, bias, C, (int)M, N, K); break;
        case 1: hipLaunchKernelGGL(gemm_small_kernel<1>, grid, dim3(256), 0, stream, A, W, bias, C, (int)M, N, K); break;
        case 2: hipLaunchKernelGGL(gemm_small_kernel<2>, grid, dim3(256), 0, stream, A, W, bias, C, (int)M, N, K); break;
        default: hipLaunchKernelGGL(gemm_small_kernel<3>, grid, dim3(256), 0, stream, A, W, bias, C, (int)M, N, K); break;
    }
    return (int)hipGetLastError();
}

}  // namespace ppn
