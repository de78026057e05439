// nat_gemm128.hip — the NAT projections of ppn_nat_gemm_bf16 (same three modes, same algebra: nat_gemm.hip's header) for the
// HBM-BOUND levels, C = 256 and 512 (DiNAT-B / NAT-Base levels 1 and 2: 22 of the 27 layers), as a small-tile kernel that hides
// memory latency with OCCUPANCY instead of one deep pipeline.
//
// Why a second kernel: with K = C <= 512 (2 048 for fc2) a projection moves 130 - 400 MB for 26 - 100 GFLOP: it is bound by HBM, and
// what sets its speed is how many bytes a CU keeps in flight.  nat_gemm.hip's persistent 256 x 256 tile owns a CU with ONE stream of
// k-tiles, 1.25 - 1.5 of them in flight (48 KB of A): 3.0 - 3.6 TB/s, where the vendor's kernels reach 4.5 - 5.7 on the same shapes.
// Here a workgroup is 4 waves on a 128 x 128 tile with a 3-slot ring of 32-wide k-steps (48 KB of LDS): three workgroups share a CU,
// each with two k-steps (32 KB) in flight and at a different point of its tile — loads, MFMAs, the epilogue's old-C reads and the
// stores of different tiles overlap without any cross-tile pipelining in the code.  The old C of the accumulating mode is simply
// read in the epilogue (no identity k-tiles: 1/3 less MFMA and LDS work at K = 512), the row statistics leave as one partial per
// 128 columns.
//
// Core: D^T = W A^T with v_mfma_f32_16x16x32_bf16; a wave owns 64 x 64 of the tile; the W rows of its four MFMA row-tiles are
// interleaved (MFMA row i of tile nt = column (i / 4) * 16 + nt * 4 + i % 4) so that a lane ends up with 16 CONSECUTIVE columns of
// each of its 4 rows: 16-byte stores, 16-byte bias / colsum reads.  Both operands go global -> LDS by LDS-DMA (asm: the compiler
// does not see them and cannot put a full drain in front of an LDS read), rows of 64 bytes with the 32-byte halves swapped on
// every other group of 4 rows (conflict-free 16-byte fragment reads; the swap is applied on the source address).
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include "ppn_device.h"
#include "ppn_kernels.h"

namespace ppn {
namespace ng128 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int TM = 128, TN = 128, TK = 32, NTHR = 256;
constexpr int OP_BYTES = 128 * 64;                 // one operand of one k-step: 128 rows x 64 bytes
constexpr int STAGE = 2 * OP_BYTES;                // A | W
constexpr int lds_bytes(int slots) { return slots * STAGE + 2 * 128 * 8; }   // the ring | ACC: float2 red[2 wave columns][128 rows]

struct Params {
    const __bf16* A; const __bf16* W; __bf16* C;
    const float* bias; const float* colsum; const float* stats_in; float* stats_out;
    int M, N, K, P_in;
    float inv_k, eps;
};

__device__ __forceinline__ void dma16(const void* base_uniform, unsigned off, unsigned lds_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base_uniform), "s"(lds_uniform) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

__device__ __forceinline__ float gelu_erf(float x) {                      // x * sigmoid(x (p0 + p1 x^2 + p2 x^4)): nat_gemm.hip's fit, |err| < 3e-5
    const float x2 = fminf(x * x, 64.0f);
    const float t = x * fmaf(x2, fmaf(x2, -1.0350827e-3f, 0.10690469f), 2.3009787f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-t));
}

// MODE 0: C = rstd (A W'^T - mean colsum) + bias';  1: gelu of that;  2: C += A W^T + bias in place, row partials of the new C out
template <int MODE, int SLOTS>
__global__ __launch_bounds__(NTHR, SLOTS == 2 ? 4 : 3) void nat_gemm128_kernel(const Params p) {
    constexpr int RED_OFF = SLOTS * STAGE;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;                               // this wave's 64 x 64 quadrant of the tile
    // tile order: XCD x (workgroups x, x + 8, ...) walks a contiguous run of tiles, the column tiles of a row block side by side —
    // the workgroups of an XCD share the A rows they read and the whole (small) W through its L2
    const int tiles_n = p.N / TN;
    int t = blockIdx.x;
    {
        const int n = gridDim.x, q = n >> 3, r = n & 7, x = t & 7;
        t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (t >> 3);
    }
    const int m0 = (t / tiles_n) * TM, n0 = (t % tiles_n) * TN;

    // ---- staging: a wave-instruction fills 16 rows x 64 bytes; lane l -> row l / 4, position l % 4 holding chunk (l % 4) ^ swap(row)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int srow = lane >> 2;
    const unsigned schunk = (unsigned)((lane & 3) ^ (((srow >> 2) & 1) << 1)) * 16;
    const unsigned soff = (unsigned)srow * (unsigned)p.K * 2u + schunk;    // from the instruction's first row, bytes
    const char* Ab = reinterpret_cast<const char*>(p.A) + (size_t)m0 * p.K * 2;
    const char* Wb = reinterpret_cast<const char*>(p.W) + (size_t)n0 * p.K * 2;
    auto stage = [&](int ks, int slot) __attribute__((always_inline)) {   // k-step ks -> ring slot: 2 + 2 instructions per wave
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int blk = wave * 2 + j;                                  // 16-row block of the operand
            dma16(Ab + ((size_t)blk * 16 * p.K + ks * TK) * 2, soff, lds0 + slot * STAGE + blk * 1024);
            dma16(Wb + ((size_t)blk * 16 * p.K + ks * TK) * 2, soff, lds0 + slot * STAGE + OP_BYTES + blk * 1024);
        }
    };
    // ---- fragment reads.  MFMA row i of W tile nt is tile column (i >> 2) * 16 + nt * 4 + (i & 3) of the wave's 64; A tile mt is rows
    // 16 mt + i.  Lane (i, g) reads chunk g of its row, stored at position g ^ swap(row)
    const int i = lane & 15, g = lane >> 4;
    int a_rd[4], w_rd[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ar = wr * 64 + q * 16 + i;
        a_rd[q] = ar * 64 + ((g ^ (((ar >> 2) & 1) << 1)) * 16);
        const int wrow = wc * 64 + (i >> 2) * 16 + q * 4 + (i & 3);
        w_rd[q] = OP_BYTES + wrow * 64 + ((g ^ (((wrow >> 2) & 1) << 1)) * 16);
    }
    f32x4 acc[4][4];                                                       // [W tile nt][A tile mt]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // MODE 2: the tile's old C values are requested FIRST, before any copy of the ring — the vector-memory counter retires in order,
    // so the first k-step's wait (which leaves only younger copies in flight) also covers them, every later wait sees the same queue
    // as without them, and they have the whole k-loop to land instead of a full memory latency in front of the epilogue of every
    // tile (32 registers; the kernel was at 2.9 TB/s on fc2 of level 2 where the vendor's reaches 4.3)
    const int col0 = n0 + wc * 64 + g * 16;
    uint4 oldc[4][2];
    if (MODE == 2) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const __bf16* src = p.C + (size_t)(m0 + wr * 64 + mt * 16 + i) * p.N + col0;
            oldc[mt][0] = *reinterpret_cast<const uint4*>(src);
            oldc[mt][1] = *reinterpret_cast<const uint4*>(src + 8);
        }
    }
    const int nk = p.K / TK;                                               // >= 2 (host)
#pragma unroll
    for (int q = 0; q < SLOTS - 1; ++q)
        if (q < nk) stage(q, q);
    for (int ks = 0; ks < nk; ++ks) {
        // k-step ks has landed (this wave's copies: all but the 4 of each later step in flight; behind the barrier everyone's), and
        // every wave has left step ks - 1, whose slot the copies of step ks + SLOTS - 1 fill
        const int ahead = min(nk - 1 - ks, SLOTS - 2);                     // steps in flight behind ks (wave-uniform)
        if (ahead >= 2) wait_vm<8>(); else if (ahead == 1) wait_vm<4>(); else wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (ks + SLOTS - 1 < nk) stage(ks + SLOTS - 1, (ks + SLOTS - 1) % SLOTS);
        const unsigned char* st = lds + (ks % SLOTS) * STAGE;
        bf16x8 af[4], wf[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            af[q] = *reinterpret_cast<const bf16x8*>(st + a_rd[q]);
            wf[q] = *reinterpret_cast<const bf16x8*>(st + w_rd[q]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
    }

    // ---- epilogue.  acc[nt][mt][r]: row m0 + wr*64 + mt*16 + i, column n0 + wc*64 + g*16 + nt*4 + r: 16 consecutive columns per lane and row
    float bv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + col0 + 4 * q);
        bv[4 * q] = b4.x; bv[4 * q + 1] = b4.y; bv[4 * q + 2] = b4.z; bv[4 * q + 3] = b4.w;
    }
    float cs[16];
    if (MODE != 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 c4 = *reinterpret_cast<const float4*>(p.colsum + col0 + 4 * q);
            cs[4 * q] = c4.x; cs[4 * q + 1] = c4.y; cs[4 * q + 2] = c4.z; cs[4 * q + 3] = c4.w;
        }
    }
    float2* red = reinterpret_cast<float2*>(lds + RED_OFF);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int row = m0 + wr * 64 + mt * 16 + i;
        __bf16* dst = p.C + (size_t)row * p.N + col0;
        float v[16];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[nt * 4 + r] = acc[nt][mt][r];
        if (MODE != 2) {
            // mean / rstd of the row from the partial sums the accumulating kernel (or row_stats) left, in partial order
            float s = 0.f, q2 = 0.f;
            for (int pi = 0; pi < p.P_in; ++pi) {
                const float2 st2 = *reinterpret_cast<const float2*>(p.stats_in + ((size_t)pi * p.M + row) * 2);
                s += st2.x; q2 += st2.y;
            }
            const float mean = s * p.inv_k;
            const float rstd = rsqrtf(fmaxf(q2 * p.inv_k - mean * mean, 0.f) + p.eps);
            const float nmr = -mean * rstd;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float y = fmaf(rstd, v[c], fmaf(nmr, cs[c], bv[c]));
                if (MODE == 1) y = gelu_erf(y);
                v[c] = y;
            }
        } else {
            const uint4 o0 = oldc[mt][0], o1 = oldc[mt][1];
            const uint32_t ow[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                v[2 * c] += __uint_as_float(ow[c] << 16) + bv[2 * c];
                v[2 * c + 1] += __uint_as_float(ow[c] & 0xffff0000u) + bv[2 * c + 1];
            }
        }
        uint32_t pk[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) pk[c] = pack_bf16x2(v[2 * c], v[2 * c + 1]);
        *reinterpret_cast<uint4*>(dst) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        *reinterpret_cast<uint4*>(dst + 8) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        if (MODE == 2) {
            // (sum, sum of squares) of the bf16 values just stored: over this lane's 16, the row's 4 lanes, then the two wave columns
            float s = 0.f, q2 = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float f0 = __uint_as_float(pk[c] << 16), f1 = __uint_as_float(pk[c] & 0xffff0000u);
                s += f0 + f1; q2 += f0 * f0 + f1 * f1;
            }
            s += __shfl_xor(s, 16, 64); q2 += __shfl_xor(q2, 16, 64);
            s += __shfl_xor(s, 32, 64); q2 += __shfl_xor(q2, 32, 64);
            if (g == 0) red[wc * 128 + wr * 64 + mt * 16 + i] = make_float2(s, q2);
        }
    }
    if (MODE == 2) {
        __syncthreads();
        if (threadIdx.x < 128) {
            const float2 x0 = red[threadIdx.x], x1 = red[128 + threadIdx.x];
            *reinterpret_cast<float2*>(p.stats_out + ((size_t)(n0 / TN) * p.M + m0 + threadIdx.x) * 2) = make_float2(x0.x + x1.x, x0.y + x1.y);
        }
    }
}

template <int MODE, int SLOTS>
static int launch_s(const Params& p, hipStream_t stream) {
    static DeviceOnce attr;
    if (const int e = dynamic_lds_once(attr, (const void*)nat_gemm128_kernel<MODE, SLOTS>, lds_bytes(SLOTS))) return e;
    const int tiles = (p.M / TM) * (p.N / TN);
    hipLaunchKernelGGL((nat_gemm128_kernel<MODE, SLOTS>), dim3(tiles), dim3(NTHR), lds_bytes(SLOTS), stream, p);
    return (int)hipGetLastError();
}
template <int MODE>
static int launch(const Params& p, hipStream_t stream) {
    static const int slots = getenv("PPNET_NG128_SLOTS") ? atoi(getenv("PPNET_NG128_SLOTS")) : 3;      // A/B: ring depth (2: four workgroups per CU)
    if (slots == 2) return launch_s<MODE, 2>(p, stream);
    if (slots == 4) return launch_s<MODE, 4>(p, stream);
    return launch_s<MODE, 3>(p, stream);
}

}  // namespace ng128

// What this kernel takes from ppn_nat_gemm_bf16.  Residual streams of width C <= 512 keep their row statistics as one partial per 128
// columns (nat_gemm128_partials: producer and consumers agree by C alone); their ACCUMULATING projections (N = C: proj, fc2) run here —
// 4.3 / 3.8 TB/s at C = 256 / 512 against the persistent kernel's 3.6 / 3.0.  The LayerNorm modes (N = 2C, 3C: more stores and
// more MFMA work per byte read) are faster on the persistent kernel, which reads up to 4 partials per row; PPNET_NAT_GEMM128=all
// sends them here too, =0 nothing (A/B).
static int knob() {
    static const int k = [] { const char* e = getenv("PPNET_NAT_GEMM128"); return !e ? 1 : (e[0] == '0' ? 0 : (e[0] == 'a' ? 2 : 1)); }();
    return k;
}
// (round 5: widths above 256 keep one partial per 256 columns — their producers are the 256 x 256 core's accumulating epilogue now, and the
// LayerNorm-folded consumers read half as many partials per row; width 256 stays at two because the fused MLP kernel emits two)
bool nat_gemm128_partials(int C) { return knob() != 0 && C <= 256 && (C % 128) == 0; }
bool nat_gemm128_wanted(int N, int K, int mode) {
    const int C = mode == 2 ? N : K;
    if (!nat_gemm128_partials(C) || (N % 128) != 0 || (K % 32) != 0 || K < 64) return false;
    return mode == 2 || knob() == 2;
}

int nat_gemm128_launch(const void* a, const void* w, const float* bias, const float* colsum, const float* stats_in, int p_in, float* stats_out,
                       void* c, long long M, int N, int K, int mode, float eps, hipStream_t stream) {
    ng128::Params p{};
    p.A = (const __bf16*)a; p.W = (const __bf16*)w; p.C = (__bf16*)c; p.bias = bias; p.colsum = colsum; p.stats_in = stats_in;
    p.stats_out = stats_out; p.M = (int)M; p.N = N; p.K = K; p.P_in = p_in; p.inv_k = 1.0f / (float)K; p.eps = eps;
    switch (mode) {
        case 0: return ng128::launch<0>(p, stream);
        case 1: return ng128::launch<1>(p, stream);
        case 2: return ng128::launch<2>(p, stream);
        default: return -1;
    }
}

}  // namespace ppn
