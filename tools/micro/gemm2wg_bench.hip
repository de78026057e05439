// gemm2wg_bench.hip — EXPERIMENT (round 5): a bf16 MFMA GEMM built from TWO independent 4-wave workgroups per CU instead of one
// 8-wave workgroup, timed against the shipped core (ppnet_amd/csrc/mfma_gemm.h) in the same process.
//
// Why: at K = 512 a tile round of the shipped persistent kernel is a 13 us k-loop + 5.6 - 13 us of epilogue with the matrix pipe idle
// (DESIGN.md section 8).  Inside one workgroup there is no room to overlap the two; two workgroups per CU overlap them by
// construction — one's epilogue, barrier and LDS waits run under the other's MFMAs.
//
//   C[m][n] = bf16( sum_k A[m][k] * W[n][k] + bias[n] )          A [M][K], W [N][K] bf16 row-major, M % 128 == 0, N % 256 == 0, K % 32 == 0
//
// Workgroup: 256 threads, tile 128 x 256, wave w owns all 128 rows x columns 64 w .. 64 w + 63 (128 accumulator registers: the
// shipped core's wave tile, so the same LDS fragment traffic per MFMA).  k-stages of 32 (one MFMA k-step): A 8 KB + W 16 KB per
// stage, a ring of 3 stages = 72 KB, so two workgroups share a CU (144 KB of LDS, 2 x 256 registers per SIMD lane).  Per stage:
//     counted wait: this wave's six 1-KB pieces of the stage have landed (LDS-DMA, issued two stages ago)
//     s_barrier                       (every wave's pieces have landed; every wave has finished reading the stage before)
//     LDS-DMA of the stage after next into the slot the previous stage used
//     12 fragment reads (8 A tiles, 4 W tiles: ds_read_b128, conflict-free through a 4-position swizzle of a row's 16-byte chunks)
//     32 MFMAs (D^T = W . A^T: a lane holds 4 consecutive columns of one row)
// The ring runs across tiles (persistent workgroups, grid = 2 x CUs); the epilogue (bias, bf16, 8-byte stores) leaves its stores in flight.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/gemm2wg_bench.hip -o tools/micro/gemm2wg_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../ppnet_amd/csrc/mfma_gemm.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

namespace g2 {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int BM = 128, BN = 256, BK = 32, NT = 256, NSTAGE = 3;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES, LDS_BYTES = NSTAGE * STAGE;

struct Params { const __bf16* A; const __bf16* W; __bf16* C; const float* bias; int M, N, K; int delay; };

__device__ __forceinline__ void dma16(const void* base_uniform, unsigned off, unsigned lds_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base_uniform), "s"(lds_uniform) : "memory");
}

__global__ __launch_bounds__(NT, 2) void gemm_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // chunk g of a 64-byte row r sits at position g ^ SW[(r >> 2) & 3], SW = {0, 3, 2, 1}: the 16 lanes ds_read_b128 serves in one
    // cycle — (g, i >> 2) in {(0,0), (0,3), (1,1), (1,2)} or {(0,1), (0,2), (1,0), (1,3)} per row residue i & 3 — hit 16 distinct slots
    auto sw = [](int q) { return (4 - q) & 3; };
    // staging: piece = 16 rows; this lane copies row (lane >> 2) of a piece into position lane & 3, i.e. it FETCHES chunk pos ^ SW
    const int prow = lane >> 2, pchunk = (lane & 3) ^ sw((prow >> 2) & 3);
    unsigned offA[2], offB[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) offA[k] = (unsigned)((wave + 4 * k) * 16 + prow) * (unsigned)p.K * 2u + pchunk * 16u;      // pieces w, w + 4 of A's 8
#pragma unroll
    for (int k = 0; k < 4; ++k) offB[k] = (unsigned)((wave + 4 * k) * 16 + prow) * (unsigned)p.K * 2u + pchunk * 16u;      // pieces w + 4k of W's 16
    // fragment reads: row (tile * 16 + i), chunk g
    const unsigned fpos = (unsigned)(i * 64 + ((g ^ sw((i >> 2) & 3)) * 16));
    const unsigned fa0 = fpos, fb0 = (unsigned)A_BYTES + (unsigned)(wave * 64) * 64u + fpos;

    const int tiles_m = p.M / BM, tiles = tiles_m * (p.N / BN), nk = p.K / BK;
    int my = 0;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) ++my;
    const int total = my * nk;                                             // stages of this workgroup
    if (total == 0) return;
    // staging cursor, two stages ahead of the compute cursor
    int s_tile = blockIdx.x, s_k = 0, s_n = 0;
    const char* st_ab = nullptr; const char* st_bb = nullptr; unsigned st_slot = 0;
    auto stage_begin = [&]() __attribute__((always_inline)) {
        const int tm = s_tile % tiles_m, tn = s_tile / tiles_m;
        st_ab = reinterpret_cast<const char*>(p.A) + ((size_t)tm * BM * p.K + (size_t)s_k * BK) * 2;
        st_bb = reinterpret_cast<const char*>(p.W) + ((size_t)tn * BN * p.K + (size_t)s_k * BK) * 2;
        st_slot = lds0 + (unsigned)(s_n % NSTAGE) * STAGE;
        ++s_n;
        if (++s_k == nk) { s_k = 0; s_tile += gridDim.x; }
    };
    auto stage_piece = [&](int k) __attribute__((always_inline)) {          // k = 0 .. 5: this wave's two A pieces, then its four W pieces
        if (k < 2) dma16(st_ab, offA[k], st_slot + (unsigned)(wave + 4 * k) * 1024u);
        else dma16(st_bb, offB[k - 2], st_slot + A_BYTES + (unsigned)(wave + 4 * (k - 2)) * 1024u);
    };
    auto stage = [&]() __attribute__((always_inline)) {
        stage_begin();
#pragma unroll
        for (int k = 0; k < 6; ++k) stage_piece(k);
    };
    // the two workgroups of a CU do identical work and would reach their epilogues together: the second one (dispatched in the second
    // half of the grid) starts `delay` x ~0.5 us late, so that one's epilogue falls into the other's k-loop
    if (p.delay > 0 && blockIdx.x >= gridDim.x / 2)
        for (int r = 0; r < p.delay; ++r) __builtin_amdgcn_s_sleep(16);
    stage();
    if (total > 1) stage();

    f32x4 acc[4][8];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int c_tile = blockIdx.x, c_k = 0;
    int after_epi = 0;                                                     // stages whose counted wait still has an epilogue's 32 stores behind it
    for (int s = 0; s < total; ++s) {
        // this wave's pieces of stage s have landed: younger than them are only the next stage's six (and an epilogue's stores)
        if (s + 1 >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (after_epi > 0) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        if (after_epi > 0) --after_epi;
        asm volatile("s_barrier" ::: "memory");
        const bool more = s + 2 < total;                                       // uniform
        if (more) stage_begin();
        const unsigned char* slot = lds + (s % NSTAGE) * STAGE;
        bf16x8 fa[8], fb[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) fb[nt] = *reinterpret_cast<const bf16x8*>(slot + fb0 + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) fa[mt] = *reinterpret_cast<const bf16x8*>(slot + fa0 + mt * 1024);
        // the stage after next is requested BETWEEN the MFMAs, one piece per row of four: a piece's issue (~60 cycles of the
        // texture-address path) then runs under MFMAs already in the pipe instead of in front of all of them
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[nt][mt], 0, 0, 0);
            if (mt < 6 && more) stage_piece(mt);
        }
        if (++c_k == nk) {
            // epilogue: lane (j = i, g) holds C[m = 16 mt + i][n = 64 w + 16 nt + 4 g + r], r = 0 .. 3
            const int tm = c_tile % tiles_m, tn = c_tile / tiles_m;
            const int n0 = tn * BN + wave * 64 + 4 * g;
            __bf16* crow = p.C + (size_t)(tm * BM + i) * p.N + n0;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n0 + nt * 16);
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    const f32x4 v = acc[nt][mt];
                    const bf16x4 w = {(__bf16)(v[0] + b4.x), (__bf16)(v[1] + b4.y), (__bf16)(v[2] + b4.z), (__bf16)(v[3] + b4.w)};
                    *reinterpret_cast<bf16x4*>(crow + (size_t)mt * 16 * p.N + nt * 16) = w;
                    acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            c_k = 0; c_tile += gridDim.x;
            after_epi = 2;
        }
    }
}
}  // namespace g2

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; return (uint16_t)u; }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t rng_state = 12345u;
static float urand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }

int main() {
    int dev = 0, cus = 256;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CK(hipFuncSetAttribute((const void*)g2::gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, g2::LDS_BYTES));
    CK(hipFuncSetAttribute((const void*)ppn::gemm::gemm_bf16_kernel<ppn::gemm::DENSE, ppn::gemm::EPI_BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                           ppn::gemm::LDS_BYTES));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)g2::gemm_kernel, g2::NT, g2::LDS_BYTES));
    printf("CUs %d, resident 4-wave workgroups per CU: %d\n", cus, occ);
    const int shapes[][3] = {{512, 512, 256}, {4096, 4096, 4096}, {65536, 1536, 512}, {65536, 1024, 512}, {65536, 512, 512}, {65536, 512, 1024},
                             {16384, 3072, 1024}, {16384, 1024, 2048}, {262144, 768, 256}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
        std::vector<float> hbias(N);
        for (auto& v : hA) v = f2bf(urand());
        for (auto& v : hB) v = f2bf(urand() * 0.25f);
        for (auto& v : hbias) v = urand();
        __bf16 *dA, *dB, *dC, *dC2; float* dbias;
        CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 2)); CK(hipMalloc(&dC2, (size_t)M * N * 2));
        CK(hipMalloc(&dbias, N * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dC, 0, (size_t)M * N * 2));
        g2::Params q{dA, dB, dC, dbias, M, N, K, getenv("G2_DELAY") ? atoi(getenv("G2_DELAY")) : 0};
        const int tiles2 = (M / g2::BM) * (N / g2::BN);
        const int grid2 = tiles2 < occ * cus ? tiles2 : occ * cus;
        ppn::gemm::Params p{};
        p.A = dA; p.B = dB; p.C = dC2; p.bias = dbias; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldc = N;
        const int tiles1 = (M / ppn::gemm::BM) * (N / ppn::gemm::BN);
        const int grid1 = tiles1 > cus ? cus : tiles1;
        auto go2 = [&]() { hipLaunchKernelGGL(g2::gemm_kernel, dim3(grid2), dim3(g2::NT), g2::LDS_BYTES, 0, q); };
        auto go1 = [&]() {
            hipLaunchKernelGGL((ppn::gemm::gemm_bf16_kernel<ppn::gemm::DENSE, ppn::gemm::EPI_BIAS>), dim3(grid1), dim3(ppn::gemm::NTHREADS), ppn::gemm::LDS_BYTES, 0, p);
        };
        go2(); go1();
        CK(hipDeviceSynchronize());
        std::vector<uint16_t> out((size_t)M * N), out1((size_t)M * N);
        CK(hipMemcpy(out.data(), dC, out.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(out1.data(), dC2, out1.size() * 2, hipMemcpyDeviceToHost));
        double maxerr = 0; size_t differ = 0;
        for (int r = 0; r < 64; ++r) {                                     // sampled rows against a float64 host reference
            const int m = (int)(((long long)r * 2654435761LL) % M);
            for (int n = 0; n < N; n += 7) {
                double a = hbias[n];
                for (int k = 0; k < K; ++k) a += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
                const double e = fabs(a - bf2f(out[(size_t)m * N + n])) / (fabs(a) + 1.0);
                if (e > maxerr) maxerr = e;
            }
        }
        for (size_t k = 0; k < out.size(); k += 13) differ += out[k] != out1[k];
        const int iters = M * (long long)N * K > (1LL << 36) ? 10 : 30;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms2 = 0, ms1 = 0;
        for (int rep = 0; rep < 2; ++rep) {                                // alternate the two kernels
            CK(hipEventRecord(e0)); for (int k = 0; k < iters; ++k) go2(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms2, e0, e1));
            CK(hipEventRecord(e0)); for (int k = 0; k < iters; ++k) go1(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms1, e0, e1));
        }
        const double fl = 2.0 * M * N * K;
        printf("M %6d N %5d K %5d: two 4-wave workgroups per CU %.4f ms (%5.0f TF/s)   shipped 8-wave core %.4f ms (%5.0f TF/s)   ratio %.2f   max rel err %.2e, "
               "sampled outputs differing from the shipped core's %zu\n", M, N, K, ms2 / iters, fl / (ms2 / iters) / 1e9, ms1 / iters, fl / (ms1 / iters) / 1e9,
               ms2 / ms1, maxerr, differ);
        fflush(stdout);
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dC2)); CK(hipFree(dbias));
    }
    return 0;
}
