// gemm4w_bench.hip — the 4-wave GEMM core (ppnet_amd/csrc/mfma_gemm4w.h) against the 8-wave one (mfma_gemm.h) on the shapes PPNet
// runs: random bf16 operands, sampled rows checked against a float64 host reference, both kernels timed in alternation.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/gemm4w_bench.hip -o tools/micro/gemm4w_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../ppnet_amd/csrc/mfma_gemm.h"
#include "mfma_gemm4w.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; return (uint16_t)u; }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t rng_state = 12345u;
static float urand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }
static int g_cus = 256;

template <int EPI> static void launch_old(const ppn::gemm::Params& p) {
    using namespace ppn::gemm;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)gemm_bf16_kernel<DENSE, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)); attr = true; }
    const int tiles = (p.M / BM) * (p.N / BN);
    hipLaunchKernelGGL((gemm_bf16_kernel<DENSE, EPI>), dim3(tiles > g_cus ? g_cus : tiles), dim3(NTHREADS), LDS_BYTES, 0, p);
}
template <int EPI> static void launch_new(const ppn::g4::Params& p) {
    using namespace ppn::g4;
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)gemm4w_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)); attr = true; }
    const int tiles = (p.M / BM) * (p.N / BN);
    int grid = tiles > g_cus ? g_cus : tiles;
    if (grid > 8) grid &= ~7;
    hipLaunchKernelGGL((gemm4w_kernel<EPI>), dim3(grid), dim3(NTHREADS), LDS_BYTES, 0, p);
}

static int run(int M, int N, int K, int epi, int iters) {
    std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    std::vector<float> hbias(N);
    for (auto& v : hA) v = f2bf(urand());
    for (auto& v : hB) v = f2bf(urand() * 0.25f);
    for (auto& v : hC) v = f2bf(urand());
    for (auto& v : hbias) v = urand();
    __bf16 *dA, *dB, *dC, *dC2; float* dbias;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, hC.size() * 2)); CK(hipMalloc(&dC2, hC.size() * 2)); CK(hipMalloc(&dbias, N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, hC.data(), hC.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC2, hC.data(), hC.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice));
    ppn::gemm::Params po{};
    po.A = dA; po.B = dB; po.C = dC2; po.bias = dbias; po.M = M; po.N = N; po.K = K; po.lda = K; po.ldc = N;
    ppn::g4::Params pn{};
    pn.A = dA; pn.B = dB; pn.C = dC; pn.bias = dbias; pn.M = M; pn.N = N; pn.K = K; pn.lda = K; pn.ldc = N;
    auto go_new = [&]() { if (epi == 0) launch_new<0>(pn); else if (epi == 1) launch_new<1>(pn); else if (epi == 2) launch_new<2>(pn); else launch_new<3>(pn); };
    auto go_old = [&]() { if (epi == 0) launch_old<0>(po); else if (epi == 1) launch_old<1>(po); else if (epi == 2) launch_old<2>(po); else launch_old<3>(po); };
    go_new();
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> out((size_t)M * N);
    CK(hipMemcpy(out.data(), dC, out.size() * 2, hipMemcpyDeviceToHost));
    double max_err = 0, max_ref = 0;
    int bad = 0;
    for (int s = 0; s < 64; ++s) {
        const int m = (s < 6) ? (s == 0 ? 0 : s == 1 ? M - 1 : s == 2 ? 255 : s == 3 ? 256 : s == 4 ? 127 : 128) % M : (int)((rng_state = rng_state * 1664525u + 1013904223u) % (uint32_t)M);
        for (int n = 0; n < N; ++n) {
            double acc = 0;
            for (int k = 0; k < K; ++k) acc += (double)bf2f(hA[(size_t)m * K + k]) * (double)bf2f(hB[(size_t)n * K + k]);
            double ref;
            if (epi == 0) ref = acc + hbias[n];
            else if (epi == 1) { const double x = acc + hbias[n]; ref = 0.5 * x * (1.0 + erf(x * 0.7071067811865476)); }
            else if (epi == 2) ref = acc + bf2f(hC[(size_t)m * N + n]);
            else ref = fmax(acc + hbias[n], 0.0);
            const double got = bf2f(out[(size_t)m * N + n]);
            const double err = fabs(got - ref);
            max_err = fmax(max_err, err); max_ref = fmax(max_ref, fabs(ref));
            if (err > 0.02 * fmax(1.0, fabs(ref))) { if (bad < 5) printf("   mismatch m=%d n=%d got %f ref %f\n", m, n, got, ref); ++bad; }
        }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float t_new = 1e9f, t_old = 1e9f;
    for (int rnd = 0; rnd < 3; ++rnd) {
        for (int which = 0; which < 2; ++which) {
            for (int i = 0; i < 3; ++i) { if (which) go_old(); else go_new(); }
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) { if (which) go_old(); else go_new(); }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
            if (which) t_old = fminf(t_old, ms); else t_new = fminf(t_new, ms);
        }
    }
    const double fl = 2.0 * M * N * K;
    printf("M=%8d N=%5d K=%5d epi=%d : 4-wave %8.4f ms %7.1f TF/s | 8-wave %8.4f ms %7.1f TF/s | new/old %.3f  max_err %.4f (max |ref| %.2f)  %s\n", M, N, K, epi,
           t_new, fl / (t_new * 1e-3) / 1e12, t_old, fl / (t_old * 1e-3) / 1e12, t_new / t_old, max_err, max_ref, bad ? "FAIL" : "ok");
    fflush(stdout);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dC2)); CK(hipFree(dbias));
    return bad;
}

int main(int argc, char** argv) {
    { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); g_cus = pr.multiProcessorCount; printf("CUs: %d\n", g_cus); }
    int bad = 0;
    bad += run(512, 256, 128, 0, 2);
    bad += run(256 * 40, 512, 192, 2, 2);
    bad += run(256 * 33, 768, 320, 1, 2);
    bad += run(256 * 24, 256, 128, 3, 2);
    if (argc > 1 && !strcmp(argv[1], "quick")) return bad != 0;
    bad += run(4096, 4096, 4096, 0, 10);
    bad += run(8192, 8192, 8192, 0, 4);
    const int Ms[4] = {1048576, 262144, 65536, 16384}, Cs[4] = {128, 256, 512, 1024};
    for (int l = 1; l < 4; ++l) {
        bad += run(Ms[l], 3 * Cs[l], Cs[l], 0, 10);
        bad += run(Ms[l], Cs[l], Cs[l], 2, 10);
        bad += run(Ms[l], 2 * Cs[l], Cs[l], 1, 10);
        bad += run(Ms[l], Cs[l], 2 * Cs[l], 2, 10);
    }
    printf(bad ? "FAILED\n" : "all ok\n");
    return bad != 0;
}
