// gemm_bench.hip — standalone check + timing of the MFMA GEMM core (ppnet_amd/csrc/mfma_gemm.h) on the shapes PPNet runs.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/gemm_bench.hip -o tools/micro/gemm_bench
//   tools/micro/gemm_bench            (on the GPU box)
// Every shape: random bf16 operands in [-1, 1), result compared on sampled rows with a float64 host reference, then timed
// over interleaved launches (HIP events).  Diagnostic only; the product path calls the same kernels through capi.hip.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../ppnet_amd/csrc/mfma_gemm.h"

using namespace ppn::gemm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; return (uint16_t)u; }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t rng_state = 12345u;
static float urand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }

static int g_cus = 256;

template <int AMODE, int EPI>
static void launch(const Params& p, hipStream_t s) {
    static bool attr = false;
    if (!attr) { CK(hipFuncSetAttribute((const void*)gemm_bf16_kernel<AMODE, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)); attr = true; }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const bool aligned = p.M % BM == 0 && p.N % BN == 0;
    // persistent: one block per CU — except the classifier epilogue, which reduces its tile's sums once at the end of the workgroup
    // (one tile per workgroup: what conv3x3_mfma_launch launches)
    const int grid = (aligned && tiles > g_cus && !getenv("GEMM_ONE_TILE_PER_BLOCK") && EPI != EPI_RELU_DOT2) ? g_cus : tiles;
    hipLaunchKernelGGL((gemm_bf16_kernel<AMODE, EPI>), dim3(grid), dim3(NTHREADS), LDS_BYTES, s, p);
}

static int run_dense(int M, int N, int K, int epi, int iters) {
    std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    std::vector<float> hbias(N);
    for (auto& v : hA) v = f2bf(urand());
    for (auto& v : hB) v = f2bf(urand() * 0.25f);
    for (auto& v : hC) v = f2bf(urand());
    for (auto& v : hbias) v = urand();
    __bf16 *dA, *dB, *dC; float* dbias;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, hC.size() * 2)); CK(hipMalloc(&dbias, N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, hC.data(), hC.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice));
    Params p{};
    p.A = dA; p.B = dB; p.C = dC; p.bias = dbias; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldc = N;
    auto go = [&]() {
        if (epi == EPI_BIAS) launch<DENSE, EPI_BIAS>(p, 0);
        else if (epi == EPI_BIAS_GELU) launch<DENSE, EPI_BIAS_GELU>(p, 0);
        else launch<DENSE, EPI_ACCUM>(p, 0);
    };
    go();
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> out((size_t)M * N);
    CK(hipMemcpy(out.data(), dC, out.size() * 2, hipMemcpyDeviceToHost));
    double max_err = 0, max_ref = 0;
    int bad = 0;
    for (int s = 0; s < 96; ++s) {
        const int m = (s < 4) ? (s == 0 ? 0 : s == 1 ? M - 1 : s == 2 ? 255 : 256) % M : (int)((rng_state = rng_state * 1664525u + 1013904223u) % (uint32_t)M);
        for (int n = 0; n < N; ++n) {
            double acc = 0;
            for (int k = 0; k < K; ++k) acc += (double)bf2f(hA[(size_t)m * K + k]) * (double)bf2f(hB[(size_t)n * K + k]);
            double ref;
            if (epi == EPI_BIAS) ref = acc + hbias[n];
            else if (epi == EPI_BIAS_GELU) { const double x = acc + hbias[n]; ref = 0.5 * x * (1.0 + erf(x * 0.7071067811865476)); }
            else ref = acc + bf2f(hC[(size_t)m * N + n]);
            const double got = bf2f(out[(size_t)m * N + n]);
            const double err = fabs(got - ref);
            max_err = fmax(max_err, err); max_ref = fmax(max_ref, fabs(ref));
            if (err > 0.02 * fmax(1.0, fabs(ref))) { if (bad < 5) printf("   mismatch m=%d n=%d got %f ref %f\n", m, n, got, ref); ++bad; }
        }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) go();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) go();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    const double tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12;
    const double gbs = ((double)M * K * 2 + (double)M * N * 2 * (epi == EPI_ACCUM ? 2 : 1) + (double)N * K * 2) / (ms * 1e-3) / 1e9;
    printf("dense M=%8d N=%5d K=%5d epi=%d : %8.4f ms  %7.1f TF/s  %7.1f GB/s  max_err %.4f (max |ref| %.2f)  %s\n", M, N, K, epi, ms, tf, gbs,
           max_err, max_ref, bad ? "FAIL" : "ok");
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dbias));
    return bad;
}

static int run_conv(int Bn, int H, int W, int Cin, int Cout, int epi, int iters) {
    const int M = Bn * H * W, K = 9 * Cin, N = Cout, slots = (N + 63) / 64;
    std::vector<uint16_t> hA((size_t)M * Cin), hB((size_t)N * K);
    std::vector<float> hbias(N), hw2(2 * N);
    for (auto& v : hA) v = f2bf(urand());
    for (auto& v : hB) v = f2bf(urand() * 0.1f);
    for (auto& v : hbias) v = urand();
    for (auto& v : hw2) v = urand() * 0.1f;
    __bf16 *dA, *dB, *dC, *dz; float *dbias, *dw2, *dlog;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 2)); CK(hipMalloc(&dz, 256));
    CK(hipMalloc(&dbias, N * 4)); CK(hipMalloc(&dw2, 2 * N * 4)); CK(hipMalloc(&dlog, (size_t)slots * M * 2 * 4));       // EPI_RELU_DOT2 writes one partial sum per 64 columns: [N / 64][M][2]
    CK(hipMemset(dz, 0, 256)); CK(hipMemset(dlog, 0, (size_t)slots * M * 8));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw2, hw2.data(), 2 * N * 4, hipMemcpyHostToDevice));
    Params p{};
    p.A = dA; p.B = dB; p.C = dC; p.bias = dbias; p.M = M; p.N = N; p.K = K; p.lda = Cin; p.ldc = N; p.H = H; p.W = W; p.Cin = Cin; p.Ho = H; p.Wo = W; p.stride = 1; p.zero = dz;
    p.w2 = dw2; p.logits = dlog;
    auto go = [&]() { if (epi == EPI_BIAS_RELU) launch<CONV3, EPI_BIAS_RELU>(p, 0); else launch<CONV3, EPI_RELU_DOT2>(p, 0); };
    go();
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> out((size_t)M * N);
    std::vector<float> logit((size_t)slots * M * 2);
    CK(hipMemcpy(out.data(), dC, out.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(logit.data(), dlog, logit.size() * 4, hipMemcpyDeviceToHost));
    double max_err = 0, max_ref = 0;
    int bad = 0;
    for (int s = 0; s < 48; ++s) {
        int m = (int)((rng_state = rng_state * 1664525u + 1013904223u) % (uint32_t)M);
        if (s == 0) m = 0; if (s == 1) m = M - 1; if (s == 2) m = W - 1; if (s == 3) m = (H - 1) * W;
        const int x = m % W, y = (m / W) % H;
        double l0 = 0, l1 = 0;
        for (int n = 0; n < N; ++n) {
            double acc = 0;
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const size_t src = ((size_t)m + (size_t)((t / 3 - 1) * W + (t % 3 - 1))) * Cin;
                for (int c = 0; c < Cin; ++c) acc += (double)bf2f(hA[src + c]) * (double)bf2f(hB[(size_t)n * K + t * Cin + c]);
            }
            const double ref = fmax(acc + hbias[n], 0.0);
            if (epi == EPI_BIAS_RELU) {
                const double got = bf2f(out[(size_t)m * N + n]);
                const double err = fabs(got - ref);
                max_err = fmax(max_err, err); max_ref = fmax(max_ref, fabs(ref));
                if (err > 0.02 * fmax(1.0, fabs(ref))) { if (bad < 5) printf("   mismatch m=%d n=%d got %f ref %f\n", m, n, got, ref); ++bad; }
            } else { l0 += ref * hw2[n]; l1 += ref * hw2[N + n]; }
        }
        if (epi == EPI_RELU_DOT2) {
            for (int c = 0; c < 2; ++c) {
                double got = 0;
                for (int sl = 0; sl < slots; ++sl) got += logit[((size_t)sl * M + m) * 2 + c];
                const double ref = c ? l1 : l0, err = fabs(got - ref);
                max_err = fmax(max_err, err); max_ref = fmax(max_ref, fabs(ref));
                if (err > 2e-3 * fmax(1.0, fabs(ref))) { if (bad < 5) printf("   mismatch m=%d c=%d got %f ref %f\n", m, c, got, ref); ++bad; }
            }
        }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) go();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) go();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    printf("conv3 B=%3d %3dx%3d Cin=%4d Cout=%4d epi=%d : %8.4f ms  %7.1f TF/s  max_err %.4f (max |ref| %.2f)  %s\n", Bn, H, W, Cin, Cout, epi, ms,
           2.0 * M * N * K / (ms * 1e-3) / 1e12, max_err, max_ref, bad ? "FAIL" : "ok");
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dz)); CK(hipFree(dbias)); CK(hipFree(dw2)); CK(hipFree(dlog));
    return bad;
}

int main(int argc, char** argv) {
    { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); g_cus = pr.multiProcessorCount; printf("CUs: %d\n", g_cus); }
    int bad = 0;
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    if (argc > 1 && !strcmp(argv[1], "conv64")) {          // the SETR-UP head's last stage alone (for counter passes)
        bad += run_conv(256, 64, 64, 512, 512, EPI_RELU_DOT2, 5);
        bad += run_dense(16384, 4096, 4096, EPI_BIAS, 5);
        return bad != 0;
    }
    // small, ragged: correctness of clamping / partial tiles
    bad += run_dense(300, 264, 128, EPI_BIAS, 2);
    bad += run_dense(512, 256, 64, EPI_ACCUM, 2);
    bad += run_dense(256, 512, 192, EPI_BIAS_GELU, 2);
    bad += run_conv(2, 16, 16, 64, 256, EPI_BIAS_RELU, 2);
    bad += run_conv(3, 8, 8, 128, 512, EPI_RELU_DOT2, 2);
    // aligned, more tiles than CUs: the persistent path with an epilogue between tiles (every epilogue kind)
    bad += run_dense(256 * 40, 512, 128, EPI_BIAS, 2);
    bad += run_dense(256 * 40, 768, 192, EPI_ACCUM, 2);
    bad += run_dense(256 * 33, 512, 320, EPI_BIAS_GELU, 2);
    bad += run_conv(40, 16, 16, 64, 512, EPI_BIAS_RELU, 2);
    bad += run_conv(160, 8, 8, 128, 512, EPI_RELU_DOT2, 1);
    if (quick) return bad != 0;
    // reference point
    bad += run_dense(4096, 4096, 4096, EPI_BIAS, 10);
    bad += run_dense(8192, 8192, 8192, EPI_BIAS, 5);
    // DiNAT-B at 256 x 256, batch 256 (M = tokens): qkv / proj / fc1 / fc2 per level
    const int Ms[4] = {1048576, 262144, 65536, 16384}, Cs[4] = {128, 256, 512, 1024};
    for (int l = 1; l < 4; ++l) {
        bad += run_dense(Ms[l], 3 * Cs[l], Cs[l], EPI_BIAS, 10);
        bad += run_dense(Ms[l], Cs[l], Cs[l], EPI_ACCUM, 10);
        bad += run_dense(Ms[l], 2 * Cs[l], Cs[l], EPI_BIAS_GELU, 10);
        bad += run_dense(Ms[l], Cs[l], 2 * Cs[l], EPI_ACCUM, 10);
    }
    // SETR-UP head, batch 256
    bad += run_conv(256, 8, 8, 1024, 512, EPI_BIAS_RELU, 10);
    bad += run_conv(256, 16, 16, 512, 512, EPI_BIAS_RELU, 10);
    bad += run_conv(256, 32, 32, 512, 512, EPI_BIAS_RELU, 5);
    bad += run_conv(256, 64, 64, 512, 512, EPI_BIAS_RELU, 3);
    bad += run_conv(256, 64, 64, 512, 512, EPI_RELU_DOT2, 3);
    printf(bad ? "FAILURES: %d\n" : "all ok\n", bad);
    return bad != 0;
}
