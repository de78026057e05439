// mfma_gemm.hip — launchers of the bf16 MFMA GEMM core (mfma_gemm.h): the implicit-GEMM 3x3 convolutions of the SETR-UP head
// and the NAT downsamplers (reference SegNet/mmseg/decode_heads/setr_up_head.py:53-66, SegNet/nat.py:48-59), and the dense
// projection form (SegNet/nat.py:62-85,111-120).
// The library is built with -ffp-contract=off because the generator kernels promise unfused IEEE double arithmetic (the parity
// contract with the oracle).  This file holds network arithmetic checked against float32 / float64 references to a tolerance:
// here a * b + c is one v_fma (otherwise every multiply-add of the LayerNorm, the GELU polynomial and the per-token linear
// phases is two instructions — these kernels are VALU-bound).
#pragma clang fp contract(fast)
#include "ppn_kernels.h"
#include "mfma_gemm.h"

namespace ppn {

// >= 256 bytes of zeros in device memory (the source of out-of-image taps / out-of-halo slots)
const void* zero_line() {
    static DeviceBuffer z;                                                   // one line per device
    const int dev = current_device();
    if (dev < 0) return nullptr;
    void* p = z.p[dev].load();
    if (!p) {
        void* q = nullptr;
        if (hipMalloc(&q, 256) != hipSuccess || hipMemset(q, 0, 256) != hipSuccess) return nullptr;
        void* expect = nullptr;
        if (!z.p[dev].compare_exchange_strong(expect, q)) { (void)hipFree(q); q = expect; }
        p = q;
    }
    return p;
}

namespace {
template <int AMODE, int EPI>
int launch(const gemm::Params& p, int persistent, hipStream_t stream) {
    static DeviceOnce attr;
    if (const int e = dynamic_lds_once(attr, (const void*)gemm::gemm_bf16_kernel<AMODE, EPI>, gemm::lds_bytes(EPI))) return e;
    const int tiles = ((p.M + gemm::BM - 1) / gemm::BM) * ((p.N + gemm::BN - 1) / gemm::BN);
    int grid = tiles;
    if (persistent > 0 && tiles > persistent && p.M % gemm::BM == 0 && p.N % gemm::BN == 0) grid = persistent;   // one block per CU
    hipLaunchKernelGGL((gemm::gemm_bf16_kernel<AMODE, EPI>), dim3(grid), dim3(gemm::NTHREADS), gemm::lds_bytes(EPI), stream, p);
    return (int)hipGetLastError();
}

}  // namespace

// logits[m][c] += sum over the column slots, in slot order (logits holds the classifier's bias on entry)
__global__ __launch_bounds__(256) void classify2_reduce_kernel(const float* __restrict__ partial, float* __restrict__ logits, int M, int slots) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;           // one float2 = one pixel
    if (i >= M) return;
    float2 a = reinterpret_cast<const float2*>(logits)[i];
    for (int s = 0; s < slots; ++s) {
        const float2 v = reinterpret_cast<const float2*>(partial)[(size_t)s * M + i];
        a.x += v.x; a.y += v.y;
    }
    reinterpret_cast<float2*>(logits)[i] = a;
}

int conv3x3_mfma_launch(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout, int stride,
                        int relu, const float* w2, float* logits, float* partial, hipStream_t stream) {
    gemm::Params p{};
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    p.A = (const __bf16*)x; p.B = (const __bf16*)w; p.C = (__bf16*)y; p.bias = bias;
    p.M = B * Ho * Wo; p.N = Cout; p.K = 9 * Cin; p.lda = Cin; p.ldc = Cout;
    p.H = H; p.W = W; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.stride = stride;
    p.zero = (const __bf16*)zero_line();
    if (!p.zero) return (int)hipErrorOutOfMemory;
    p.w2 = w2;
    if (logits) {
        // one slot per 256-column block (its four wave columns are added in order inside the workgroup); a single block adds straight
        // onto the logits and needs no workspace
        const int slots = (Cout + 255) / 256;
        if (slots > 1 && !partial) return (int)hipErrorInvalidValue;
        p.logits = slots > 1 ? partial : logits;
        // columns beyond N inside the last 256-wide block contribute zeros (their w2 is read as 0), so every slot of every row is written
        const int e = launch<gemm::CONV3, gemm::EPI_RELU_DOT2>(p, 0, stream);
        if (e != 0 || slots == 1) return e;
        hipLaunchKernelGGL(classify2_reduce_kernel, dim3((unsigned)((p.M + 255) / 256)), dim3(256), 0, stream, partial, logits, p.M, slots);
        return (int)hipGetLastError();
    }
    return relu ? launch<gemm::CONV3, gemm::EPI_BIAS_RELU>(p, 0, stream) : launch<gemm::CONV3, gemm::EPI_BIAS>(p, 0, stream);
}

int gemm_mfma_launch(const void* a, const void* w, const float* bias, void* c, long long M, int N, int K, int epi, int persistent,
                     hipStream_t stream) {
    gemm::Params p{};
    p.A = (const __bf16*)a; p.B = (const __bf16*)w; p.C = (__bf16*)c; p.bias = bias;
    p.M = (int)M; p.N = N; p.K = K; p.lda = K; p.ldc = N;
    switch (epi) {
        case 0: return launch<gemm::DENSE, gemm::EPI_BIAS>(p, persistent, stream);
        case 1: return launch<gemm::DENSE, gemm::EPI_BIAS_GELU>(p, persistent, stream);
        case 2: return launch<gemm::DENSE, gemm::EPI_ACCUM>(p, persistent, stream);
        case 3: return launch<gemm::DENSE, gemm::EPI_BIAS_RELU>(p, persistent, stream);
        default: return -1;
    }
}

// ppn_nat_gemm_bf16 mode 2 (round 5): c += a w^T + bias in place + row partials of the new c, on the 256 x 256 core with the old c
// read in the epilogue.  M % 256 == 0, N % 256 == 0, K % 64 == 0, K >= 128; p128: one partial per 128 columns (else per 256).
int gemm_acc_stats_launch(const void* a, const void* w, const float* bias, float* stats, void* c, long long M, int N, int K, int p128,
                          int n_cu, hipStream_t stream) {
    gemm::Params p{};
    p.A = (const __bf16*)a; p.B = (const __bf16*)w; p.C = (__bf16*)c; p.bias = bias;
    p.M = (int)M; p.N = N; p.K = K; p.lda = K; p.ldc = N;
    p.stats = stats; p.stats_p128 = p128;
    static DeviceOnce attr;
    if (const int e = dynamic_lds_once(attr, (const void*)gemm::gemm_bf16_kernel<gemm::DENSE, gemm::EPI_ACCUM_STATS>, gemm::lds_bytes(gemm::EPI_ACCUM_STATS))) return e;
    const int tiles = (p.M / gemm::BM) * (p.N / gemm::BN);
    int grid = tiles < n_cu ? tiles : n_cu;
    if (grid > 8) grid &= ~7;                                        // the XCD-aware tile order wants a multiple of 8
    hipLaunchKernelGGL((gemm::gemm_bf16_kernel<gemm::DENSE, gemm::EPI_ACCUM_STATS>), dim3(grid), dim3(gemm::NTHREADS),
                       gemm::lds_bytes(gemm::EPI_ACCUM_STATS), stream, p);
    return (int)hipGetLastError();
}

// ppn_nat_gemm_bf16 modes 0 / 1 (round 5): c = [gelu](LN(a) w^T + b) with the LayerNorm folded into the epilogue of the 256 x 256 core
// (persistent launch, the two wave groups' epilogues side by side); stats [parts][M][2] = row partials of a.
int gemm_ln_launch(const void* a, const void* w, const float* bias, const float* colsum, const float* stats, int parts, void* c, long long M,
                   int N, int K, int gelu, float eps, int n_cu, hipStream_t stream) {
    gemm::Params p{};
    p.A = (const __bf16*)a; p.B = (const __bf16*)w; p.C = (__bf16*)c; p.bias = bias; p.colsum = colsum;
    p.M = (int)M; p.N = N; p.K = K; p.lda = K; p.ldc = N;
    p.stats = const_cast<float*>(stats); p.stats_parts = parts; p.inv_k = 1.0f / (float)K; p.eps = eps;
    const int tiles = (p.M / gemm::BM) * (p.N / gemm::BN);
    int grid = tiles < n_cu ? tiles : n_cu;
    if (grid > 8) grid &= ~7;
    if (gelu) {
        static DeviceOnce attr;
        if (const int e = dynamic_lds_once(attr, (const void*)gemm::gemm_bf16_kernel<gemm::DENSE, gemm::EPI_LN_BIAS_GELU>, gemm::LDS_BYTES)) return e;
        hipLaunchKernelGGL((gemm::gemm_bf16_kernel<gemm::DENSE, gemm::EPI_LN_BIAS_GELU>), dim3(grid), dim3(gemm::NTHREADS), gemm::LDS_BYTES, stream, p);
    } else {
        static DeviceOnce attr;
        if (const int e = dynamic_lds_once(attr, (const void*)gemm::gemm_bf16_kernel<gemm::DENSE, gemm::EPI_LN_BIAS>, gemm::LDS_BYTES)) return e;
        hipLaunchKernelGGL((gemm::gemm_bf16_kernel<gemm::DENSE, gemm::EPI_LN_BIAS>), dim3(grid), dim3(gemm::NTHREADS), gemm::LDS_BYTES, stream, p);
    }
    return (int)hipGetLastError();
}

}  // namespace ppn
