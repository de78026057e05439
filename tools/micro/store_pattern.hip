// Diagnostic microbenchmark: 10000 workgroups x 64 KiB of stores, lane stride 32 B (two 16-B stores per lane, the
// stage-B pattern) against lane stride 16 B (fully contiguous per instruction).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(uint4* out, unsigned v) {
    uint4* g = out + (size_t)blockIdx.x * 4096;              // 64 KiB per block
    const int tid = threadIdx.x;
    const uint4 a = make_uint4(v, v + tid, v, v), b = make_uint4(v + 1, v, v + tid, v);
    if (MODE == 0) {
        for (int w = tid; w < 2048; w += 256) { g[2 * w] = a; g[2 * w + 1] = b; }
    } else if (MODE == 1) {
        for (int w = tid; w < 4096; w += 512) { g[w] = a; g[w + 256] = b; }
    } else {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        v4u* gv = reinterpret_cast<v4u*>(g);
        const v4u av = {a.x, a.y, a.z, a.w}, bv = {b.x, b.y, b.z, b.w};
        for (int w = tid; w < 4096; w += 512) { __builtin_nontemporal_store(av, &gv[w]); __builtin_nontemporal_store(bv, &gv[w + 256]); }
    }
}
template <int MODE> float run(uint4* d, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(n), dim3(256), 0, 0, d, i);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(n), dim3(256), 0, 0, d, i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20;
}
int main() {
    const int n = 10000; uint4* d; hipMalloc(&d, (size_t)n * 65536);
    float a = run<0>(d, n), b = run<1>(d, n), c = run<2>(d, n);
    printf("stride-32B pairs %.4f ms (%.2f TB/s) | contiguous %.4f ms (%.2f TB/s) | contiguous nontemporal %.4f ms (%.2f TB/s)\n",
           a, n * 65536.0 / a / 1e9, b, n * 65536.0 / b / 1e9, c, n * 65536.0 / c / 1e9);
    return 0;
}
