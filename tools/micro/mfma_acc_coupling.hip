// Does a huge value in ONE accumulator element of v_mfma_f32_16x16x32_bf16 change the OTHER elements of the result?
// (na2d_halo16: the window mask as -1e30 in the initial accumulator.)  One wave; A, B random bf16, C random f32; the second product has
// C[e] = -1e30 on a pseudo-random subset of elements.  Prints how many of the untouched elements differ bitwise.
//   hipcc -O2 --offload-arch=gfx950 tools/micro/mfma_acc_coupling.hip -o tools/micro/mfma_acc_coupling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const uint16_t* a, const uint16_t* b, const float* c, const float* c2, float* d, float* d2, int n) {
    const int lane = threadIdx.x;
    for (int i = 0; i < n; ++i) {
        bf16x8 av, bv;
        for (int e = 0; e < 8; ++e) {
            av[e] = __builtin_bit_cast(__bf16, a[(i * 64 + lane) * 8 + e]);
            bv[e] = __builtin_bit_cast(__bf16, b[(i * 64 + lane) * 8 + e]);
        }
        f32x4 cv = *reinterpret_cast<const f32x4*>(c + (i * 64 + lane) * 4), cw = *reinterpret_cast<const f32x4*>(c2 + (i * 64 + lane) * 4);
        f32x4 r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, cv, 0, 0, 0);
        f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, cw, 0, 0, 0);
        *reinterpret_cast<f32x4*>(d + (i * 64 + lane) * 4) = r;
        *reinterpret_cast<f32x4*>(d2 + (i * 64 + lane) * 4) = s;
    }
}
static float rnd() { return (float)rand() / RAND_MAX * 2.f - 1.f; }
static uint16_t bf(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
int main() {
    const int n = 256, N = n * 64;
    uint16_t *a, *b; float *c, *c2, *d, *d2;
    hipMallocManaged(&a, N * 16); hipMallocManaged(&b, N * 16);
    hipMallocManaged(&c, N * 16); hipMallocManaged(&c2, N * 16); hipMallocManaged(&d, N * 16); hipMallocManaged(&d2, N * 16);
    for (int i = 0; i < N * 8; ++i) { a[i] = bf(rnd() * 3.f); b[i] = bf(rnd() * 3.f); }
    int masked = 0;
    for (int i = 0; i < N * 4; ++i) { c[i] = rnd() * 2.f; c2[i] = c[i]; if (rand() % 3 == 0) { c2[i] = -1.0e30f; ++masked; } }
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, c, c2, d, d2, n);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    int diff = 0, bad_masked = 0; double maxrel = 0;
    for (int i = 0; i < N * 4; ++i) {
        if (c2[i] != c[i]) { if (d2[i] != -1.0e30f) ++bad_masked; continue; }
        uint32_t x, y; __builtin_memcpy(&x, &d[i], 4); __builtin_memcpy(&y, &d2[i], 4);
        if (x != y) { ++diff; double r = fabs((double)d[i] - d2[i]) / (fabs((double)d[i]) + 1e-30); if (r > maxrel) maxrel = r; }
    }
    printf("elements %d, masked %d (of them not exactly -1e30 afterwards: %d), untouched elements that differ: %d, max relative difference %.3g\n", N * 4, masked,
           bad_masked, diff, maxrel);
    return 0;
}
