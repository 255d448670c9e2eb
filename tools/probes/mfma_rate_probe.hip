// Probe (round 6): issue cost of v_mfma_f64_4x4x4 (4 blocks) against v_mfma_f64_16x16x4 on gfx950, one wave per SIMD, eight
// independent accumulators, back to back.   build: hipcc --offload-arch=gfx950 -O2 -o mfma_rate_probe mfma_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k16(double* out, long long* cyc, int iters) {
    const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    d4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = d4{0, 0, 0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    const long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
__global__ void k4(double* out, long long* cyc, int iters) {
    const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double c[8];
    for (int i = 0; i < 8; ++i) c[i] = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
    const long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += c[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
int main() {
    double* out; long long* cyc; long long h[16];
    hipMalloc(&out, 1024 * sizeof(double)); hipMalloc(&cyc, 16 * sizeof(long long));
    const int iters = 20000;
    for (int waves = 4; waves <= 16; waves *= 2) {
        hipLaunchKernelGGL(k16, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
        hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        printf("16x16x4, %d waves on one CU: %.1f clock64 ticks per MFMA per wave (last wave %.1f)\n", waves, (double)h[0] / (8.0 * iters), (double)h[waves - 1] / (8.0 * iters));
        hipLaunchKernelGGL(k4, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
        hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        printf("4x4x4 (4 blocks), %d waves on one CU: %.1f clock64 ticks per MFMA per wave\n", waves, (double)h[0] / (8.0 * iters));
    }
    return 0;
}
