// Issue rate of v_mfma_f64_16x16x4_f64 with the GEMM kernel's register pattern: 4 A operands x 5 B operands -> 20 independent
// accumulator tiles, operands distinct (and optionally refreshed every iteration), vs the same-operand probe of gpu_probe.py.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256, 2) k_rate(long iters, const double* __restrict__ src, double* out) {
    v4d acc[4][5];
    for (int i = 0; i < 4; ++i) for (int s = 0; s < 5; ++s) acc[i][s] = v4d{0, 0, 0, 0};
    double a[4], b[5];
    for (int i = 0; i < 4; ++i) a[i] = 1.0 + (threadIdx.x + 64 * i) * 1e-9;
    for (int s = 0; s < 5; ++s) b[s] = 1.0 - (threadIdx.x + 7 * s) * 1e-9;
    const long long c0 = clock64();
    for (long it = 0; it < iters; ++it) {
        if (MODE == 2) {
            const double* q = src + (it & 15) * 1024 + (threadIdx.x & 63);
            for (int i = 0; i < 4; ++i) a[i] = q[64 * i];
            for (int s = 0; s < 5; ++s) b[s] = q[512 + 64 * s];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < 5; ++s)
                acc[i][s] = __builtin_amdgcn_mfma_f64_16x16x4f64(MODE == 0 ? a[0] : a[i], MODE == 0 ? b[0] : b[s], acc[i][s], 0, 0, 0);
        if (MODE == 1) { for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i])); }
    }
    const long long c1 = clock64();
    double sum = 0;
    for (int i = 0; i < 4; ++i) for (int s = 0; s < 5; ++s) sum += acc[i][s][0] + acc[i][s][3];
    if (sum == 12345.678) out[0] = sum;
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) out[1] = (double)(c1 - c0);
}

template <int MODE> void run(const char* name, const double* src, double* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_rate<MODE>), 512, 256, 0, 0, iters, src, out);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("%-46s %6.2f TFLOP/s   %6.1f cycles per MFMA per SIMD\n", name, 512.0 * 4 * iters * 20 * 2048 / (ms * 1e-3) / 1e12,
           h[1] / (iters * 20.0 * 2));
}

int main() {
    double *src, *out;
    hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20); hipMalloc(&out, 128);
    run<0>("same a, b registers, 20 accumulators", src, out);
    run<1>("4 x 5 distinct operands, loop-invariant", src, out);
    run<2>("4 x 5 operands reloaded every iteration", src, out);
    return 0;
}
