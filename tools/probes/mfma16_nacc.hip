// v_mfma_f64_16x16x4_f64 issue rate vs the number of independent accumulator tiles per wave (2 waves/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256, 2) k_rate(long iters, double* out) {
    v4d acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = v4d{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    double sum = 0;
    for (int j = 0; j < NACC; ++j) sum += acc[j][0] + acc[j][3];
    if (sum == 12345.678) out[0] = sum;
}

template <int NACC> void run(double* out, int blocks_per_cu) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const long iters = 400000 / NACC;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_rate<NACC>), 256 * blocks_per_cu, 256, 0, 0, iters, out);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = 256.0 * blocks_per_cu * 4 * iters * NACC;
    printf("%2d accumulators, %d wave(s)/SIMD: %6.2f TFLOP/s  %5.1f cycles per MFMA per SIMD\n", NACC, blocks_per_cu, n * 2048 / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / (n / 1024));
}

int main() {
    double* out; hipMalloc(&out, 128);
    for (int w = 1; w <= 2; ++w) { run<1>(out, w); run<2>(out, w); run<4>(out, w); run<8>(out, w); run<12>(out, w); run<16>(out, w); run<20>(out, w); run<24>(out, w); }
    return 0;
}
