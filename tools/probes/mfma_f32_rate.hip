// Issue rates of the fp32 MFMA shapes on gfx950 (independent accumulators, 2 waves/SIMD unless noted).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>
__global__ void __launch_bounds__(256) k_rate(long iters, float* out) {
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    float s = 0;
    if (MODE == 0) {                       // v_mfma_f32_16x16x4_f32: 2048 flop
        v4f acc[NACC];
        for (int j = 0; j < NACC; ++j) acc[j] = v4f{0, 0, 0, 0};
        for (long it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][3];
    } else if (MODE == 1) {                // v_mfma_f32_32x32x2_f32: 4096 flop
        v16f acc[NACC / 2];
        for (int j = 0; j < NACC / 2; ++j) for (int k = 0; k < 16; ++k) acc[j][k] = 0;
        for (long it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < NACC / 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        for (int j = 0; j < NACC / 2; ++j) s += acc[j][0] + acc[j][15];
    } else {                               // v_mfma_f32_4x4x1_16b_f32: 16 blocks x 32 flop = 512 flop
        v4f acc[NACC];
        for (int j = 0; j < NACC; ++j) acc[j] = v4f{0, 0, 0, 0};
        for (long it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0);
        for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][3];
    }
    if (s == 12345.678f) out[0] = s;
}

template <int MODE> void run(const char* name, double flop_per_instr, int div, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 2; ++wps) {
        const long iters = 200000 / wps;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((k_rate<MODE, 8>), 256 * wps, 256, 0, 0, iters, out);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n_instr = 256.0 * wps * 4 * iters * (8 / div);
        printf("%-28s %d waves/SIMD: %7.2f TFLOP/s  (%.1f cycles per instruction per SIMD at 2.4 GHz)\n", name, wps,
               n_instr * flop_per_instr / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (n_instr / 1024.0));
    }
}

int main() {
    float* out; hipMalloc(&out, 64);
    run<0>("v_mfma_f32_16x16x4_f32", 2048, 1, out);
    run<1>("v_mfma_f32_32x32x2_f32", 4096, 2, out);
    run<2>("v_mfma_f32_4x4x1_16b_f32", 512, 1, out);
    return 0;
}
