// What slows v_mfma_f64_4x4x4_4b_f64 down in a real GEMM loop?  Variants of the issue-rate probe.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CTRL> __device__ __forceinline__ double dpp_row(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
#define M44(a, b, c) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)

// MODE 0: same a, b.  1: 4 distinct a x 4 distinct b (loop-invariant).  2: as 1, b rotations recomputed by DPP every iteration.
// 3: as 2 plus operands reloaded from global memory every iteration (L1/L2 hits).  NACC accumulators = 16 * NT.
template <int MODE, int NT>
__global__ void __launch_bounds__(256) k_rate(long iters, const double* __restrict__ src, double* out) {
    double acc[NT][16];
    for (int n = 0; n < NT; ++n) for (int j = 0; j < 16; ++j) acc[n][j] = 0;
    double a[4], b[NT];
    for (int i = 0; i < 4; ++i) a[i] = 1.0 + (threadIdx.x + 64 * i) * 1e-9;
    for (int n = 0; n < NT; ++n) b[n] = 1.0 - (threadIdx.x + 7 * n) * 1e-9;
    const long long c0 = clock64();
    for (long it = 0; it < iters; ++it) {
        if (MODE == 3) {
            for (int i = 0; i < 4; ++i) a[i] = src[(it & 63) * 1024 + 64 * i + (threadIdx.x & 63)];
            for (int n = 0; n < NT; ++n) b[n] = src[(it & 63) * 1024 + 512 + 64 * n + (threadIdx.x & 63)];
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            double r0 = b[n], r1, r2, r3;
            if (MODE == 0) { r1 = r2 = r3 = r0; }
            else {
                r1 = dpp_row<0x12C>(r0); r2 = dpp_row<0x128>(r0); r3 = dpp_row<0x124>(r0);
                if (MODE == 1) { asm volatile("" : "+v"(r1), "+v"(r2), "+v"(r3)); }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double av = MODE == 0 ? a[0] : a[i];
                M44(av, r0, acc[n][4 * i + 0]); M44(av, r1, acc[n][4 * i + 1]); M44(av, r2, acc[n][4 * i + 2]); M44(av, r3, acc[n][4 * i + 3]);
            }
        }
        if (MODE == 1) { for (int n = 0; n < NT; ++n) asm volatile("" : "+v"(b[n])); }
    }
    double s = 0;
    for (int n = 0; n < NT; ++n) for (int j = 0; j < 16; ++j) s += acc[n][j];
    const long long c1 = clock64();
    if (s == 12345.678) out[0] = s;
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) out[1] = (double)(c1 - c0);
}

template <int MODE, int NT> void run(const char* name, int wps, const double* src, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const long iters = 200000 / (wps * NT);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_rate<MODE, NT>), 256 * wps, 256, 0, 0, iters, src, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    const double flops = 256.0 * wps * 4 * iters * NT * 4 * 2048;
    printf("%-34s NT=%d %d waves/SIMD: %6.2f TFLOP/s, %6.1f cycles per 16x16x4-equivalent\n", name, NT, wps, flops / (ms * 1e-3) / 1e12,
           h[1] / (iters * NT * 4.0 * wps));
}

int main() {
    double *src, *out;
    hipMalloc(&src, 64 * 1024 * 8); hipMemset(src, 0, 64 * 1024 * 8); hipMalloc(&out, 128);
    for (int wps = 1; wps <= 2; ++wps) {
        run<0, 1>("same operands", wps, src, out);
        run<1, 1>("distinct a/b, invariant", wps, src, out);
        run<2, 1>("distinct a/b + DPP in loop", wps, src, out);
        run<3, 1>("+ global loads in loop", wps, src, out);
        run<1, 5>("distinct a/b, invariant", wps, src, out);
        run<2, 5>("distinct a/b + DPP in loop", wps, src, out);
        run<3, 5>("+ global loads in loop", wps, src, out);
    }
    return 0;
}
