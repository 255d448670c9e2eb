// Do MFMA and VALU instructions of a SIMD execute at the same time on gfx950?  Per loop iteration a wave issues NM MFMAs and NV VALU
// instructions, finely interleaved in program order (one MFMA, then NV / NM VALU); the wave's cycles per iteration (s_memtime) with
// the MFMAs only, the VALU only, and both tell: both = max -> concurrent, both = sum -> the two share the SIMD's issue / datapath.
// MFMA kinds: i8 16x16x64 (4 passes = 16 cycles), f64 16x16x4 (64 cycles).  VALU kinds: f64 fma, i32 shift-add, f32 fma.
// Waves per SIMD: 1 or 2 (block 256 or 512 on one workgroup per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v4d __attribute__((ext_vector_type(4)));

template <int MK, int VK, bool DO_M, bool DO_V>
__global__ void __launch_bounds__(512, 1) k_mix(long iters, double* out) {
    v4i ai = {(int)threadIdx.x, 1, 2, 3}, bi = {3, 2, 1, (int)threadIdx.x};
    v4i acci[6];
    v4d accd[6];
    for (int i = 0; i < 6; ++i) { acci[i] = v4i{0, 0, 0, 0}; accd[i] = v4d{0, 0, 0, 0}; }
    constexpr int NC = 16;                         // independent VALU chains: a chain is touched every NC-th VALU instruction (no latency bound)
    double xd[NC]; int xi[NC]; float xf[NC];
    for (int i = 0; i < NC; ++i) { xd[i] = 1.0 + threadIdx.x * 1e-9 * (i + 1); xi[i] = threadIdx.x + i; xf[i] = 1.0f + threadIdx.x * 1e-6f * (i + 1); }
    const double cd = 0.999999; const float cf = 0.99999f;
    const long long c0 = clock64();
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            if (DO_M) {
                if (MK == 0) acci[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ai, bi, acci[m], 0, 0, 0);
                else accd[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(xd[0] * 0 + 1.0, 1.0, accd[m], 0, 0, 0);
            }
            if (DO_V) {
                constexpr int NVM = MK == 0 ? 4 : 16;             // VALU instructions per MFMA: 4 x 4 cycles = the i8 MFMA's 16, 16 x 4 = the f64 MFMA's 64
#pragma unroll
                for (int v = 0; v < NVM; ++v) {
                    const int j = (m * NVM + v) % NC;
                    if (VK == 0) xd[j] = fma(xd[j], cd, 1e-9);
                    else if (VK == 1) xi[j] = (xi[j] << 1) + 12345;
                    else xf[j] = fmaf(xf[j], cf, 1e-6f);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long c1 = clock64();
    double sum = 0;
    for (int i = 0; i < 6; ++i) sum += acci[i][0] + accd[i][1];
    for (int i = 0; i < NC; ++i) sum += xd[i] + xi[i] + xf[i];
    if (sum == 12345.678) out[0] = sum;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = (double)(c1 - c0) / iters;
}

template <int MK, int VK> void run(const char* name, int block, double* out) {
    const long iters = 20000;
    double r[3];
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL((k_mix<MK, VK, true, false>), 256, block, 0, 0, iters, out);
            else if (mode == 1) hipLaunchKernelGGL((k_mix<MK, VK, false, true>), 256, block, 0, 0, iters, out);
            else hipLaunchKernelGGL((k_mix<MK, VK, true, true>), 256, block, 0, 0, iters, out);
            hipDeviceSynchronize();
        }
        double h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        r[mode] = h[1];
    }
    printf("%-34s %d wave(s)/SIMD: cycles per 6 MFMA + VALU group   MFMA only %7.1f   VALU only %7.1f   both %7.1f   (max %7.1f, sum %7.1f)\n", name,
           block / 256, r[0], r[1], r[2], r[0] > r[1] ? r[0] : r[1], r[0] + r[1]);
}

int main() {
    double* out; hipMalloc(&out, 64);
    for (int block = 256; block <= 512; block += 256) {
        run<0, 0>("i8 16x16x64 + f64 fma", block, out);
        run<0, 1>("i8 16x16x64 + i32 shift-add", block, out);
        run<0, 2>("i8 16x16x64 + f32 fma", block, out);
        run<1, 0>("f64 16x16x4 + f64 fma", block, out);
        run<1, 1>("f64 16x16x4 + i32 shift-add", block, out);
        run<1, 2>("f64 16x16x4 + f32 fma", block, out);
    }
    return 0;
}
