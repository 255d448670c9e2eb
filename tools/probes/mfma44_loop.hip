// Candidate inner loops for an fp64 GEMM on v_mfma_f64_4x4x4_4b_f64: how close to the 75 TFLOP/s issue rate does a
// realistic k-step (4 A tiles rotated by DPP, 5 B tiles, operands prefetched from memory) get, and does forcing the
// instruction interleave (sched_group_barrier) help?
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CTRL> __device__ __forceinline__ double dpp_row(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
#define M44(a, b, c) c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0)

constexpr int NB = 4, ST = 5;

typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void kstep_vec(const double (&a)[NB], const double (&b)[ST], v4d (&acc)[NB][ST]) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const double r1 = dpp_row<0x12C>(a[i]), r2 = dpp_row<0x128>(a[i]), r3 = dpp_row<0x124>(a[i]);
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            M44(a[i], b[s], acc[i][s][0]); M44(r1, b[s], acc[i][s][1]); M44(r2, b[s], acc[i][s][2]); M44(r3, b[s], acc[i][s][3]);
        }
    }
}

template <int SCHED>
__device__ __forceinline__ void kstep(const double (&a)[NB], const double (&b)[ST], double (&acc)[NB][ST][4]) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const double r1 = dpp_row<0x12C>(a[i]), r2 = dpp_row<0x128>(a[i]), r3 = dpp_row<0x124>(a[i]);
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            M44(a[i], b[s], acc[i][s][0]); M44(r1, b[s], acc[i][s][1]); M44(r2, b[s], acc[i][s][2]); M44(r3, b[s], acc[i][s][3]);
        }
    }
    if (SCHED == 1) {
        // 80 MFMA, 24 DPP movs, 9 loads per k-step: one non-MFMA instruction behind every MFMA until they run out
#pragma unroll
        for (int i = 0; i < 24; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); }
#pragma unroll
        for (int i = 0; i < 9; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 47, 0);
    }
}

// LOADS 0: operands loop-invariant.  1: operands of k-step k+2 loaded from (L2-resident) memory while k runs.
template <int LOADS, int SCHED>
__global__ void __launch_bounds__(256, 2) k_loop(long iters, const double* __restrict__ src, double* out) {
    double acc[NB][ST][4];
    for (int i = 0; i < NB; ++i) for (int s = 0; s < ST; ++s) for (int t = 0; t < 4; ++t) acc[i][s][t] = 0;
    double a0[NB], b0[ST], a1[NB], b1[ST];
    const double* p = src + (threadIdx.x & 63) + (blockIdx.x & 7) * 4096;
    for (int i = 0; i < NB; ++i) { a0[i] = p[64 * i]; a1[i] = p[1024 + 64 * i]; }
    for (int s = 0; s < ST; ++s) { b0[s] = p[512 + 64 * s]; b1[s] = p[1536 + 64 * s]; }
    const long long c0 = clock64();
    for (long it = 0; it < iters; it += 2) {
        double a2[NB], b2[ST], a3[NB], b3[ST];
        if (LOADS) {
            const double* q = p + ((it + 2) & 62) * 1024;
            for (int i = 0; i < NB; ++i) a2[i] = q[64 * i];
            for (int s = 0; s < ST; ++s) b2[s] = q[512 + 64 * s];
        }
        kstep<SCHED>(a0, b0, acc);
        if (LOADS) {
            const double* q = p + ((it + 3) & 63) * 1024;
            for (int i = 0; i < NB; ++i) a3[i] = q[64 * i];
            for (int s = 0; s < ST; ++s) b3[s] = q[512 + 64 * s];
        }
        kstep<SCHED>(a1, b1, acc);
        if (LOADS) {
            for (int i = 0; i < NB; ++i) { a0[i] = a2[i]; a1[i] = a3[i]; }
            for (int s = 0; s < ST; ++s) { b0[s] = b2[s]; b1[s] = b3[s]; }
        } else {
            for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(a0[i]), "+v"(a1[i]));
        }
    }
    const long long c1 = clock64();
    double sum = 0;
    for (int i = 0; i < NB; ++i) for (int s = 0; s < ST; ++s) for (int t = 0; t < 4; ++t) sum += acc[i][s][t];
    if (sum == 12345.678) out[0] = sum;
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) out[1] = (double)(c1 - c0);
}

// same loop with the accumulators held as 4-vectors (8 consecutive VGPRs per 16x16 tile), as the GEMM kernel holds them
template <int ASTR, int BSTR>
__global__ void __launch_bounds__(256, 2) k_loop_vec(long iters, const double* __restrict__ src, double* out) {
    v4d acc[NB][ST];
    for (int i = 0; i < NB; ++i) for (int s = 0; s < ST; ++s) acc[i][s] = v4d{0, 0, 0, 0};
    double a0[NB], b0[ST], a1[NB], b1[ST];
    // lane (lr, lq) reads row lq (ASTR / BSTR doubles apart), 16 consecutive doubles per row and tile: the GEMM's operand pattern
    const int lr = threadIdx.x & 15, lq = (threadIdx.x >> 4) & 3;
    const double* pa = src + (blockIdx.x & 7) * 4096 + lq * ASTR + lr;
    const double* pb = src + 65536 + (blockIdx.x & 7) * 4096 + lq * BSTR + lr;
    for (int i = 0; i < NB; ++i) { a0[i] = pa[16 * i]; a1[i] = pa[4 * ASTR + 16 * i]; }
    for (int s = 0; s < ST; ++s) { b0[s] = pb[16 * s]; b1[s] = pb[4 * BSTR + 16 * s]; }
    const long long c0 = clock64();
    for (long it = 0; it < iters; it += 2) {
        double a2[NB], b2[ST], a3[NB], b3[ST];
        { const int k = (it + 2) & 14;
          for (int i = 0; i < NB; ++i) a2[i] = pa[(size_t)4 * k * ASTR + 16 * i];
          for (int s = 0; s < ST; ++s) b2[s] = pb[(size_t)4 * k * BSTR + 16 * s]; }
        kstep_vec(a0, b0, acc);
        { const int k = (it + 3) & 15;
          for (int i = 0; i < NB; ++i) a3[i] = pa[(size_t)4 * k * ASTR + 16 * i];
          for (int s = 0; s < ST; ++s) b3[s] = pb[(size_t)4 * k * BSTR + 16 * s]; }
        kstep_vec(a1, b1, acc);
        for (int i = 0; i < NB; ++i) { a0[i] = a2[i]; a1[i] = a3[i]; }
        for (int s = 0; s < ST; ++s) { b0[s] = b2[s]; b1[s] = b3[s]; }
    }
    const long long c1 = clock64();
    double sum = 0;
    for (int i = 0; i < NB; ++i) for (int s = 0; s < ST; ++s) for (int t = 0; t < 4; ++t) sum += acc[i][s][t];
    if (sum == 12345.678) out[0] = sum;
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) out[1] = (double)(c1 - c0);
}

template <int LOADS, int SCHED> void run(const char* name, const double* src, double* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const long iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_loop<LOADS, SCHED>), 512, 256, 0, 0, iters, src, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    const double flops = 512.0 * 4 * iters * NB * ST * 2048;
    printf("%-44s %6.2f TFLOP/s   %6.1f cycles per k-step per wave (2 waves/SIMD; 1280 = MFMA pipe time of one wave)\n", name,
           flops / (ms * 1e-3) / 1e12, h[1] / iters);
}

int main() {
    double *src, *out;
    hipMalloc(&src, 4 << 20); hipMemset(src, 0, 4 << 20); hipMalloc(&out, 128);
    run<0, 0>("invariant operands, compiler schedule", src, out);
    run<0, 1>("invariant operands, forced interleave", src, out);
    run<1, 0>("prefetched loads, compiler schedule", src, out);
    run<1, 1>("prefetched loads, forced interleave", src, out);
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const long iters = 20000;
        float ms;
#define RUNV(A, B) for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0, 0); hipLaunchKernelGGL((k_loop_vec<A, B>), 512, 256, 0, 0, iters, src, out); hipEventRecord(e1, 0); hipEventSynchronize(e1); } \
        hipEventElapsedTime(&ms, e0, e1); printf("vector accumulators, rot A, row strides A %4d B %4d doubles: %6.2f TFLOP/s\n", A, B, 512.0 * 4 * iters * NB * ST * 2048 / (ms * 1e-3) / 1e12);
        RUNV(16, 16) RUNV(256, 80) RUNV(256, 16) RUNV(16, 80) RUNV(64, 64)
    }
    return 0;
}
