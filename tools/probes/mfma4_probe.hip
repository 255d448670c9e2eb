// Probe (round 6): which source lanes feed each output lane of v_mfma_f64_4x4x4 (4 blocks) under the CBSZ / ABID broadcast controls?
// One-hot src0 (src1 = ones): workgroup b sets lane b of src0 to 1; D[lane] counts how often that element is used by the lane.
// Question behind it: can the 16-column B operand of a 16x16x4 float64 product (lane (lq, lr): X[k = lq][slot lr]) serve as it stands
// as a broadcast operand so that 4 of its 16 columns are multiplied in 16 cycles instead of 64?
// build: hipcc --offload-arch=gfx950 -O2 -o mfma4_probe mfma4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CBSZ, int ABID, bool ONEHOT_A>
__global__ void k(double* out) {
    const int lane = threadIdx.x, b = blockIdx.x;
    const double hot = lane == b ? 1.0 : 0.0;
    const double a = ONEHOT_A ? hot : 1.0, bb = ONEHOT_A ? 1.0 : hot;
    out[b * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, 0.0, CBSZ, ABID, 0);
}
template <int CBSZ, int ABID, bool ONEHOT_A>
void run(double* dev) {
    static double h[64 * 64];
    hipLaunchKernelGGL((k<CBSZ, ABID, ONEHOT_A>), dim3(64), dim3(64), 0, 0, dev);
    hipMemcpy(h, dev, sizeof h, hipMemcpyDeviceToHost);
    printf("cbsz=%d abid=%d one-hot %s: sources of output lanes 0, 1, 4, 5, 16, 21, 63:", CBSZ, ABID, ONEHOT_A ? "src0" : "src1");
    const int show[7] = {0, 1, 4, 5, 16, 21, 63};
    for (int q = 0; q < 7; ++q) {
        printf("  [%d:", show[q]);
        for (int b = 0; b < 64; ++b) if (h[b * 64 + show[q]] != 0.0) printf(" %d", b);
        printf("]");
    }
    printf("\n");
}
int main() {
    double* dev; hipMalloc(&dev, 64 * 64 * sizeof(double));
    run<0, 0, true>(dev); run<0, 0, false>(dev);
    run<1, 0, true>(dev); run<1, 1, true>(dev);
    run<2, 0, true>(dev); run<2, 1, true>(dev); run<2, 2, true>(dev); run<2, 3, true>(dev);
    run<2, 1, false>(dev);
    return 0;
}
