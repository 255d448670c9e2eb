// Lane layout of v_mfma_f64_4x4x4_4b_f64 by exhaustion: a = one-hot(la), b = one-hot(lb).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CBSZ, int ABID>
__global__ void k_all(double* c) {   // block (la, lb): one wave
    const int la = blockIdx.x, lb = blockIdx.y, l = threadIdx.x;
    c[((size_t)la * 64 + lb) * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(l == la ? 1.0 : 0.0, l == lb ? 1.0 : 0.0, 0.0, CBSZ, ABID, 0);
}

int main() {
    double* c;
    hipMalloc(&c, 64 * 64 * 64 * 8);
    std::vector<double> h(64 * 64 * 64);
    for (int mode = 0; mode < 5; ++mode) {
        switch (mode) {
            case 0: hipLaunchKernelGGL((k_all<0, 0>), dim3(64, 64), 64, 0, 0, c); break;
            case 1: hipLaunchKernelGGL((k_all<2, 0>), dim3(64, 64), 64, 0, 0, c); break;
            case 2: hipLaunchKernelGGL((k_all<2, 1>), dim3(64, 64), 64, 0, 0, c); break;
            case 3: hipLaunchKernelGGL((k_all<2, 2>), dim3(64, 64), 64, 0, 0, c); break;
            case 4: hipLaunchKernelGGL((k_all<2, 3>), dim3(64, 64), 64, 0, 0, c); break;
        }
        hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
        printf("mode %d (cbsz=%d abid=%d)\n", mode, mode ? 2 : 0, mode ? mode - 1 : 0);
        for (int d = 0; d < 64; ++d) {
            if (mode && d % 16 > 1) continue;
            printf("  D lane %2d <-", d);
            for (int la = 0; la < 64; ++la)
                for (int lb = 0; lb < 64; ++lb)
                    if (h[((size_t)la * 64 + lb) * 64 + d] != 0.0) printf(" (a%d,b%d)", la, lb);
            printf("\n");
        }
    }
    return 0;
}
