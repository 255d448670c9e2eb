// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: lane layout, A-broadcast (cbsz/abid) semantics, issue rate.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/mfma44_probe.hip -o tools/probes/mfma44_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int CBSZ, int ABID>
__global__ void k_one(const double* a, const double* b, double* c) {
    c[threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[threadIdx.x], b[threadIdx.x], 0.0, CBSZ, ABID, 0);
}

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(long iters, double* out) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    const long long c0 = clock64();
    double s = 0;
    if (MODE == 0) {            // 4 x (4x4x4_4b with A broadcast) = one 16x16x4 product; 4 such tiles in flight
        double acc[16];
        for (int j = 0; j < 16; ++j) acc[j] = 0;
        for (long it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[4 * t + 0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[4 * t + 0], 2, 0, 0);
                acc[4 * t + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[4 * t + 1], 2, 1, 0);
                acc[4 * t + 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[4 * t + 2], 2, 2, 0);
                acc[4 * t + 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[4 * t + 3], 2, 3, 0);
            }
        }
        for (int j = 0; j < 16; ++j) s += acc[j];
    } else {                    // 4 x 16x16x4
        typedef double v4 __attribute__((ext_vector_type(4)));
        v4 acc[4];
        for (int j = 0; j < 4; ++j) acc[j] = v4{0, 0, 0, 0};
        for (long it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
        for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    }
    const long long c1 = clock64();
    if (s == 12345.678) out[0] = s;
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) out[1] = (double)(c1 - c0);
}

int main() {
    std::vector<double> ha(64), hb(64), hc(64);
    srand(1);
    for (int i = 0; i < 64; ++i) { ha[i] = (rand() % 1000) / 100.0; hb[i] = (rand() % 1000) / 100.0; }
    double *a, *b, *c;
    hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&c, 512);
    hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
    auto run = [&](int which) {
        switch (which) {
            case 0: hipLaunchKernelGGL((k_one<0, 0>), 1, 64, 0, 0, a, b, c); break;
            case 1: hipLaunchKernelGGL((k_one<2, 0>), 1, 64, 0, 0, a, b, c); break;
            case 2: hipLaunchKernelGGL((k_one<2, 1>), 1, 64, 0, 0, a, b, c); break;
            case 3: hipLaunchKernelGGL((k_one<2, 2>), 1, 64, 0, 0, a, b, c); break;
            case 4: hipLaunchKernelGGL((k_one<2, 3>), 1, 64, 0, 0, a, b, c); break;
        }
        hipMemcpy(hc.data(), c, 512, hipMemcpyDeviceToHost);
    };
    // hypotheses: lane = 16*blk + 4*x + y
    //   A: (x,y) = (k,i) or (i,k);  B: (k,j) or (j,k);  D: (i,j) or (j,i)
    run(0);
    for (int ha_ = 0; ha_ < 2; ++ha_) for (int hb_ = 0; hb_ < 2; ++hb_) for (int hd = 0; hd < 2; ++hd) {
        double err = 0;
        for (int blk = 0; blk < 4; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) {
                const int la = 16 * blk + (ha_ ? 4 * i + k : 4 * k + i), lb = 16 * blk + (hb_ ? 4 * j + k : 4 * k + j);
                s += ha[la] * hb[lb];
            }
            const int ld = 16 * blk + (hd ? 4 * j + i : 4 * i + j);
            err += (s - hc[ld]) * (s - hc[ld]);
        }
        printf("layout A:%s B:%s D:%s  err %.3e\n", ha_ ? "4i+k" : "4k+i", hb_ ? "4j+k" : "4k+j", hd ? "4j+i" : "4i+j", err);
    }
    // broadcast: which block's A feeds all blocks for (cbsz=2, abid=t)?  use the best layout found per print above
    for (int t = 0; t < 4; ++t) {
        run(1 + t);
        for (int src = 0; src < 4; ++src)
            for (int ha_ = 0; ha_ < 2; ++ha_) for (int hb_ = 0; hb_ < 2; ++hb_) for (int hd = 0; hd < 2; ++hd) {
                double err = 0;
                for (int blk = 0; blk < 4; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
                    double s = 0;
                    for (int k = 0; k < 4; ++k) {
                        const int la = 16 * src + (ha_ ? 4 * i + k : 4 * k + i), lb = 16 * blk + (hb_ ? 4 * j + k : 4 * k + j);
                        s += ha[la] * hb[lb];
                    }
                    const int ld = 16 * blk + (hd ? 4 * j + i : 4 * i + j);
                    err += (s - hc[ld]) * (s - hc[ld]);
                }
                if (err < 1e-18) printf("cbsz=2 abid=%d: A of block %d broadcast (A:%s B:%s D:%s)\n", t, src, ha_ ? "4i+k" : "4k+i",
                                        hb_ ? "4j+k" : "4k+j", hd ? "4j+i" : "4i+j");
            }
    }
    // issue rate
    double* out;
    hipMalloc(&out, 128);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const long iters = 400000 / wps;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL((k_rate<0>), 256 * wps, 256, 0, 0, iters, out);
                else hipLaunchKernelGGL((k_rate<1>), 256 * wps, 256, 0, 0, iters, out);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            const double flops = 256.0 * wps * 4 * iters * 4 * 2048;      // 4 tile-products of 16x16x4 per iteration per wave
            printf("%s  %d waves/SIMD: %.2f TFLOP/s, %.1f cycles per 16x16x4-equivalent per SIMD\n", mode == 0 ? "4x(4x4x4_4b)" : "16x16x4     ",
                   wps, flops / (ms * 1e-3) / 1e12, h[1] / (iters * 4.0 * wps));
        }
    return 0;
}
