// Development harness of the two-workgroup int8 layer (csrc/ds_i8s.h): both int8 layer kernels behind a C ABI of their own, so
// that a variant compiles in seconds.  tools/i8s_probe.py drives it (random jets, old kernel as the reference, HIP-event timing).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared tools/probes/i8s_probe.hip -o tools/probes/libi8sprobe.so [-DI8S_...]
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "ds_i8s.h"

extern "C" {
void i8s_prep_w(const double* W, int K, int ldw, void* WP, double* SW, hipStream_t st) {
    hipLaunchKernelGGL(ds::i8::k_i8_prep_w, dim3(ds::i8::NOUT / 16), dim3(256), 0, st, W, K, ldw, (uint8_t*)WP, SW);
}
void i8s_old(const double* X, size_t ts, const void* WP, const double* SW, const double* Sb, int N, double* G, int ntiles, int ncu, hipStream_t st) {
    hipLaunchKernelGGL((ds::i8::k_layer_i8<5, 2>), dim3(ntiles < ncu ? ntiles : ncu), dim3(512), ds::i8::lds_bytes(), st, X, ts, (const uint4*)WP, SW, Sb, N, G, ntiles);
}
void i8s_new(const double* X, size_t ts, const void* WP, const double* SW, const double* Sb, int N, double* G, int ntiles, int ncu, hipStream_t st) {
    const char* e = getenv("I8S_GRID_MULT");
    const int mult = e ? atoi(e) : 2;
    hipLaunchKernelGGL(ds::i8::k_layer_i8_split, dim3(ntiles < mult * ncu ? ntiles : mult * ncu), dim3(256), ds::i8::lds_bytes_split(), st, X, ts, (const uint4*)WP, SW, Sb, N, G, ntiles);
}
long i8s_wp_bytes() { return (long)ds::i8::wp_bytes(320); }
}
