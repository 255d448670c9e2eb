// ds_det_inst.hip -- instances of k_det_inv_wave (ds_value.h) behind launch_det_inv_wave (ds_tiles.h): float64 up to 24 x 24
// (the 24- and 48-electron cells), float32 up to 48 x 48 (the 96-electron cell); other sizes use the LDS Gauss-Jordan kernel.
#include <hip/hip_runtime.h>

#include "ds_value.h"
#include "ds_tiles.h"

namespace ds {

#define DS_GJW(NCV) hipLaunchKernelGGL((k_det_inv_wave<T, NCV>), grid, dim3(64), 0, st, S, MOUT, mout_stride, mout_off, ch, es, MINV, minv_stride, minv_off, \
                                       DETS, dets_stride, dets_off, P)
template <typename T>
bool launch_det_inv_wave(int n, dim3 grid, hipStream_t st, const SysDev<T>& S, const T* MOUT, size_t mout_stride, size_t mout_off, int ch, int es,
                         T* MINV, size_t minv_stride, size_t minv_off, T* DETS, size_t dets_stride, size_t dets_off, int P) {
    if (n <= 12) { DS_GJW(12); return true; }
    if (n <= 24) { DS_GJW(24); return true; }
    if constexpr (sizeof(T) == 4) {
        if (n <= 48) { DS_GJW(48); return true; }
    }
    return false;
}
#undef DS_GJW
template bool launch_det_inv_wave<double>(int, dim3, hipStream_t, const SysDev<double>&, const double*, size_t, size_t, int, int, double*, size_t, size_t,
                                          double*, size_t, size_t, int);
template bool launch_det_inv_wave<float>(int, dim3, hipStream_t, const SysDev<float>&, const float*, size_t, size_t, int, int, float*, size_t, size_t,
                                         float*, size_t, size_t, int);

}  // namespace ds
