// ds_tiles.h -- the kernels that are instantiated per jet-slot tile count ST = P / 16 (k_jet_gemm layer / orbital epilogues,
// k_shared_term, k_layer1_lr, k_layer0_stats), reached through a table of launchers so that the instances can be compiled in
// several translation units (ds_tiles_inst.hip, one per slot-tile range and element type) next to the host code (ds_api.hip).
//   ST = 1 .. 25: N <= 128 electrons (matrices up to 64 x 64 per spin).  Wave tile = 16 NB features x 16 ST slots:
//   NB = 4 for ST <= 5, 2 for ST <= 10, beyond that 1 (float64) / 2 (float32) with four waves per workgroup and one workgroup per CU.
#pragma once
#include "ds_gemm.h"
#include "ds_wide.h"
#include "ds_ldsb.h"

namespace ds {

template <typename T> struct GemmArgs {          // the arguments of k_jet_gemm
    const T* X; size_t xws, xts; const T* W; int K; const T* X2; size_t x2ws; const T* W2; int K2; int n_tiles;
    T* Z; size_t zws, zts; int Nout, P; const T* Sb; const T* bias; OrbEpi<T> oe;
};

template <typename T> struct TileOps {
    int NB, ST;
    // k_jet_gemm<T, NB, ST, epi> for epi = 1, 2, 5, 6, 9 (dynamic LDS of the residual stash added for epi 2)
    void (*gemm)(int epi, dim3 grid, dim3 block, hipStream_t st, const GemmArgs<T>& a);
    // k_jet_gemm<T, 3, ST, 5>: 48-column waves of the orbital head (ST <= 5), or null
    void (*gemm_orb3)(dim3 grid, dim3 block, hipStream_t st, const GemmArgs<T>& a);
    void (*shared_term)(dim3 grid, dim3 block, size_t lds, hipStream_t st, const SysDev<T>& S, const T* G, const T* Wsh, int Kh, T* Sb,
                        int Nout, int P, const T* bias, int bias_all_slots);
    // k_layer1_lr<T, NB, ST, nc, res> (nc = 2, 3, 4)
    void (*layer1_lr)(int nc, bool res, dim3 grid, dim3 block, hipStream_t st, const LrArgs<T>& a);
    // k_layer0_stats<T, ST, nks> (ST <= 5, nks = 2, 3, 4): returns false when there is no such instance; or null
    bool (*layer0_stats)(int nks, dim3 grid, hipStream_t st, const SysDev<T>& S, const T* XL, size_t xl_ws, size_t xl_ts, const T* W0,
                         const T* S0, int Nout, int P, T* YO, T* MEAN1);
    // wide slot ranges (ST > 10; ds_wide.h): the same products with 64-feature x <= 5-tile wave tiles walking the slot range in chunks,
    // two waves per SIMD -- k_jet_gemm_wide<T, STC, epi> for epi = 1, 2, 5, 9 (grid / block for NB = 4: gemm_geom(Nout, 4, .., 5)) and
    // k_layer1_lr_wide<T, STC, nc, res>; null for ST <= 10
    // Both return false (nothing launched) when the chunked form is not the faster one for that kernel and element type -- measured:
    // float64 all but the low-rank layer with three or more column tiles of C (its LDS block then leaves one workgroup per CU);
    // float32 only layer 0 -- unless force is set (DS_WIDE_ALL=1: tests, A/B runs).
    bool (*gemm_wide)(int epi, bool force, dim3 grid, dim3 block, hipStream_t st, const GemmArgs<T>& a);
    bool (*layer1_lr_wide)(int nc, bool res, bool force, dim3 grid, dim3 block, hipStream_t st, const LrArgs<T>& a);
    // float32, ST > 10 (ds_ldsb.h): k_jet_gemm_lb<float, ST, 5, 4>, the orbital head with four 16-feature waves per workgroup and the
    // tile's jet rows staged in LDS; grid = (n_tiles x Nout / 64, walkers), block 256.  Null otherwise.
    void (*orbital_lb)(dim3 grid, hipStream_t st, const GemmArgs<T>& a);
};

constexpr int DS_MAX_TILES = 25;
constexpr int tile_nb(int st, int elem_bytes) { return st <= 5 ? 4 : (st <= 10 ? 2 : (elem_bytes == 4 ? 2 : 1)); }

// the table entry of a slot-tile count, or null (ds_api.hip; the parts are defined in ds_tiles_inst.hip)
template <typename T> const TileOps<T>* tile_ops(int st_tiles);

// k_det_inv_wave<T, NC> (ds_value.h; instances in ds_det_inst.hip: the unrolled elimination takes minutes to compile): inverse +
// log det of the value matrices by in-place Gauss-Jordan with one lane per row.  Returns false when no instance covers n.
template <typename T>
bool launch_det_inv_wave(int n, dim3 grid, hipStream_t st, const SysDev<T>& S, const T* MOUT, size_t mout_stride, size_t mout_off, int ch, int es,
                         T* MINV, size_t minv_stride, size_t minv_off, T* DETS, size_t dets_stride, size_t dets_off, int P);

}  // namespace ds
