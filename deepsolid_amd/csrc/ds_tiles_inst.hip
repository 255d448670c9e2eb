// ds_tiles_inst.hip -- one part of the per-slot-tile-count kernel instances (ds_tiles.h).  Compiled once per (DS_PART, DS_F32):
//   DS_PART 0: ST 1..5   1: ST 6..10   2: ST 11..15   3: ST 16..20   4: ST 21..25        DS_F32 0: double   1: float
#include <hip/hip_runtime.h>

#include "ds_tiles.h"

#if DS_F32
typedef float DS_T;
#else
typedef double DS_T;
#endif

#if DS_PART == 0
#define DS_ST_LIST(X) X(1) X(2) X(3) X(4) X(5)
#elif DS_PART == 1
#define DS_ST_LIST(X) X(6) X(7) X(8) X(9) X(10)
#elif DS_PART == 2
#define DS_ST_LIST(X) X(11) X(12) X(13) X(14) X(15)
#elif DS_PART == 3
#define DS_ST_LIST(X) X(16) X(17) X(18) X(19) X(20)
#elif DS_PART == 4
#define DS_ST_LIST(X) X(21) X(22) X(23) X(24) X(25)
#else
#error "DS_PART must be 0..4"
#endif

namespace ds {

template <typename T, int NB, int ST> struct TileImpl {
    static void gemm(int epi, dim3 grid, dim3 block, hipStream_t st, const GemmArgs<T>& a) {
#define DS_G(E, LDS) hipLaunchKernelGGL((k_jet_gemm<T, NB, ST, E>), grid, block, (LDS), st, a.X, a.xws, a.xts, a.W, a.K, a.X2, a.x2ws, a.W2, a.K2, \
                                        a.n_tiles, a.Z, a.zws, a.zts, a.Nout, a.P, a.Sb, a.bias, a.oe)
        switch (epi) {
            case 1: DS_G(1, 0); break;
            case 2:
                if constexpr (NB == 4 && ST == 5 && sizeof(T) == 8) {
                    // (the last slot tile as three groups of four columns: 74 jets of the 24-electron cells)
                    if (a.oe.g4 == 3) {
                        hipLaunchKernelGGL((k_jet_gemm<T, NB, ST, 2, 3>), grid, block, (gemm_stash_bytes<T, NB, ST>(block.x)), st, a.X, a.xws, a.xts, a.W, a.K, a.X2, a.x2ws,
                                           a.W2, a.K2, a.n_tiles, a.Z, a.zws, a.zts, a.Nout, a.P, a.Sb, a.bias, a.oe);
                        break;
                    }
                }
                if constexpr (NB == 2 && ST == 10 && sizeof(T) == 8) {
                    // (48 electrons: one group of four columns on the last slot tile, staged in the stash's padding slots like the <4, 5> instance)
                    if (a.oe.g4 == 1) {
                        hipLaunchKernelGGL((k_jet_gemm<T, NB, ST, 2, 1>), grid, block, (gemm_stash_bytes<T, NB, ST>(block.x)), st, a.X, a.xws, a.xts, a.W, a.K, a.X2, a.x2ws, a.W2, a.K2,
                                           a.n_tiles, a.Z, a.zws, a.zts, a.Nout, a.P, a.Sb, a.bias, a.oe);
                        break;
                    }
                }
                DS_G(2, (gemm_stash_bytes<T, NB, ST>(block.x)));
                break;
            case 5:
                if constexpr (ST == 10 && sizeof(T) == 8) {
                    // (48 electrons: 146 jets, two of them on the last slot tile -- one group of four columns, re-laid through 512 bytes of LDS per wave)
                    if (a.oe.g4 == 1) {
                        hipLaunchKernelGGL((k_jet_gemm<T, NB, ST, 5, 1>), grid, block, (block.x / 64) * 64 * sizeof(T), st, a.X, a.xws, a.xts, a.W, a.K, a.X2, a.x2ws, a.W2, a.K2,
                                           a.n_tiles, a.Z, a.zws, a.zts, a.Nout, a.P, a.Sb, a.bias, a.oe);
                        break;
                    }
                }
                DS_G(5, 0);
                break;
            case 6: DS_G(6, 0); break;
            case 9: DS_G(9, 0); break;
            default: break;
        }
#undef DS_G
    }
    static void gemm_orb3(dim3 grid, dim3 block, hipStream_t st, const GemmArgs<T>& a) {
        if constexpr (ST == 5 && sizeof(T) == 8) {
            // (the last slot tile as three groups of four columns, re-laid through 512 bytes of LDS per wave)
            if (a.oe.g4 == 3) {
                hipLaunchKernelGGL((k_jet_gemm<T, 3, ST, 5, 3>), grid, block, (block.x / 64) * 64 * sizeof(T), st, a.X, a.xws, a.xts, a.W, a.K, a.X2, a.x2ws, a.W2, a.K2,
                                   a.n_tiles, a.Z, a.zws, a.zts, a.Nout, a.P, a.Sb, a.bias, a.oe);
                return;
            }
        }
        if constexpr (ST <= 5)
            hipLaunchKernelGGL((k_jet_gemm<T, 3, ST, 5>), grid, block, 0, st, a.X, a.xws, a.xts, a.W, a.K, a.X2, a.x2ws, a.W2, a.K2, a.n_tiles, a.Z,
                               a.zws, a.zts, a.Nout, a.P, a.Sb, a.bias, a.oe);
    }
    static void shared_term(dim3 grid, dim3 block, size_t lds, hipStream_t st, const SysDev<T>& S, const T* G, const T* Wsh, int Kh, T* Sb,
                            int Nout, int P, const T* bias, int bias_all_slots) {
        hipLaunchKernelGGL((k_shared_term<T, NB, ST>), grid, block, lds, st, S, G, Wsh, Kh, Sb, Nout, P, bias, bias_all_slots);
    }
    static void layer1_lr(int nc, bool res, dim3 grid, dim3 block, hipStream_t st, const LrArgs<T>& a) {
        // float64, up to 10 slot tiles: a last column tile with 4 or 8 columns in use (K0 + 4 = 16 (nc - 1) + 4 or + 8) as one or two
        // four-block MFMA groups
        if constexpr (sizeof(T) == 8 && ST <= 10) {
            const int rem = a.K0loc + a.K0sh + 4 - 16 * (nc - 1);
            if (nc >= 2 && nc <= 3 && (rem == 4 || rem == 8)) {
#define DS_LRG(NCV, NGV, RESV) hipLaunchKernelGGL((k_layer1_lr<T, NB, ST, NCV, RESV, NGV>), grid, block, (lr_lds_bytes<T, NB, NCV, NGV>(block.x, a.Kh)), st, a)
                if (nc == 2) { if (rem == 4) { if (res) DS_LRG(1, 1, true); else DS_LRG(1, 1, false); } else { if (res) DS_LRG(1, 2, true); else DS_LRG(1, 2, false); } }
                else { if (rem == 4) { if (res) DS_LRG(2, 1, true); else DS_LRG(2, 1, false); } else { if (res) DS_LRG(2, 2, true); else DS_LRG(2, 2, false); } }
#undef DS_LRG
                return;
            }
        }
#define DS_LR(NCV, RESV) hipLaunchKernelGGL((k_layer1_lr<T, NB, ST, NCV, RESV>), grid, block, (lr_lds_bytes<T, NB, NCV>(block.x, a.Kh)), st, a)
        if (nc <= 2) { if (res) DS_LR(2, true); else DS_LR(2, false); }
        else if (nc == 3) { if (res) DS_LR(3, true); else DS_LR(3, false); }
        else { if (res) DS_LR(4, true); else DS_LR(4, false); }
#undef DS_LR
    }
    static bool layer0_stats(int nks, dim3 grid, hipStream_t st, const SysDev<T>& S, const T* XL, size_t xl_ws, size_t xl_ts, const T* W0,
                             const T* S0, int Nout, int P, T* YO, T* MEAN1) {
        if constexpr (ST <= 5) {
#define DS_L0S(NKSV) hipLaunchKernelGGL((k_layer0_stats<T, ST, NKSV>), grid, dim3(256), 0, st, S, XL, xl_ws, xl_ts, W0, S0, Nout, P, YO, MEAN1)
            if (nks == 2) { DS_L0S(2); return true; }
            if (nks == 3) { DS_L0S(3); return true; }
            if (nks == 4) { DS_L0S(4); return true; }
#undef DS_L0S
        }
        return false;
    }
    // chunk width of the wide kernels: 5 slot tiles in float32 (7 for the orbital head), 4 in float64 (five spill there)
    static constexpr int STCW = sizeof(T) == 4 ? 5 : 4, STCO = sizeof(T) == 4 ? 7 : 4;
    static bool gemm_wide(int epi, bool force, dim3 grid, dim3 block, hipStream_t st, const GemmArgs<T>& a) {
        if constexpr (ST > 10) {
            if (!force && sizeof(T) == 4 && epi != 1 && epi != 9) return false;
#define DS_GW(STCV, E, LDS) hipLaunchKernelGGL((k_jet_gemm_wide<T, STCV, E>), grid, block, (LDS), st, a.X, a.xws, a.xts, a.W, a.K, a.n_tiles, a.Z, a.zws, a.zts, \
                                               a.Nout, a.P, a.Sb, a.oe)
            switch (epi) {
                case 1: DS_GW(STCW, 1, 0); return true;
                case 2: DS_GW(STCW, 2, (wide_stash_bytes<T, STCW>(block.x))); return true;
                case 5: DS_GW(STCO, 5, 0); return true;
                case 9: DS_GW(STCW, 9, 0); return true;
                default: return false;
            }
#undef DS_GW
        }
        return false;
    }
    static bool layer1_lr_wide(int nc, bool res, bool force, dim3 grid, dim3 block, hipStream_t st, const LrArgs<T>& a) {
        if constexpr (ST > 10) {
            if (!force && (sizeof(T) == 4 || nc > 2)) return false;
#define DS_LRW(NCV, RESV) hipLaunchKernelGGL((k_layer1_lr_wide<T, STCW, NCV, RESV>), grid, block, (lr_lds_bytes<T, 4, NCV>(block.x, a.Kh)), st, a)
            if (nc <= 2) { if (res) DS_LRW(2, true); else DS_LRW(2, false); }
            else if (nc == 3) { if (res) DS_LRW(3, true); else DS_LRW(3, false); }
            else { if (res) DS_LRW(4, true); else DS_LRW(4, false); }
#undef DS_LRW
            return true;
        }
        return false;
    }
    static void orbital_lb(dim3 grid, hipStream_t st, const GemmArgs<T>& a) {
        if constexpr (ST > 10 && sizeof(T) == 4)
            hipLaunchKernelGGL((k_jet_gemm_lb<T, ST, 5, 4>), grid, dim3(256), ldsb_bytes<T>(a.P), st, a.X, a.xws, a.xts, a.W, a.K, a.n_tiles, a.Z, a.zws,
                               a.zts, a.Nout, a.P, a.Sb, a.oe);
    }
    static const TileOps<T>* ops() {
        static const TileOps<T> o = {NB, ST, &gemm, (ST <= 5 ? &gemm_orb3 : nullptr), &shared_term, &layer1_lr, (ST <= 5 ? &layer0_stats : nullptr),
                                      (ST > 10 ? &gemm_wide : nullptr), (ST > 10 ? &layer1_lr_wide : nullptr),
                                      ((ST > 10 && sizeof(T) == 4) ? &orbital_lb : nullptr)};
        return &o;
    }
};

#define DS_CAT2(a, b, c) a##b##c
#define DS_CAT(a, b, c) DS_CAT2(a, b, c)
#if DS_F32
#define DS_PART_FN DS_CAT(tile_ops_f32_p, DS_PART, )
#else
#define DS_PART_FN DS_CAT(tile_ops_f64_p, DS_PART, )
#endif

const TileOps<DS_T>* DS_PART_FN(int st_tiles) {
    switch (st_tiles) {
#define DS_CASE(STV) case STV: return TileImpl<DS_T, tile_nb(STV, (int)sizeof(DS_T)), STV>::ops();
        DS_ST_LIST(DS_CASE)
#undef DS_CASE
        default: return nullptr;
    }
}

}  // namespace ds
