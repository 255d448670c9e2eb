// ds_ldsb.h -- the per-electron contraction of the wide float32 cells (more than 10 jet-slot tiles: diamond 2x2x2, 19 tiles) with the
// jet rows staged in LDS.
//
// k_jet_gemm streams both operands global -> register per wave.  With 19 slot tiles a 32-feature wave holds 152 accumulators, runs
// alone on its SIMD and puts 21 operand loads between the 38 MFMAs of a k-step (0.62 of the float32 matrix peak on the orbital head
// of BASELINE config 5; 16-feature waves at two per SIMD WITHOUT staging are slower still: one load per MFMA, EXPERIMENTS.md).  Here a
// workgroup of four 16-feature waves (76 accumulators each, 126 registers: four workgroups per CU) shares the B operand -- the tile's jet rows,
// the same for every feature block: 16 rows (four k-steps) at a time go global -> registers -> LDS, double-buffered, one barrier per
// group; a wave reads its B values from LDS (conflict-free ds_read_b32: 16 consecutive slots per lane group) and only its own weight
// column from memory (one load per k-step, a group ahead).  Epilogue: the orbital head (EPI 5) of ds_gemm.h on a one-block
// accumulator tile.  Measured on config 5 (1024 walkers): orbital head 112.7 -> 102.2 ms (0.63 -> 0.70 of the float32 matrix peak).
// The hidden layers (EPI 2) were tried in the same form and lost (44.9 -> 48.6 ms: 152 registers, the shared-term read up front and
// the residual read from memory instead of the stash); they stay on k_jet_gemm.  NW = waves per workgroup: 8 (half the row traffic
// per product) measured no faster than 4 (105.9 vs 103.5 ms) and is not instantiated.  EXPERIMENTS.md, round 5.
#pragma once
#include "ds_gemm.h"

namespace ds {

template <typename T> inline size_t ldsb_bytes(int P) { return (size_t)2 * 16 * P * sizeof(T); }

// grid.x = n_tiles * (Nout / (16 NW)) (feature blocks folded into x: the workgroups sharing an electron tile run side by side), grid.y =
// walkers; block = 64 NW; dynamic LDS = ldsb_bytes(P).  K % 16 == 0, Nout % (16 NW) == 0.
//   X : [walker][tile][rows][P]   W : [K][Nout]
//   EPI 5: Sb = shared term of the orbital head with use_last_layer, or null; oe = the orbital epilogue's arguments
template <typename T, int ST, int EPI, int NW>
__global__ void __launch_bounds__(64 * NW, 2)
k_jet_gemm_lb(const T* __restrict__ X, size_t x_walker_stride, size_t x_tile_stride, const T* __restrict__ W, int K, int n_tiles,
              T* __restrict__ Z, size_t z_walker_stride, size_t z_tile_stride, int Nout, int P, const T* __restrict__ Sb, OrbEpi<T> oe) {
    typedef typename Acc4<T>::type acc_t;
    static_assert(sizeof(T) == 4, "float32 instances only (float64 wide cells run the chunked kernels of ds_wide.h)");
    static_assert(EPI == 5, "orbital head only");
    // XCD-aware placement as in k_jet_gemm: workgroup ids b, b + 8, ... share an L2; all feature blocks and tiles of a walker get ids
    // of one residue class
    int tile = blockIdx.x, w = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const unsigned b = blockIdx.y * gridDim.x + blockIdx.x, q = b >> 3;
        w = (q / gridDim.x) * 8 + (b & 7);
        tile = q % gridDim.x;
    }
    const int gzf = gridDim.x / n_tiles, zb = tile % gzf;
    tile /= gzf;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int n0 = (zb * NW + wave) * 16;
    extern __shared__ __attribute__((aligned(16))) char ldsb_smem[];
    T* lb = reinterpret_cast<T*>(ldsb_smem);                     // [2][16][P]
    const T* Xp = X + (size_t)w * x_walker_stride + (size_t)tile * x_tile_stride;
    const int ng = K / 16, gsz = 16 * P;                         // groups of 16 rows; elements per group
    const int npc = gsz / 4;                                     // 16-byte pieces per group (P is a multiple of 16)
    constexpr int MAXPC = (ST + NW - 1) / NW;                    // pieces per thread: 16 rows x 16 ST slots x 4 bytes / 16 / (64 NW threads); P = 16 ST
    acc_t acc[1][ST];
#pragma unroll
    for (int s = 0; s < ST; ++s) acc[0][s] = acc_t{0, 0, 0, 0};
    typedef float vec4 __attribute__((ext_vector_type(4)));
    vec4 stg[MAXPC];
    T an[4], ac[4];
    auto fetch = [&](int g) {                                    // group g: its rows -> registers, this wave's weight column -> an
        const vec4* src = reinterpret_cast<const vec4*>(Xp + (size_t)g * gsz);
#pragma unroll
        for (int u = 0; u < MAXPC; ++u) {
            const int pc = threadIdx.x + 64 * NW * u;
            if (pc < npc) stg[u] = src[pc];
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) an[ks] = W[(size_t)(16 * g + 4 * ks + lq) * Nout + n0 + lr];
    };
    auto park = [&](int g) {
        vec4* dst = reinterpret_cast<vec4*>(lb + (size_t)(g & 1) * gsz);
#pragma unroll
        for (int u = 0; u < MAXPC; ++u) {
            const int pc = threadIdx.x + 64 * NW * u;
            if (pc < npc) dst[pc] = stg[u];
        }
    };
    fetch(0);
    park(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ac[ks] = an[ks];
    __syncthreads();
#pragma unroll 1
    for (int g = 0; g < ng; ++g) {
        if (g + 1 < ng) fetch(g + 1);
        const T* bp = lb + (size_t)(g & 1) * gsz + lq * P + lr;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            T bv[ST];
#pragma unroll
            for (int s = 0; s < ST; ++s) bv[s] = bp[(4 * ks) * P + 16 * s];
#pragma unroll
            for (int s = 0; s < ST; ++s) acc[0][s] = mfma16(ac[ks], bv[s], acc[0][s]);
        }
        if (g + 1 < ng) {
            park(g + 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ac[ks] = an[ks];
        }
        __syncthreads();
    }
    orbital_epilogue<T, 1, ST>(acc, oe, tile, w, n0, lane, Sb, Nout, P);
}

}  // namespace ds
