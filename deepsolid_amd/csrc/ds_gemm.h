// ds_gemm.h -- the dense contraction of the chain and its element-wise epilogues.
//
//   k_jet_gemm          Z[tile][n][slot] = sum_k W[k][n] * X[tile][k][slot]      (fp64/fp32 MFMA)
//                       with optional fused epilogues: one-electron layer (tanh chain rule, residual)
//                       or orbital head (envelope x phase product rule)
//   k_shared_term       the per-walker spin-mean term of a hidden layer, means formed on the fly
//
// The GEMM main loop needs only its accumulators and two k-steps of operands in registers, which
// keeps two waves per SIMD resident: one wave's loads / epilogue hide behind the other's MFMAs.
#pragma once
#include <type_traits>

#include "ds_kernels.h"

// The dense 24-electron hidden-layer instance adds the shared term S in its epilogue (layer_epilogue_sadd); compile-time switches
// of that variant, kept for A/B builds (-DDS_SADD=0: accumulators start at S as in every other instance)
#ifndef DS_SADD
#define DS_SADD 1
#endif
#ifndef DS_SADD_NSF
#define DS_SADD_NSF 2
#endif

namespace ds {

// Orbital-head weights are packed so that the MFMA accumulator hands one lane the pairs (Re, Im) of
// two orbitals: in every 16-column tile t, lane group q = lane >> 4 owns accumulator registers
// r = 0..3 = (Re p, Im p, Re p', Im p') with p = 8t + q, p' = p + 4.  Column of (p, part):
template <typename T> __device__ __forceinline__ int orb_col(int p, int part) {
    const int t = p >> 3, q = p & 3, r = 2 * ((p >> 2) & 1) + part;
    return 16 * t + acc_row<T>(q << 4, r);
}

// fused orbital epilogue arguments (EPI = 5): M = phi * q with the product rule on the jets
template <typename T> struct OrbEpi {
    const T* Q;            // [walker][electron][nparam_max][10]
    T* MOUT;               // [walker][channel][det][slot tile][elec][orb][re,im][16]
    size_t mout_stride, mout_off;   // walker stride and offset of this spin's determinant channel
    int N, i0, nparam, nparam_max;
    int norb, n, row0;              // orbitals per det, matrix size, first matrix row of this spin's electrons
    const T* bias;                  // optional orbital bias (2*nparam: Re then Im), value slot only; or null
    // optional in-kernel clock probe (hidden layers, EPI = 2, while profiling): wave 0 of every workgroup adds its shader-clock
    // cycles (s_memtime) to clk[0] and its constant-rate 100 MHz ticks (s_memrealtime) to clk[1]
    unsigned long long* clk;
    int dbg;                        // DS_DBG & 32 (kernel development): one wave also writes phase stamps to clk[2 + i]
    // dense hidden layer (EPI 2, the 24-electron float64 instance): the pair-mean rows -- k-steps pm_k0 .. of the contraction, pm_ks per
    // partner spin -- are exactly zero outside slot tile 0, the electron's own tile(s) and the tiles of the partners' slots
    // (k_m2_expand writes zeros there): the products on the other tiles are skipped.  pm_ks = 0: no such rows / no skipping
    int pm_k0, pm_ks, pm_nup, pm_nch;
    // (round 6) > 0: the last slot tile holds at most 4 g4 jets -- the dense 24-electron instance multiplies it as g4 groups of 4 columns
    // (v_mfma_f64_4x4x4, 17 cycles each) instead of one 16-column tile (64 cycles): k_jet_gemm<.., 2, G4>
    int g4;
};

// DPP move inside the 16-lane rows under a bank mask (BANK bit q = quad q of every row is written)
template <int CTRL, int BANK> __device__ __forceinline__ int dpp_mov(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xF, BANK, false); }
// DPP control that brings quad SRC of a row to quad DST: row_shr:4(DST - SRC) / row_shl:4(SRC - DST); the identity quad_perm for SRC == DST
template <int SRC, int DST> constexpr int quad_ctrl() { return DST == SRC ? 0xE4 : (DST > SRC ? 0x110 + 4 * (DST - SRC) : 0x100 + 4 * (SRC - DST)); }
// Element r of a 16x16 accumulator tile (feature lq + 4 r, column lr) from the results c[g] of v_mfma_f64_4x4x4 on column group g (lane
// (lq, lr): feature 4 (lr >> 2) + lq, column 4 g + (lr & 3)): quad g of the row takes c[g] from quad r; columns beyond 4 NG are zero.
template <int NG, int R> __device__ __forceinline__ double quads_to_tile(const double (&c)[NG]) {
    int hi = 0, lo = 0;
    if constexpr (NG > 0) { hi = dpp_mov<quad_ctrl<R, 0>(), 1>(hi, __double2hiint(c[0])); lo = dpp_mov<quad_ctrl<R, 0>(), 1>(lo, __double2loint(c[0])); }
    if constexpr (NG > 1) { hi = dpp_mov<quad_ctrl<R, 1>(), 2>(hi, __double2hiint(c[1])); lo = dpp_mov<quad_ctrl<R, 1>(), 2>(lo, __double2loint(c[1])); }
    if constexpr (NG > 2) { hi = dpp_mov<quad_ctrl<R, 2>(), 4>(hi, __double2hiint(c[2])); lo = dpp_mov<quad_ctrl<R, 2>(), 4>(lo, __double2loint(c[2])); }
    if constexpr (NG > 3) { hi = dpp_mov<quad_ctrl<R, 3>(), 8>(hi, __double2hiint(c[3])); lo = dpp_mov<quad_ctrl<R, 3>(), 8>(lo, __double2loint(c[3])); }
    return __hiloint2double(hi, lo);
}

// Residual stash (EPI = 2 / 4): the residual rows of a layer are rows n0..n0+16*NB-1 of the SAME tile the wave streams as
// its B operand, and the lane that needs X[n][slot] in the epilogue is a lane that held it in its operand registers at
// k-step n / 4.  The first stash_blocks() 16-row blocks are parked in LDS on the way (one ds_write per operand, no extra
// global traffic); only the remaining blocks are re-read from memory in the epilogue.  Budget: half of the CU's 160 KB
// when two workgroups share a CU (ST <= 10), all of it otherwise.
template <typename T, int NB, int ST> constexpr int stash_blocks() {
    constexpr int waves = (NB == 3 || ST > 5) ? 4 : 16 / NB;
    constexpr int budget = (ST <= 10 ? 80 : 160) * 1024;                      // two workgroups per CU unless ST > 10 (one)
    constexpr int per_block = waves * 4 * ST * 64 * (int)sizeof(T);
    return budget / per_block > NB ? NB : budget / per_block;
}
template <typename T, int NB, int ST> inline size_t gemm_stash_bytes(unsigned threads) {
    return (size_t)(threads / 64) * stash_blocks<T, NB, ST>() * 4 * ST * 64 * sizeof(T);
}

// Residual-layer instances (k_jet_gemm<T, NB, ST, 2>) that skip the structurally zero slot tiles of the pair-mean rows: float64 up to
// 20 slot tiles (beyond, the masked rounds spill -- cells that run the chunked kernels of ds_wide.h anyway).  ds_api.hip asks the same
// question before it lets k_m2_expand leave those tiles unwritten.
// float64 only: a float32 MFMA lasts 32 cycles, a mask branch then guards too little work (diamond f32, 19 tiles: hidden layers
// 45.2 -> 48.1 ms WITH the masks); a float64 branch guards 128-256 cycles of products.
template <typename T> constexpr bool pm_instance(int st) { return sizeof(T) == 8 && st <= 20; }

// Depth of the operand ring: as many k-steps of operands as the register file leaves next to the accumulators (256 VGPRs per
// wave with two waves per SIMD, 512 for the four-wave workgroups of the widest tiles), at most four, at least two.
template <typename T, int NB, int ST> constexpr int ring_sets() {
    constexpr int vg = (int)sizeof(T) / 4, limit = ST > 10 ? 512 : 256;
    constexpr int n = (limit - NB * ST * 4 * vg - 24) / ((NB + ST) * vg);
    return n > 4 ? 4 : (n < 2 ? 2 : n);
}

// The hidden-layer / orbital / plain-product instantiations run the four-set operand ring and need K % 16 == 0 (true for
// every K they are launched with: hidden widths are multiples of 64, pair widths 16 or 32); layer 0 and the shared term of
// layer 0 (K = 12, 8, ...) use the plain loop.
__host__ __device__ constexpr bool gemm_uses_ring(int epi) { return epi == 0 || epi == 2 || epi == 4 || epi == 5 || epi == 8; }

// Fused one-electron-layer epilogue (network.py:524-528) on a wave's accumulator tile: rows n0 + 16 a + acc_row(lane, r),
// slots 16 s + lr.  EPI 1 / 2: tanh chain rule on the jets without / with the residual; EPI 3 / 4: plain tanh (value chain).
//   Gi: residual rows of the tile (+ lr), row stride P;  Go: output tile (+ lr), row stride P;  stash: the wave's LDS copy of the
//   first NA 16-row blocks of its residual rows (stash_blocks), the others are re-read from Gi.
//   rf: optional residual + store stage instead of Gi / stash / Go: the epilogue leaves the pre-residual output of 16-row block a
//   in acc[a] and calls rf(a), which adds the residual rows and stores (k_layer1_lr recomputes those rows block by block)
struct NoResidFn {};
template <typename T, int NB, int ST, int EPI, int NA, typename RF = NoResidFn>
__device__ __forceinline__ void layer_epilogue(typename Acc4<T>::type (&acc)[NB][ST], const T* __restrict__ Gi, T* __restrict__ Go,
                                               const T* stash, int n0, int lane, int P, RF&& rf = RF()) {
    constexpr bool RESID = EPI == 2 || EPI == 4;
    constexpr bool CUSTOM = !std::is_same<typename std::decay<RF>::type, NoResidFn>::value;
    const int lr = lane & 15;
    const T rs2 = T(0.70710678118654752440);
    // Row groups q = 4a + r (4 rows x P slots each) are independent.  The residual of the first NQL groups comes from
    // the LDS stash; the others are re-read from memory, up to DEPTH groups in flight, and those loads are issued
    // before the stash rows are worked on, so their latency overlaps that arithmetic.
    constexpr int NQ = NB * 4, NQL = NA * 4, NQG = (RESID && !CUSTOM) ? NQ - NQL : 0, DMAX = 40 / (ST * (int)sizeof(T) / 4) > 1 ? 40 / (ST * (int)sizeof(T) / 4) : 1,
                  DEPTH = NQG < DMAX ? NQG : DMAX;      // about 40 VGPRs of loads in flight
    T hq[DEPTH > 0 ? DEPTH : 1][ST];
    auto fetch = [&](int q, int slot) {
        const int n = n0 + 16 * (q >> 2) + acc_row<T>(lane, q & 3);
#pragma unroll
        for (int s = 0; s < ST; ++s) hq[slot][s] = Gi[n * P + 16 * s];
    };
#pragma unroll
    for (int g = 0; g < DEPTH; ++g) fetch(NQL + g, g);
    if (NA > 0) {      // the stash is read by the wave that wrote it (f32: by another lane of it): order LDS within the wave
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // tanh of the NQ <= 16 value slots in ONE evaluation: lane lr of every 16-lane row takes row group q = lr (its
    // value slot sits in lane 0 of the row), the results go back with row broadcasts
    T yall = 0;
    if (EPI < 3 || EPI == 9) {
        T zsel = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const T v = row16_bcast<0>(acc[q >> 2][0][q & 3]);
            zsel = lr == q ? v : zsel;
        }
        yall = ds_tanh(zsel);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int a = q >> 2, r = q & 3;
        const int n = n0 + 16 * a + acc_row<T>(lane, r);
        T z[ST], hv[ST];
#pragma unroll
        for (int s = 0; s < ST; ++s) z[s] = acc[a][s][r];
        if (RESID) {
            if constexpr (CUSTOM) {
            } else if (q < NQL) {
                const int rr = 16 * a + acc_row<T>(lane, r);       // row of the wave's block: k-step rr / 4, operand lane group rr % 4
#pragma unroll
                for (int s = 0; s < ST; ++s) hv[s] = stash[((rr >> 2) * ST + s) * 64 + ((rr & 3) << 4) + lr];
            } else {
                const int slot = (q - NQL) % (DEPTH > 0 ? DEPTH : 1);
#pragma unroll
                for (int s = 0; s < ST; ++s) hv[s] = hq[slot][s];
                if (q + DEPTH < NQ) fetch(q + DEPTH, slot);
            }
        }
        if (EPI == 3 || EPI == 4) {
#pragma unroll
            for (int s = 0; s < ST; ++s) {
                T o = ds_tanh(z[s]);
                if (EPI == 4) o = (hv[s] + o) * rs2;
                Go[n * P + 16 * s] = o;
            }
            continue;
        }
        T ss = 0;
#pragma unroll
        for (int s = 0; s < ST; ++s)
            if (16 * s + lr >= 2) ss += z[s] * z[s];
        ss = row16_sum(ss);
        const T zL = row16_bcast<1>(z[0]);
        const T y = row16_bcast_dyn<NQ>(yall, q), d1 = 1 - y * y, d2 = -2 * y * d1;
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            T o = d1 * z[s];
            if (s == 0) { if (lr == 0) o = y; else if (lr == 1) o = d1 * zL + d2 * ss; }
            if (EPI == 9) { if (s == 0 && lr < 2) Go[2 * n] = o; continue; }      // (Go carries + lr: slot 0 -> y, slot 1 -> oL)
            if (CUSTOM && RESID) { acc[a][s][r] = o; continue; }
            if (EPI == 2) o = (hv[s] + o) * rs2;
            __builtin_nontemporal_store(o, &Go[n * P + 16 * s]);      // streamed out: the next reader comes after the whole launch
        }
        if constexpr (CUSTOM && RESID) { if (r == 3) rf(a); }
    }
}

// The residual layer epilogue (EPI 2) of the 64-feature x 5-tile wave tile with the shared term S ADDED HERE instead of loaded into the
// accumulators before the first MFMA (where 80 loads per lane sit in front of the products): the rows of S of the first NSF row groups
// and the value-tile entries of all groups were requested during the last k-steps into the operand ring's dying registers (sfull, s0),
// the other groups' rows follow two groups ahead like the residual rows.
template <typename T, int NB, int ST, int NA, int NSF>
__device__ __forceinline__ void layer_epilogue_sadd(typename Acc4<T>::type (&acc)[NB][ST], const T* __restrict__ Gi, T* __restrict__ Go,
                                                    const T* stash, int n0, int lane, int P, const T* __restrict__ Sp, T (&sfull)[NSF][ST],
                                                    T (&s0)[NB * 4]) {
    const int lr = lane & 15;
    const T rs2 = T(0.70710678118654752440);
    constexpr int NQ = NB * 4, NQL = NA * 4, NQG = NQ - NQL, DEPTH = NQG < 2 ? NQG : 2, DS = 2;
    T hq[DEPTH > 0 ? DEPTH : 1][ST], sq[DS][ST];
    auto fetch = [&](int q, int slot) {
        const int n = n0 + 16 * (q >> 2) + acc_row<T>(lane, q & 3);
#pragma unroll
        for (int s = 0; s < ST; ++s) hq[slot][s] = Gi[n * P + 16 * s];
    };
    auto fetch_s = [&](int q, int slot) {                      // (tile 0 of the row is already in s0[q])
        const int n = n0 + 16 * (q >> 2) + acc_row<T>(lane, q & 3);
#pragma unroll
        for (int s = 1; s < ST; ++s) sq[slot][s] = Sp[n * P + 16 * s];
    };
#pragma unroll
    for (int g = 0; g < DS; ++g)
        if (NSF + g < NQ) fetch_s(NSF + g, g);
#pragma unroll
    for (int g = 0; g < DEPTH; ++g) fetch(NQL + g, g);
    if (NA > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    T zsel = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const T v = row16_bcast<0>(acc[q >> 2][0][q & 3] + s0[q]);
        zsel = lr == q ? v : zsel;
    }
    const T yall = ds_tanh(zsel);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int a = q >> 2, r = q & 3;
        const int n = n0 + 16 * a + acc_row<T>(lane, r);
        T z[ST], hv[ST];
        z[0] = acc[a][0][r] + s0[q];
        if (q < NSF) {
#pragma unroll
            for (int s = 1; s < ST; ++s) z[s] = acc[a][s][r] + sfull[q][s];
        } else {
            const int slot = (q - NSF) % DS;
#pragma unroll
            for (int s = 1; s < ST; ++s) z[s] = acc[a][s][r] + sq[slot][s];
            if (q + DS < NQ) fetch_s(q + DS, slot);
        }
        if (q < NQL) {
            const int rr = 16 * a + acc_row<T>(lane, r);
#pragma unroll
            for (int s = 0; s < ST; ++s) hv[s] = stash[((rr >> 2) * ST + s) * 64 + ((rr & 3) << 4) + lr];
        } else {
            const int slot = (q - NQL) % (DEPTH > 0 ? DEPTH : 1);
#pragma unroll
            for (int s = 0; s < ST; ++s) hv[s] = hq[slot][s];
            if (q + DEPTH < NQ) fetch(q + DEPTH, slot);
        }
        T ss = 0;
#pragma unroll
        for (int s = 0; s < ST; ++s)
            if (16 * s + lr >= 2) ss += z[s] * z[s];
        ss = row16_sum(ss);
        const T zL = row16_bcast<1>(z[0]);
        const T y = row16_bcast_dyn<NQ>(yall, q), d1 = 1 - y * y, d2 = -2 * y * d1;
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            T o = d1 * z[s];
            if (s == 0) { if (lr == 0) o = y; else if (lr == 1) o = d1 * zL + d2 * ss; }
            o = (hv[s] + o) * rs2;
            __builtin_nontemporal_store(o, &Go[n * P + 16 * s]);
        }
    }
}

// Fused orbital-head epilogue (network.py:543-557) on a wave's accumulator tile: complex phi from the packed (Re, Im) columns, M = phi * q
// (envelope x Bloch phase 5-jet of the tile's electron) with the product rule on the jets, stored into MOUT (slot-tile major).
template <typename T, int NB, int ST>
__device__ __forceinline__ void orbital_epilogue(typename Acc4<T>::type (&acc)[NB][ST], const OrbEpi<T>& oe, int tile, int w, int n0, int lane,
                                                 const T* __restrict__ Sb, int Nout, int P) {
    const int lr = lane & 15, lq = lane >> 4;
    const int i = oe.i0 + tile, so = 2 + 3 * i, base = lane & 48;
    T* Mw = oe.MOUT + (size_t)w * oe.mout_stride + oe.mout_off;
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int ab = 0; ab < 2; ++ab) {
            const int p = 8 * (n0 / 16 + a) + lq + 4 * ab;
            const bool valid = p < oe.nparam;
            const T* q = oe.Q + ((size_t)(w * oe.N + i) * oe.nparam_max + (valid ? p : 0)) * 10;
            const Cx<T> qv(q[0], q[1]), qg0(q[2], q[3]), qg1(q[4], q[5]), qg2(q[6], q[7]), ql(q[8], q[9]);
            Cx<T> phi[ST];
#pragma unroll
            for (int s = 0; s < ST; ++s) phi[s] = Cx<T>(acc[a][s][2 * ab], acc[a][s][2 * ab + 1]);
            if (Sb) {   // use_last_layer: the orbital input has spin-mean rows too (network.py:535) -> shared term
                const T* Sr = Sb + (size_t)w * Nout * P + (size_t)(n0 + 16 * a + acc_row<T>(lane, 2 * ab)) * P + lr;
                const T* Si = Sb + (size_t)w * Nout * P + (size_t)(n0 + 16 * a + acc_row<T>(lane, 2 * ab + 1)) * P + lr;
#pragma unroll
                for (int s = 0; s < ST; ++s) { phi[s].re += Sr[16 * s]; phi[s].im += Si[16 * s]; }
            }
            if (oe.bias && valid && lr == 0) { phi[0].re += oe.bias[p]; phi[0].im += oe.bias[oe.nparam + p]; }
            const Cx<T> f0(row16_bcast<0>(phi[0].re), row16_bcast<0>(phi[0].im));
            const Cx<T> fL(row16_bcast<1>(phi[0].re), row16_bcast<1>(phi[0].im));
            Cx<T> fo[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // slot sl = so + c of this row: its slot tile st is workgroup-uniform -- a scalar branch around the ONE cross-lane read of
                // that tile (a select chain over the tiles costs 16 vector-ALU instructions per slot, and those take matrix-pipe issue cycles)
                const int sl = so + c, st = __builtin_amdgcn_readfirstlane(sl >> 4), src = base | (sl & 15);
                fo[c] = Cx<T>(T(0), T(0));
#pragma unroll
                for (int s = 0; s < ST; ++s)
                    if (s == st) fo[c] = Cx<T>(__shfl(phi[s].re, src), __shfl(phi[s].im, src));
            }
            const Cx<T> lap = fL * qv + f0 * ql + T(2) * (fo[0] * qg0 + fo[1] * qg1 + fo[2] * qg2);
            if (valid) {
                const int kdet = p / oe.norb, m = p % oe.norb;
                // MOUT is slot-tile major: [det][slot tile][elec][orb][re,im][16]
                T* mo = Mw + (size_t)kdet * oe.n * oe.n * 2 * P + (((size_t)(oe.row0 + tile) * oe.n + m) * 2) * 16 + lr;
                const size_t tstride = (size_t)oe.n * oe.n * 2 * 16;
                // the electron's own three coordinate slots live in slot tile(s) st0 (.. st1), workgroup-uniform: only those
                // tiles pay for the lane selects; the Laplacian slot is lane 1 of tile 0
                const Cx<T> t0 = f0 * qg0, t1 = f0 * qg1, t2 = f0 * qg2;
                const int st0 = so >> 4, st1 = (so + 2) >> 4;
#pragma unroll
                for (int s = 0; s < ST; ++s) {
                    T vr = phi[s].re * qv.re - phi[s].im * qv.im, vi = phi[s].re * qv.im + phi[s].im * qv.re;
                    if (s == st0 || s == st1) {
                        const int dsl = 16 * s + lr - so;
                        vr += dsl == 0 ? t0.re : (dsl == 1 ? t1.re : (dsl == 2 ? t2.re : T(0)));
                        vi += dsl == 0 ? t0.im : (dsl == 1 ? t1.im : (dsl == 2 ? t2.im : T(0)));
                    }
                    if (s == 0) { vr = lr == 1 ? lap.re : vr; vi = lr == 1 ? lap.im : vi; }
                    __builtin_nontemporal_store(vr, &mo[s * tstride]);          // MOUT is read again only by the determinant kernels
                    __builtin_nontemporal_store(vi, &mo[s * tstride + 16]);
                }
            }
        }
}

// One workgroup = one "tile" (the P jet slots of one electron, or of the spin means) x up to 1024/NB
// output features (grid.z walks further column blocks); every wave owns 16*NB features.  Tiles 0..n_tiles-1 use (X, W, K);
// the optional extra tile (blockIdx.x == n_tiles) uses (X2, W2, K2): the shared spin-mean term.
//   X  : [walker][tile][ldx rows][P]      W : [K][Nout]      Z : [walker][tile (+1)][Nout][P]   (EPI 0 / 6 / 7)
//   layer epilogues (EPI 1 - 4): Z = the next layer's G, [walker][tile][z_tile_stride / P rows][P]
//   EPI = 0: store the raw products Z.
//   EPI = 1/2: fused one-electron-layer epilogue (network.py:524-528): z = Z + S + b (the accumulators START at S + b, so
//              the epilogue has nothing to load for it), tanh chain rule on the jets, (EPI = 2) residual with the layer
//              input rows (parked in LDS during the main loop, see stash_blocks), store into the next layer's G.
//              S : [walker][Nout][P] shared spin-mean term, Gout : [walker][tile][ldo rows][P].
//   EPI = 9: layer epilogue that keeps only (y_n, oL_n) = slots 0 / 1 of its output: Z = YO [walker][tile][Nout][2] (the dense
//            output is not needed: low-rank first hidden layer, k_layer1_lr / k_layer0_means)
//   EPI = 3/4: value chain (slots = walkers): plain tanh(Z + S + b) without / with residual.
//   EPI = 5: orbital head (network.py:543-557): complex phi from packed columns, M = phi * q (envelope x Bloch
//            phase 5-jet of the tile's electron) with the product rule, stored into MOUT.
//   EPI = 8: the same for the value chain (slots = walkers): M = phi * q, values only.
template <typename T, int NB, int ST, int EPI, int G4 = 0>
// (very wide slot ranges, ST > 10: four waves per workgroup so that a wave may use the whole register file)
__global__ void __launch_bounds__((NB == 3 || ST > 5 ? 256 : 1024 / NB), (ST <= 10 ? 2 : 1))
k_jet_gemm(const T* __restrict__ X, size_t x_walker_stride, size_t x_tile_stride, const T* __restrict__ W, int K,
           const T* __restrict__ X2, size_t x2_walker_stride, const T* __restrict__ W2, int K2, int n_tiles,
           T* __restrict__ Z, size_t z_walker_stride, size_t z_tile_stride, int Nout, int P, const T* __restrict__ Sb,
           const T* __restrict__ bias, OrbEpi<T> oe) {
    typedef typename Acc4<T>::type acc_t;
    // XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs in dispatch order (x fastest), so
    // linear ids b, b+8, b+16, ... share an L2.  All tiles of one walker are given ids of one residue class:
    // they then share that L2's copy of the walker's S term (and run back to back).  Pure speed, any
    // placement gives the same result.
    int tile = blockIdx.x, w = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const unsigned b = blockIdx.y * gridDim.x + blockIdx.x, q = b >> 3;
        w = (q / gridDim.x) * 8 + (b & 7);
        tile = q % gridDim.x;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;     // (wave-uniform values in SGPRs)
    // (layer launches fold the column blocks into grid.x: the workgroups sharing an electron tile then run side by side on one XCD)
    int zb = blockIdx.z;
    if ((EPI == 1 || EPI == 2 || EPI == 5 || EPI == 9) && gridDim.x > (unsigned)n_tiles) {
        const int gzf = gridDim.x / n_tiles;
        zb = tile % gzf;
        tile /= gzf;
    }
    const int lr = lane & 15, lq = lane >> 4, n0 = (zb * (blockDim.x >> 6) + wave) * 16 * NB;
    if (n0 >= Nout) return;                      // column blocks beyond Nout (grid.z rounds up)
    // (the clock probe and the phase stamps exist in `make EXP=1` builds only: the shipped kernel carries no instrumentation)
    long long clk_c0 = 0, clk_r0 = 0;
    if (EPI == 2 && DS_EXP(oe.clk != nullptr) && wave == 0) { clk_c0 = clock64(); clk_r0 = wall_clock64(); }
    unsigned long long* tl = (EPI == 2 && DS_EXP(oe.clk != nullptr) && (oe.dbg & 32) && blockIdx.x == gridDim.x / 8 && blockIdx.y == gridDim.y / 2 && wave == 0 && lane == 0) ? oe.clk + 2 : nullptr;
    int n_tl = 0;
    auto stamp = [&]() { if (EPI == 2 && tl) { __builtin_amdgcn_s_waitcnt(0); tl[n_tl++] = (unsigned long long)clock64(); } };
    stamp();
    const T* Xp;
    const T* Wp;
    int nks;
    if (tile < n_tiles) {
        Xp = X + (size_t)w * x_walker_stride + (size_t)tile * x_tile_stride;
        Wp = W;
        nks = K / 4;
    } else {
        // (tiles beyond n_tiles of a shared-operand launch are COLUMN blocks of the one operand: x_tile_stride = columns per block)
        Xp = X2 + (size_t)w * x2_walker_stride + (size_t)(tile - n_tiles) * x_tile_stride;
        Wp = W2;
        nks = K2 / 4;
    }
    // Xp / Wp stay wave-uniform (SGPR pairs advanced by scalar adds); the lane's place in the operand tile is a 32-bit offset
    Wp += n0;
    const unsigned xo = lq * P + lr, wo = lq * Nout + lr;
    constexpr bool LAYER = (EPI >= 1 && EPI <= 4) || EPI == 9, RESID = EPI == 2 || EPI == 4;
    constexpr int NA = RESID ? stash_blocks<T, NB, ST>() : 0;          // 16-row blocks of the residual parked in LDS
    extern __shared__ __attribute__((aligned(16))) char gemm_smem[];
    T* stash = reinterpret_cast<T*>(gemm_smem) + (size_t)wave * (NA * 4 * ST * 64);
    acc_t acc[NB][ST];
    // (the dense 24-electron instance adds S in its epilogue: layer_epilogue_sadd)
    constexpr bool SADD = EPI == 2 && NB == 4 && ST == 5 && sizeof(T) == 8 && DS_SADD;
    static_assert(G4 == 0 || ((EPI == 2 || EPI == 5) && sizeof(T) == 8 && G4 <= 3), "column groups on the last slot tile: float64 residual layers and orbital heads, at most 12 jets on that tile");
    // G4 > 0: the last slot tile as G4 four-column groups.  Their B operand (B_blk[k][j] in lane 16 k + 4 blk + j, the same for every
    // block) is formed through 512 bytes of LDS per wave -- the stash entries of the last tile's PADDING columns (lanes lr >= 12 of the
    // first four parked k-steps: zeros, restored behind the loop; the stash fills the workgroup's 80 KB exactly) --: the 16-column
    // operand of the tile is written as it stands at the start of the k-step and read back as
    // lane (lq, lr) <- X[k = lq][16 (ST - 1) + 4 g + (lr & 3)] -- LDS instructions, not vector ALU ones (those
    // take matrix-pipe issue cycles on this part: forming the groups with DPP moves cost what the shorter products saved), and no
    // younger memory loads in front of the ring's (the load counter completes in order).  Accumulators start at zero (SADD) and are
    // turned into acc[.][ST - 1] before the epilogue.
    double c4[G4 ? NB : 1][G4 ? G4 : 1];
#pragma unroll
    for (int a = 0; a < (G4 ? NB : 1); ++a)
#pragma unroll
        for (int g = 0; g < (G4 ? G4 : 1); ++g) c4[a][g] = 0;
    constexpr int NSF = DS_SADD_NSF;
    T sfull[SADD ? NSF : 1][ST], s0[NB * 4];
    if constexpr (G4 > 0 && LAYER && !SADD) {
        // (the group accumulators start at the shared term too: lane (lq, lr) holds feature 4 (lr >> 2) + lq, column 4 g + (lr & 3))
        const T* Sg = Sb + ((size_t)w * Nout + n0 + 4 * (lr >> 2) + lq) * P + 16 * (ST - 1) + (lr & 3);
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int g = 0; g < G4; ++g) c4[a][g] = Sg[(size_t)16 * a * P + 4 * g];
    }
    if (LAYER && !SADD && !(EPI == 2 && DS_EXP(oe.dbg & 2))) {
        // z = W x + (S + b): the accumulators start at the shared spin-mean term, which already carries the bias
        // (EPI = 6 / 7 below, k_shared_term); these loads overlap the first operand loads
        const T* Sp0 = Sb + (size_t)w * Nout * P + lr;
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 16 * a + acc_row<T>(lane, r);
#pragma unroll
                for (int s = 0; s < ST; ++s) acc[a][s][r] = Sp0[n * P + 16 * s];
            }
    } else {
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int s = 0; s < ST; ++s) acc[a][s] = acc_t{0, 0, 0, 0};
    }
    // Operand ring of four k-steps: set u holds the operands of k-step ks + u; as soon as its MFMAs are issued the set is
    // reloaded for k-step ks + u + 4.  Loads are therefore requested three k-steps (60 MFMAs) before they are needed,
    // with no register copies at the loop end (the 4x unrolled body renames the sets).
    constexpr int NSET = ring_sets<T, NB, ST>();
    T av[NSET][NB], bv[NSET][ST];
    T bg[G4 ? G4 : 1];                      // group operands of the current k-step
    // element (k = lq, column c) at gq[GQS (c >> 2)] (c & 3 = lr & 3); the orbital head (no stash) has 512 bytes per wave of its own
    constexpr bool GQ_STASH = RESID && NA >= 1;      // (instances without a stash have the LDS to themselves: 512 bytes per wave, like the orbital head)
    constexpr int GQS = GQ_STASH ? 16 : 4;
    T* gq = GQ_STASH ? stash + (lq * ST + ST - 1) * 64 + 12 + (lr & 3) : reinterpret_cast<T*>(gemm_smem) + wave * 64 + lq * 16 + (lr & 3);
    const T* Wl = Wp + wo;                  // this lane's operands of the next k-step to request
    const T* Xl = Xp + xo;

    const size_t wstep = (size_t)4 * Nout, xstep = (size_t)4 * P;
    auto load_set = [&](int u) {
#pragma unroll
        for (int a = 0; a < NB; ++a) av[u][a] = Wl[16 * a];
#pragma unroll
        for (int s = 0; s < ST; ++s) bv[u][s] = Xl[16 * s];
        Wl += wstep;
        Xl += xstep;
    };
    auto load_groups = [&](int u) {
        if constexpr (G4 > 0) {
            gq[GQS * (lr >> 2)] = bv[u][ST - 1];
#pragma unroll
            for (int g = 0; g < G4; ++g) bg[g] = gq[GQS * g];
        }
    };
    auto last_tile = [&](int u) {
        if constexpr (G4 > 0) {
#pragma unroll
            for (int g = 0; g < G4; ++g)
#pragma unroll
                for (int a = 0; a < NB; ++a) c4[a][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[u][a], bg[g], c4[a][g], 0, 0, 0);
        }
    };
    auto step = [&](int u, int k) {
        load_groups(u);
        if (NA > 0) {                    // k-steps n0/4 .. n0/4 + 4*NA - 1 carry this wave's residual rows: park the operands
            const int j = k - (n0 >> 2);
            if (j >= 0 && j < 4 * NA) {
#pragma unroll
                for (int s = 0; s < ST; ++s) stash[(j * ST + s) * 64 + lane] = bv[u][s];
            }
        }
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int s = 0; s < (G4 ? ST - 1 : ST); ++s) acc[a][s] = mfma16(av[u][a], bv[u][s], acc[a][s]);
        last_tile(u);
    };
    // (pair-mean rows: a k-step under a wave-uniform slot-tile mask; bit-identical, the skipped products add exact zeros)
    auto step_m = [&](int u, unsigned m) {
        load_groups(u);
#pragma unroll
        for (int s = 0; s < ST; ++s)
            if ((m >> s) & 1) {
                if (G4 > 0 && s == ST - 1) last_tile(u);
                else {
#pragma unroll
                    for (int a = 0; a < NB; ++a) acc[a][s] = mfma16(av[u][a], bv[u][s], acc[a][s]);
                }
            }
    };
    auto park = [&](int u, int k) {      // (the residual parking of `step`, for the masked rounds that straddle pm_k0)
        if (NA > 0) {
            const int j = k - (n0 >> 2);
            if (j >= 0 && j < 4 * NA) {
#pragma unroll
                for (int s = 0; s < ST; ++s) stash[(j * ST + s) * 64 + lane] = bv[u][s];
            }
        }
    };
    // every float64 residual-layer instance masks every pair-mean k-step
    constexpr bool PMASK = EPI == 2 && pm_instance<T>(ST);
    // ... and layer 0 (EPI 1 / 9, the plain loop below): its own-feature rows (k-steps below pm_k0) fill slot tile 0 and the own tile(s)
    constexpr bool PM0 = (EPI == 1 || EPI == 9) && pm_instance<T>(ST);
    unsigned tmask[2] = {~0u, ~0u};
    int pm_k0 = nks;                      // first masked k-step (nks: none)
    unsigned own0 = ~0u;                  // (layer 0) mask of the own-feature rows
    if constexpr (PMASK || PM0) {
        if (oe.pm_ks > 0 && tile < n_tiles) {
            pm_k0 = oe.pm_k0;
            const unsigned own = (1u << ((2 + 3 * tile) >> 4)) | (1u << ((4 + 3 * tile) >> 4));
            own0 = 1u | own;
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const int j0 = sp == 0 ? 0 : oe.pm_nup, ns = (sp == 0 || oe.pm_nch == 1) ? (oe.pm_nch == 1 ? n_tiles : oe.pm_nup) : n_tiles - oe.pm_nup;
                const int lo = (2 + 3 * j0) >> 4, hi = (4 + 3 * (j0 + ns - 1)) >> 4;
                tmask[sp] = 1u | own | (((2u << hi) - 1u) & ~((1u << lo) - 1u));
            }
        }
    }
    auto pm_mask = [&](int k) -> unsigned { return tmask[(k - pm_k0) >= oe.pm_ks ? 1 : 0]; };
    auto step_g = [&](int u, int k) {      // a k-step on either side of pm_k0: ONE code path (two inlined variants per call spill)
        if constexpr (PMASK) { park(u, k); step_m(u, k >= pm_k0 ? pm_mask(k) : ~0u); }
        else step(u, k);
    };
    // (compile-time choice: the launcher guarantees K % 16 == 0 for the ring instantiations -- gemm_uses_ring)
    if (gemm_uses_ring(EPI) || ((EPI == 6 || EPI == 7) && (nks & 3) == 0)) {      // (EPI 6 / 7: the long shared-term products too)
        // every load of the steady state is unconditional, so the outstanding-load count is the same on every path and the
        // waits stay partial (vmcnt(27)): a conditional reload would force a full drain at the loop head
#pragma unroll
        for (int u = 0; u < NSET; ++u) load_set(u);       // (nks >= 4 >= NSET)
        stamp();
        int ks = 0;
        if (NSET == 4) {
            for (; ks + 4 < ((PMASK && pm_k0 < nks) ? pm_k0 + 1 : nks); ks += 4) {      // (masked instance: up to k-step pm_k0 - 1)
#pragma unroll
                for (int u = 0; u < 4; ++u) { step(u, ks + u); load_set(u); }
            }
            if constexpr (PMASK) {
                // the pair-mean rows (pm_k0 is a multiple of 4 k-steps and beyond the residual rows: nothing to park), but the last round
                // (below, with the shared term's loads in its shadow)
                for (; ks + 4 < nks; ks += 4) {
                    const unsigned m = tmask[(ks - pm_k0) >= oe.pm_ks ? 1 : 0];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { step_m(u, m); load_set(u); }
                }
            }
            if constexpr (SADD) {
                // the last four k-steps reload nothing: rows of S go into the registers of the sets as they die -- the full rows of
                // the first NSF row groups and the value-tile entries of all sixteen (NSF = 2: 10 + 14 values; 5 would fill the
                // 36 registers of the ring but spills 68 bytes and measured slower)
                const T* Sp0 = Sb + (size_t)w * Nout * P + lr;
                auto srow = [&](int q) { return Sp0 + (size_t)(n0 + 16 * (q >> 2) + acc_row<T>(lane, q & 3)) * P; };
                auto lfull = [&](int q) {
                    const T* p = srow(q);
#pragma unroll
                    for (int s = 0; s < ST; ++s) sfull[q][s] = p[16 * s];
                    s0[q] = sfull[q][0];
                };
                const unsigned mlast = (PMASK && pm_k0 < nks) ? pm_mask(ks) : ~0u;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if constexpr (PMASK) step_m(u, mlast); else step(u, ks + u);      // (the last round carries no residual rows: nothing to park)
#pragma unroll
                    for (int q = 0; q < NB * 4; ++q)
                        if ((q & 3) == u) { if (q < NSF) lfull(q); else s0[q] = srow(q)[0]; }
                }
            } else if constexpr (PMASK) {
                const unsigned m = pm_k0 < nks ? pm_mask(ks) : ~0u;      // (pm_ks is a multiple of four k-steps: one partner spin per round)
#pragma unroll
                for (int u = 0; u < 4; ++u) { park(u, ks + u); step_m(u, m); }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) step(u, ks + u);
            }
        } else {
            // fewer sets (wide slot ranges: the accumulators leave room for two or three): nks need not divide, the last
            // one or two rounds reload conditionally
            for (; ks + 2 * NSET <= nks && (!PMASK || ks + NSET <= pm_k0); ks += NSET) {      // (rounds in front of the pair-mean rows)
#pragma unroll
                for (int u = 0; u < NSET; ++u) { step(u, ks + u); load_set(u); }
            }
            if constexpr (PMASK) {
                for (; ks + 2 * NSET <= nks; ks += NSET) {
#pragma unroll
                    for (int u = 0; u < NSET; ++u) { step_g(u, ks + u); load_set(u); }
                }
            }
#pragma unroll
            for (int u = 0; u < NSET; ++u) {
                step_g(u, ks + u);
                if (ks + u + NSET < nks) load_set(u);
            }
            ks += NSET;
#pragma unroll
            for (int u = 0; u < NSET; ++u)
                if (ks + u < nks) step_g(u, ks + u);
        }
    } else {
        // short contractions (layer 0: K = 12, 8): one set, no ring
        if constexpr (PM0) {
            for (int ks = 0; ks < nks; ++ks) { load_set(0); step_m(0, pm_k0 < nks ? (ks < pm_k0 ? own0 : pm_mask(ks)) : ~0u); }
        } else {
            for (int ks = 0; ks < nks; ++ks) { load_set(0); step(0, ks); }
        }
    }
    if constexpr (G4 > 0) {
        if constexpr (GQ_STASH) gq[GQS * (lr >> 2)] = T(0);              // (the padding columns of the parked rows are zeros again)
        // the column groups back into the accumulator layout of a 16-column tile (DPP inside the 16-lane rows): the epilogue is unchanged
#pragma unroll
        for (int a = 0; a < NB; ++a) {
            acc[a][ST - 1][0] = quads_to_tile<G4, 0>(c4[a]);
            acc[a][ST - 1][1] = quads_to_tile<G4, 1>(c4[a]);
            acc[a][ST - 1][2] = quads_to_tile<G4, 2>(c4[a]);
            acc[a][ST - 1][3] = quads_to_tile<G4, 3>(c4[a]);
        }
    }
    stamp();
    if (EPI == 0 || EPI == 6 || EPI == 7) {
        // EPI 6 / 7: the shared term of a layer, stored WITH the layer's bias (6: on the value slot of the jets,
        // 7: on every walker column of the value chain), so the consuming GEMM starts its accumulators at S + b
        T* Zp = Z + (size_t)w * z_walker_stride + (EPI == 7 ? (size_t)tile * z_tile_stride : (size_t)tile * Nout * P);
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 16 * a + acc_row<T>(lane, r);
                const T bn = EPI == 0 ? T(0) : bias[n];
#pragma unroll
                for (int s = 0; s < ST; ++s)
                    Zp[(size_t)n * P + 16 * s + lr] = acc[a][s][r] + ((EPI == 7 || (EPI == 6 && s == 0 && lr == 0)) ? bn : T(0));
            }
    } else if (EPI == 5) {
        orbital_epilogue<T, NB, ST>(acc, oe, tile, w, n0, lane, Sb, Nout, P);
    } else if (EPI == 8) {
        // value chain (slot tiles = 80 walkers of group w): M = (phi + S + bias) * q straight from the accumulators into MOUT
        // [group][spin][det][elec][orb][re,im][PV] -- the arithmetic of k_orbital_epilogue_val without the trip through PHI
        const int i = oe.i0 + tile;
        T* Mw = oe.MOUT + (size_t)w * oe.mout_stride + oe.mout_off;
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int ab = 0; ab < 2; ++ab) {
                const int p = 8 * (n0 / 16 + a) + lq + 4 * ab;
                if (p >= oe.nparam) continue;
                const T* q = oe.Q + ((size_t)(w * oe.N + i) * oe.nparam_max + p) * 2 * P + lr;
                T* mo = Mw + (((size_t)((p / oe.norb) * oe.n + oe.row0 + tile) * oe.n + p % oe.norb) * 2) * P + lr;
                const int cr = n0 + 16 * a + acc_row<T>(lane, 2 * ab), ci = n0 + 16 * a + acc_row<T>(lane, 2 * ab + 1);
#pragma unroll
                for (int s = 0; s < ST; ++s) {
                    Cx<T> phi(acc[a][s][2 * ab], acc[a][s][2 * ab + 1]);
                    if (Sb) { phi.re += Sb[((size_t)w * Nout + cr) * P + 16 * s + lr]; phi.im += Sb[((size_t)w * Nout + ci) * P + 16 * s + lr]; }
                    if (oe.bias) { phi.re += oe.bias[p]; phi.im += oe.bias[oe.nparam + p]; }
                    const Cx<T> v = phi * Cx<T>(q[16 * s], q[P + 16 * s]);
                    mo[16 * s] = v.re;
                    mo[P + 16 * s] = v.im;
                }
            }
    } else if (EPI == 2 && DS_EXP(oe.dbg & 1)) {
        // (timing experiment: no epilogue)
        T v = 0;
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int s = 0; s < ST; ++s) v += acc[a][s][0] + acc[a][s][1] + acc[a][s][2] + acc[a][s][3];
        if (v == T(12345.678)) Z[0] = v;
    } else {
        // Z here is the next layer's G: [walker][tile][z_tile_stride / P rows][P]; the residual rows are rows n of the input tile
        const T* Gi = X + (size_t)w * x_walker_stride + (size_t)tile * x_tile_stride + lr;
        T* Go = Z + (size_t)w * z_walker_stride + (size_t)tile * z_tile_stride + lr;
        if constexpr (SADD) layer_epilogue_sadd<T, NB, ST, NA, NSF>(acc, Gi, Go, stash, n0, lane, P, Sb + (size_t)w * Nout * P + lr, sfull, s0);
        else layer_epilogue<T, NB, ST, EPI, NA>(acc, Gi, Go, stash, n0, lane, P);
    }
    stamp();
    if (EPI == 2 && DS_EXP(oe.clk != nullptr) && wave == 0 && lane == 0) {
        atomicAdd(oe.clk, (unsigned long long)(clock64() - clk_c0));
        atomicAdd(oe.clk + 1, (unsigned long long)(wall_clock64() - clk_r0));
    }
}

// =====================================================================================
// First hidden layer on the LOW-RANK form of its input.
//
// The output of layer 0 has, in its derivative slots, rank K0 = (rows of the layer-0 input, per-electron + shared):
//     G1[n][s] = y'_n * sum_{k < K0} W0[k][n] X0[k][s]            (s >= 2),     G1[n][0] = y_n,   G1[n][1] = oL_n
// (network.py:524-528 with the tanh chain rule on the jets; no residual at layer 0 when its width changes), so the product of
// layer 1 over its Kh one-electron rows collapses to K0 + 2 rows with PER-ELECTRON weights
//     C[m][k] = sum_n W1[n][m] y'_n W0[k][n]   (k < K0),    C[m][K0] = sum_n W1[n][m] y_n,    C[m][K0 + 1] = sum_n W1[n][m] oL_n
//     Z[m][s] = sum_{k < K0} C[m][k] X0[k][s]  (s >= 2),    Z[m][0] = C[m][K0],               Z[m][1] = C[m][K0 + 1]
// plus the pair-mean rows (dense, as before) and the shared term S.  Per wave tile of 16 NB features x 16 ST slots:
//     phase 1   C = W1_h^T B1, B1[n][c] = (y'_n W0T[n][c] | y_n | oL_n): Kh / 4 k-steps of NB x NC MFMAs (NC = column tiles of c)
//     phase 2   Z = C X0' (A operand from the wave's LDS copy of C; X0' = X0 with its value / Laplacian slots zeroed, plus the two
//               unit rows that route C[:, K0], C[:, K0 + 1] to slots 0 / 1) + W1_m^T M2: (K0 + 4 + Km2) / 4 k-steps of NB x ST
// instead of (Kh + Km2) / 4 k-steps of NB x ST: at 24 electrons 952 MFMAs per wave tile instead of 1600.  The epilogue is the one of
// k_jet_gemm (EPI 1 / 2); the residual rows (= layer-0 output rows) are recomputed from the layer-0 input, 16 rows at a time.
// =====================================================================================
template <typename T> struct LrArgs {
    const T* XL; size_t xl_ws, xl_ts;    // layer-0 per-electron input rows [walker][tile][K0loc][P]
    const T* M0; size_t m0_ws;           // layer-0 shared input rows (spin means of the input features) [walker][K0sh][P]
    int K0loc, K0sh;                     // multiples of 4
    const T* W0T;                        // [Kh][16 NC]: W0T[n][c] = layer-0 weight of input row c (per-electron rows, then shared rows), 0 beyond
    const T* G1; size_t g_ws, g_ts;      // layer-1 input tiles [walker][tile][rows][P]: only rows Kh.. (the pair means) are written and read
    const T* W1; int Kh, Km2;            // layer-1 weights [Kh + Km2][Nout]
    T* Gout; size_t go_ws, go_ts;        // layer-1 output tiles
    const T* S1;                         // [walker][Nout][P] shared term of layer 1 (with its bias)
    int Nout, P, n_tiles;
    // the layer-0 output is never written: y, oL come from YO, the residual rows are recomputed from the layer-0 input
    const T* YO; size_t yo_ws;           // [walker][tile][Kh][2] = (y_n, oL_n)  (k_layer0_stats)
    const T* W0; const T* S0;            // layer-0 per-electron weights [K0loc][Kh], its shared term [walker][Kh][P]
    int dbg;                             // timing experiments (make EXP=1 only): 1 no epilogue, 2 no phase 1, 4 no phase 2, 8 no S1 loads
    int n_up, nch;                       // spin-up electrons, spin channels: which slot tiles of a pair-mean row can be non-zero (k_layer1_lr)
};
// NC full 16-column tiles of C + NG groups of 4 columns behind them (float64 only: the groups are v_mfma_f64_4x4x4_4b products, 16
// cycles for 16 features x 4 columns where a 16x16x4 tile of mostly padding costs 64: K0 + 4 = 24 columns are one tile + two groups)
template <int NB, int NC, int NG = 0> constexpr int lr_ncp() { return 16 * NC + 4 * NG + 1; }
template <typename T, int NB, int NC, int NG = 0> inline size_t lr_lds_bytes(unsigned threads, int Kh) {
    return ((size_t)2 * Kh + (size_t)(threads / 64) * 16 * NB * lr_ncp<NB, NC, NG>()) * sizeof(T);
}

template <typename T, int NB, int ST, int NC, bool RES, int NG = 0>
__global__ void __launch_bounds__((ST > 5 ? 256 : 1024 / NB), (ST <= 10 ? 2 : 1)) k_layer1_lr(LrArgs<T> A) {
    typedef typename Acc4<T>::type acc_t;
    static_assert(NG == 0 || sizeof(T) == 8, "column groups: float64 only");
    constexpr int NCP = lr_ncp<NB, NC, NG>(), LDW = 16 * (NC + (NG > 0 ? 1 : 0));      // LDW: row length of W0T (whole column tiles)
    // (placement as in k_jet_gemm: the tiles of one walker on one XCD, column blocks of a tile side by side)
    int tile = blockIdx.x, w = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const unsigned b = blockIdx.y * gridDim.x + blockIdx.x, q = b >> 3;
        w = (q / gridDim.x) * 8 + (b & 7);
        tile = q % gridDim.x;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int zb = 0;
    if (gridDim.x > (unsigned)A.n_tiles) {
        const int gzf = gridDim.x / A.n_tiles;
        zb = tile % gzf;
        tile /= gzf;
    }
    const int lr = lane & 15, lq = lane >> 4, n0 = (zb * (blockDim.x >> 6) + wave) * 16 * NB;
    const int P = A.P, Nout = A.Nout, Kh = A.Kh, K0 = A.K0loc + A.K0sh;
    extern __shared__ __attribute__((aligned(16))) char lr_smem[];
    T* yl = reinterpret_cast<T*>(lr_smem);                              // [Kh][2] = (y_n, oL_n) of this electron
    T* Cl = yl + 2 * Kh + (size_t)wave * (16 * NB * NCP);               // the wave's C block [16 NB][NCP]
    const T* G1t = A.G1 + (size_t)w * A.g_ws + (size_t)tile * A.g_ts;
    {
        typedef T vec2 __attribute__((ext_vector_type(2)));
        const T* yo = A.YO + (size_t)w * A.yo_ws + (size_t)tile * 2 * Kh;
        for (int n = threadIdx.x; n < Kh; n += blockDim.x)
            *reinterpret_cast<vec2*>(yl + 2 * n) = *reinterpret_cast<const vec2*>(yo + 2 * n);
    }
    __syncthreads();
    if (n0 >= Nout) return;
    // ---------------- phase 1: C[m][c] = sum_n W1[n][m] B1[n][c]
    acc_t c1[NB][NC];
    T c4[NB][NG > 0 ? NG : 1];               // column groups: D_blk[i][j] = C[feature 4 blk + i][column 16 NC + 4 g + j] in lane 16 i + 4 blk + j
#pragma unroll
    for (int a = 0; a < NB; ++a) {
#pragma unroll
        for (int s = 0; s < NC; ++s) c1[a][s] = acc_t{0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < NG; ++g) c4[a][g] = 0;
    }
    {
        T av[4][NB], wv[4][NC], w4[4][NG > 0 ? NG : 1];
        const T* Wl = A.W1 + n0 + (size_t)lq * Nout + lr;
        const T* Tl = A.W0T + lq * LDW + lr;
        const T* T4 = A.W0T + lq * LDW + 16 * NC + (lr & 3);      // the 4-block B operand: B_blk[k][j] in lane 16 k + 4 blk + j, the same for every block
        auto load_set = [&](int u) {
#pragma unroll
            for (int a = 0; a < NB; ++a) av[u][a] = Wl[16 * a];
#pragma unroll
            for (int s = 0; s < NC; ++s) wv[u][s] = Tl[16 * s];
#pragma unroll
            for (int g = 0; g < NG; ++g) w4[u][g] = T4[4 * g];
            Wl += (size_t)4 * Nout;
            Tl += 4 * LDW;
            T4 += 4 * LDW;
        };
        auto step = [&](int u, int ks) {
            typedef T vec2 __attribute__((ext_vector_type(2)));
            const vec2 yo = *reinterpret_cast<const vec2*>(yl + 2 * (4 * ks + lq));
            const T y = yo[0], d1 = 1 - y * y;
            T bv[NC];
#pragma unroll
            for (int s = 0; s < NC; ++s) {
                const int c = 16 * s + lr;
                bv[s] = c == K0 ? y : (c == K0 + 1 ? yo[1] : d1 * wv[u][s]);
            }
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int s = 0; s < NC; ++s) c1[a][s] = mfma16(av[u][a], bv[s], c1[a][s]);
            if constexpr (NG > 0) {
                // (the A operand of the 4-block form -- A_blk[i][k] in lane 16 k + 4 blk + i -- is the 16x16x4 A operand as it stands)
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int c = 16 * NC + 4 * g + (lr & 3);
                    const T b4 = c == K0 ? y : (c == K0 + 1 ? yo[1] : d1 * w4[u][g]);
#pragma unroll
                    for (int a = 0; a < NB; ++a) c4[a][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[u][a], b4, c4[a][g], 0, 0, 0);
                }
            }
        };
        const int nks = DS_EXP(A.dbg & 2) ? 4 : Kh / 4;                       // (Kh is a multiple of 16: launcher)
#pragma unroll
        for (int u = 0; u < 4; ++u) load_set(u);
        int ks = 0;
        for (; ks + 4 < nks; ks += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { step(u, ks + u); load_set(u); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) step(u, ks + u);
    }
    // the accumulator layout (row = feature, lane = column c) is not the A-operand layout (lane = feature, k = c): through LDS
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int s = 0; s < NC; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cl[(16 * a + acc_row<T>(lane, r)) * NCP + 16 * s + lr] = c1[a][s][r];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int g = 0; g < NG; ++g) Cl[(16 * a + 4 * ((lane >> 2) & 3) + lq) * NCP + 16 * NC + 4 * g + (lane & 3)] = c4[a][g];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---------------- phase 2
    acc_t acc[NB][ST];
    if (DS_EXP(A.dbg & 8)) {
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int s = 0; s < ST; ++s) acc[a][s] = acc_t{0, 0, 0, 0};
    } else {
        const T* Sp0 = A.S1 + (size_t)w * Nout * P + lr;
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 16 * a + acc_row<T>(lane, r);
#pragma unroll
                for (int s = 0; s < ST; ++s) acc[a][s][r] = Sp0[n * P + 16 * s];
            }
    }
    const T* Ca = Cl + lr * NCP + lq;
    {
        // One operand ring over the pair-mean rows (dense: A = W1 rows Kh.., B = rows Kh.. of the input tile) and then the K0
        // low-rank rows (A = the wave's C block in LDS, B = the layer-0 input rows without their slots 0 / 1): the B rows of the
        // low-rank k-steps are requested three k-steps ahead like all others.  Branch-free: the (unused) A load of a low-rank k-step
        // re-reads the last pair-mean row, the pointers are selected by the k-step index.
        constexpr int NSET = ring_sets<T, NB, ST>();
        T av[NSET][NB], bv[NSET][ST];
        const int nm2 = A.Km2 / 4, nl = A.K0loc / 4, nk = DS_EXP(A.dbg & 4) ? 4 : nm2 + K0 / 4;
        const T* Wm = A.W1 + (size_t)(Kh + lq) * Nout + n0 + lr;
        const T* Xm = G1t + (size_t)(Kh + lq) * P + lr;
        const T* Xl = A.XL + (size_t)w * A.xl_ws + (size_t)tile * A.xl_ts + (size_t)lq * P + lr;
        const T* Ml = A.M0 + (size_t)w * A.m0_ws + (size_t)lq * P + lr;
        int kl = 0;                                        // next k-step to request
        auto load_set = [&](int u) {
            const int km = kl < nm2 ? kl : nm2 - 1;
            const T* wp = Wm + (size_t)(4 * km) * Nout;
            const T* xp = kl < nm2 ? Xm + (size_t)(4 * kl) * P : (kl - nm2 < nl ? Xl + (size_t)(4 * (kl - nm2)) * P : Ml + (size_t)(4 * (kl - nm2 - nl)) * P);
#pragma unroll
            for (int a = 0; a < NB; ++a) av[u][a] = wp[16 * a];
#pragma unroll
            for (int s = 0; s < ST; ++s) bv[u][s] = xp[16 * s];
            ++kl;
        };
        // The pair-mean rows of partner spin sp are zero outside slot tile 0 (slots 0, 1), the tiles of the electron's own three
        // slots and the tiles of the partners' slots (k_m2_expand writes exact zeros elsewhere): with 12 + 12 electrons a spin-up row
        // fills tiles 0 .. 2 (+ the own tile of a spin-down electron), a spin-down row tiles 0, 2 .. 4: the products on the other
        // tiles add exact zeros and are skipped (wave-uniform mask per k-step; bit-identical).
        // The low-rank rows (round 6) have the same structure and NO value / Laplacian entries (zeroed below): the electron's own
        // features (the first h1[0] rows of the layer-0 input) live in its own tile(s) only, the layer-0 pair-mean rows and the spin-mean
        // rows of partner spin sp in the own tile(s) and the partners' tiles -- lmask, without slot tile 0.
        unsigned tmask[2], lmask[2];
        const unsigned own = (1u << ((2 + 3 * tile) >> 4)) | (1u << ((4 + 3 * tile) >> 4));
        {
            const int n_dn = A.n_tiles - A.n_up;
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const int j0 = sp == 0 ? 0 : A.n_up, ns = (sp == 0 || A.nch == 1) ? (A.nch == 1 ? A.n_tiles : A.n_up) : n_dn;
                const int lo = (2 + 3 * j0) >> 4, hi = (4 + 3 * (j0 + ns - 1)) >> 4;
                lmask[sp] = own | (((2u << hi) - 1u) & ~((1u << lo) - 1u));
                tmask[sp] = 1u | lmask[sp];
            }
        }
        const int nm2s = A.nch > 1 ? nm2 / 2 : nm2;            // k-steps of the first partner spin's rows
        const int h10 = A.K0sh / A.nch / 4, h20 = (nl - h10) / A.nch;      // k-steps of the own-feature rows / of one spin's layer-0 pair-mean rows
        auto low_mask = [&](int kk) -> unsigned {               // kk: k-step among the low-rank rows (K0loc rows of XL, then K0sh rows of M0)
            if (kk < h10) return own;
            if (kk < nl) return lmask[(kk - h10) >= h20 ? 1 : 0];
            return lmask[(kk - nl) >= h10 ? 1 : 0];
        };
        auto step = [&](int u, int k) {
            const bool low = k >= nm2;
            const int c = low ? 4 * (k - nm2) : 0;
            // (float32: the low-rank rows stay unmasked -- diamond f32 measured 32.9 -> 35.4 ms with them masked)
            const unsigned m = low ? (sizeof(T) == 8 ? low_mask(k - nm2) : ~0u) : tmask[k < nm2s ? 0 : 1];
            T b0 = bv[u][0];
            b0 = (low && lr < 2) ? T(0) : b0;
            T aa[NB];
#pragma unroll
            for (int a = 0; a < NB; ++a) {
                const T cl = Ca[16 * a * NCP + c];
                aa[a] = low ? cl : av[u][a];
            }
#pragma unroll
            for (int s = 0; s < ST; ++s)
                if ((m >> s) & 1) {
#pragma unroll
                    for (int a = 0; a < NB; ++a) acc[a][s] = mfma16(aa[a], s == 0 ? b0 : bv[u][s], acc[a][s]);
                }
        };
        // (nk >= NSET: at least one pair-mean k-step and K0 >= 8)
#pragma unroll
        for (int u = 0; u < NSET; ++u) load_set(u);
        int ks = 0;
        for (; ks + 2 * NSET <= nk; ks += NSET) {
#pragma unroll
            for (int u = 0; u < NSET; ++u) { step(u, ks + u); load_set(u); }
        }
#pragma unroll
        for (int u = 0; u < NSET; ++u) {
            if (ks + u < nk) step(u, ks + u);
            if (ks + u + NSET < nk) load_set(u);
        }
        ks += NSET;
#pragma unroll
        for (int u = 0; u < NSET; ++u)
            if (ks + u < nk) step(u, ks + u);
        // columns K0, K0 + 1 of C go to slots 0, 1: one k-step on slot tile 0 with unit rows (lane groups 2, 3 contribute zeros)
        const T one = (lq == 0 && lr == 0) || (lq == 1 && lr == 1) ? T(1) : T(0);
#pragma unroll
        for (int a = 0; a < NB; ++a) acc[a][0] = mfma16(Ca[16 * a * NCP + K0], one, acc[a][0]);
    }
    T* Got = A.Gout + (size_t)w * A.go_ws + (size_t)tile * A.go_ts + lr;
    if (DS_EXP(A.dbg & 1)) {
        T v = 0;
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int s = 0; s < ST; ++s) v += acc[a][s][0] + acc[a][s][1] + acc[a][s][2] + acc[a][s][3];
        if (v == T(12345.678)) Got[0] = v;
        return;
    }
    if constexpr (RES) {
        // residual rows = layer-0 output rows n0 .. of this electron, recomputed per 16-row block: z0 = S0 + W0^T X0 (K0loc rows),
        // G1[n][s] = y'_n z0[n][s] for s >= 2, (y_n, oL_n) in slots 0 / 1
        const T* Xl = A.XL + (size_t)w * A.xl_ws + (size_t)tile * A.xl_ts + (size_t)lq * P + lr;
        const T* S0p = A.S0 + (size_t)w * Kh * P + lr;
        unsigned rm_sp[2];
        const unsigned rm_own = (1u << ((2 + 3 * tile) >> 4)) | (1u << ((4 + 3 * tile) >> 4));
        {
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const int j0 = sp == 0 ? 0 : A.n_up, ns = (sp == 0 || A.nch == 1) ? (A.nch == 1 ? A.n_tiles : A.n_up) : A.n_tiles - A.n_up;
                const int lo = (2 + 3 * j0) >> 4, hi = (4 + 3 * (j0 + ns - 1)) >> 4;
                rm_sp[sp] = rm_own | (((2u << hi) - 1u) & ~((1u << lo) - 1u));
            }
        }
        const int rh10 = A.K0sh / A.nch / 4, rh20 = (A.K0loc / 4 - rh10) / A.nch;
        auto rmask = [&](int ks) -> unsigned { return sizeof(T) == 4 ? ~0u : (ks < rh10 ? rm_own : rm_sp[(ks - rh10) >= rh20 ? 1 : 0]); };
        auto rf = [&](int a) {
            const T rs2 = T(0.70710678118654752440);
            // (at most five slot tiles at a time: the recomputed rows then take 40 registers next to the accumulators)
#pragma unroll
            for (int c0 = 0; c0 < ST; c0 += 5) {
                constexpr int CWMAX = ST < 5 ? ST : 5;
                const int cw = ST - c0 < 5 ? ST - c0 : 5;
                acc_t racc[CWMAX];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int n = n0 + 16 * a + acc_row<T>(lane, rr);
#pragma unroll
                    for (int s = 0; s < CWMAX; ++s)
                        if (s < cw) racc[s][rr] = S0p[(size_t)n * P + 16 * (c0 + s)];
                }
                for (int ks = 0; ks < A.K0loc / 4; ++ks) {
                    const T av = A.W0[(size_t)(4 * ks + lq) * Kh + n0 + 16 * a + lr];
                    // (the same structural zeros as in phase 2; slots 0 / 1 of the recomputed rows are replaced by (y, oL) below)
                    const unsigned mr = rmask(ks) >> c0;
#pragma unroll
                    for (int s = 0; s < CWMAX; ++s)
                        if (s < cw && ((mr >> s) & 1)) racc[s] = mfma16(av, Xl[(size_t)(4 * ks) * P + 16 * (c0 + s)], racc[s]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + 16 * a + acc_row<T>(lane, r);
                    typedef T vec2 __attribute__((ext_vector_type(2)));
                    const vec2 yo = *reinterpret_cast<const vec2*>(yl + 2 * n);
                    const T d1 = 1 - yo[0] * yo[0];
#pragma unroll
                    for (int s = 0; s < CWMAX; ++s)
                        if (s < cw) {
                            T hv = d1 * racc[s][r];
                            if (c0 + s == 0) hv = lr == 0 ? yo[0] : (lr == 1 ? yo[1] : hv);
                            __builtin_nontemporal_store((hv + acc[a][c0 + s][r]) * rs2, &Got[(size_t)n * P + 16 * (c0 + s)]);
                        }
                }
            }
        };
        layer_epilogue<T, NB, ST, 2, 0>(acc, (const T*)nullptr, Got, (const T*)nullptr, n0, lane, P, rf);
    } else
        layer_epilogue<T, NB, ST, 1, 0>(acc, (const T*)nullptr, Got, (const T*)nullptr, n0, lane, P);
}

// Layer 0 in front of the low-rank layer 1: its dense output is never written.  k_jet_gemm<.., EPI 9> leaves, per electron, the two
// numbers the low-rank form needs, YO[n] = (y_n = tanh z_n0, oL_n = y' z_nL + y'' sum_d z_nd^2); this kernel forms the mean of the
// dense output over the electrons of each spin (the input of layer 1's shared term), in electron order, from YO and the layer-0 input:
//     o_i[n][s] = y'_in (S0[n][s] + sum_k W0[k][n] X_i[k][s])   (s >= 2),   (y_in, oL_in) in slots 0 / 1
//   grid (nch * Nout / 64 * slot chunks, walkers): four waves of 16 features x STC slot tiles (no coupling between slot chunks here).
template <typename T, int STC>
__global__ void __launch_bounds__(256, (sizeof(T) == 8 ? 3 : 4))
k_layer0_means(SysDev<T> S, const T* __restrict__ XL, size_t xl_ws, size_t xl_ts, const T* __restrict__ W0, int K0loc,
               const T* __restrict__ S0, int Nout, int P, const T* __restrict__ YO, T* __restrict__ MEAN1) {
    typedef typename Acc4<T>::type acc_t;
    typedef T vec2 __attribute__((ext_vector_type(2)));
    const int ntile = P / 16, nck = (ntile + STC - 1) / STC, nfb = gridDim.x / (S.nch * nck);
    const int chunk = blockIdx.x % nck, fb = (blockIdx.x / nck) % nfb, sp = blockIdx.x / (nck * nfb), w = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int n0 = (fb * 4 + wave) * 16, nks = K0loc / 4, t0 = chunk * STC;
    if (n0 >= Nout) return;
    const int i0 = sp == 0 ? 0 : S.n_up, ns = sp == 0 ? S.n_up : S.n_dn;
    const T* W0p = W0 + (size_t)lq * Nout + n0 + lr;
    // the chunk's tile of the shared term stays in registers (padding tiles beyond the last one read tile ntile - 1 and are not stored)
    acc_t s0[STC], macc[STC];
#pragma unroll
    for (int s = 0; s < STC; ++s) {
        const int t = t0 + s < ntile ? t0 + s : ntile - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) s0[s][r] = S0[(size_t)w * Nout * P + (size_t)(n0 + acc_row<T>(lane, r)) * P + 16 * t + lr];
        macc[s] = acc_t{0, 0, 0, 0};
    }
    const int h10 = S.h1[0] / 4, h20 = S.h2[0] / 4;
    unsigned prange[2];
#pragma unroll
    for (int sp2 = 0; sp2 < 2; ++sp2) {
        const int j0 = sp2 == 0 ? 0 : S.n_up, nsp = sp2 == 0 ? S.n_up : (S.n_dn > 0 ? S.n_dn : S.n_up);
        const int lo = (2 + 3 * j0) >> 4, hi = (4 + 3 * (j0 + nsp - 1)) >> 4;
        prange[sp2] = ((2u << hi) - 1u) & ~((1u << lo) - 1u);
    }
    for (int e = 0; e < ns; ++e) {
        const int i = i0 + e;
        const T* Xl = XL + (size_t)w * xl_ws + (size_t)i * xl_ts + (size_t)lq * P + lr;
        const T* yo = YO + ((size_t)w * S.N + i) * 2 * Nout;
        vec2 yv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) yv[r] = *reinterpret_cast<const vec2*>(yo + 2 * (n0 + acc_row<T>(lane, r)));
        acc_t acc[STC];
#pragma unroll
        for (int s = 0; s < STC; ++s) acc[s] = s0[s];
        // (float64: the structurally zero slot tiles of the input rows are skipped, as in k_layer0_stats)
        const unsigned own = 1u | (1u << ((2 + 3 * i) >> 4)) | (1u << ((4 + 3 * i) >> 4));
        for (int ks = 0; ks < nks; ++ks) {
            const T wv = W0p[(size_t)(4 * ks) * Nout];
            T xv[STC];
#pragma unroll
            for (int s = 0; s < STC; ++s) xv[s] = Xl[(size_t)(4 * ks) * P + 16 * (t0 + s < ntile ? t0 + s : ntile - 1)];
            const unsigned m = sizeof(T) == 8 ? ((ks < h10 ? own : (own | prange[(ks - h10) >= h20 ? 1 : 0])) >> t0) : ~0u;
#pragma unroll
            for (int s = 0; s < STC; ++s)
                if ((m >> s) & 1) acc[s] = mfma16(wv, xv[s], acc[s]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const T y = yv[r][0], d1 = 1 - y * y;
#pragma unroll
            for (int s = 0; s < STC; ++s) {
                T o = d1 * acc[s][r];
                if (s == 0 && chunk == 0) o = lr == 0 ? y : (lr == 1 ? yv[r][1] : o);
                macc[s][r] += o;
            }
        }
    }
    const T inv = T(1) / T(ns);
    T* Mp = MEAN1 + ((size_t)w * S.nch + sp) * Nout * P + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + acc_row<T>(lane, r);
#pragma unroll
        for (int s = 0; s < STC; ++s)
            if (t0 + s < ntile) Mp[(size_t)n * P + 16 * (t0 + s)] = macc[s][r] * inv;
    }
}

// The two kernels above in one, for slot ranges that fit a wave's registers (ST <= 5) and NKS = K0loc / 4 known at compile time:
// a wave owns 16 features x all slot tiles and walks the electrons of its spin; the input rows of electron e + 1 are requested
// before the products of electron e.  (y, oL) -> YO, spin means of the dense output -> MEAN1; the dense output stays in registers.
//   grid (nch * Nout / 64, walkers), four waves.
template <typename T, int ST, int NKS>
__global__ void __launch_bounds__(256, 2)
k_layer0_stats(SysDev<T> S, const T* __restrict__ XL, size_t xl_ws, size_t xl_ts, const T* __restrict__ W0,
               const T* __restrict__ S0, int Nout, int P, T* __restrict__ YO, T* __restrict__ MEAN1) {
    typedef typename Acc4<T>::type acc_t;
    const int nfb = gridDim.x / S.nch, sp = blockIdx.x / nfb, fb = blockIdx.x - sp * nfb, w = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int n0 = (fb * 4 + wave) * 16;
    if (n0 >= Nout) return;
    const int i0 = sp == 0 ? 0 : S.n_up, ns = sp == 0 ? S.n_up : S.n_dn;
    T wv[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) wv[ks] = W0[(size_t)(4 * ks + lq) * Nout + n0 + lr];
    acc_t s0[ST], macc[ST];
#pragma unroll
    for (int t = 0; t < ST; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s0[t][r] = S0[(size_t)w * Nout * P + (size_t)(n0 + acc_row<T>(lane, r)) * P + 16 * t + lr];
        macc[t] = acc_t{0, 0, 0, 0};
    }
    const T* Xw = XL + (size_t)w * xl_ws + (size_t)lq * P + lr;
    T xn[NKS][ST];
    auto load_x = [&](int i) {
        const T* Xl = Xw + (size_t)i * xl_ts;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int t = 0; t < ST; ++t) xn[ks][t] = Xl[(size_t)(4 * ks) * P + 16 * t];
    };
    load_x(i0);
    const int h10 = S.h1[0] / 4, h20 = S.h2[0] / 4;      // k-steps of the own-feature rows / of one partner spin's pair-mean rows
    unsigned prange[2];
#pragma unroll
    for (int sp2 = 0; sp2 < 2; ++sp2) {
        const int j0 = sp2 == 0 ? 0 : S.n_up, nsp = sp2 == 0 ? S.n_up : (S.n_dn > 0 ? S.n_dn : S.n_up);
        const int lo = (2 + 3 * j0) >> 4, hi = (4 + 3 * (j0 + nsp - 1)) >> 4;
        prange[sp2] = ((2u << hi) - 1u) & ~((1u << lo) - 1u);
    }
    for (int e = 0; e < ns; ++e) {
        const int i = i0 + e;
        acc_t acc[ST];
#pragma unroll
        for (int t = 0; t < ST; ++t) acc[t] = s0[t];
        T xc[NKS][ST];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int t = 0; t < ST; ++t) xc[ks][t] = xn[ks][t];
        load_x(i0 + (e + 1 < ns ? e + 1 : e));          // (unconditional: the last electron is requested twice)
        // structural zeros of the layer-0 input rows (round 6; as in k_layer1_lr): the electron's own features fill slot tile 0 (value,
        // Laplacian) and its own tile(s), the pair-mean rows of partner spin sp also the partners' tiles; the other products add exact zeros
        const unsigned own = 1u | (1u << ((2 + 3 * i) >> 4)) | (1u << ((4 + 3 * i) >> 4));
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const unsigned m = ks < h10 ? own : (own | prange[(ks - h10) >= h20 ? 1 : 0]);
#pragma unroll
            for (int t = 0; t < ST; ++t)
                if ((m >> t) & 1) acc[t] = mfma16(wv[ks], xc[ks][t], acc[t]);
        }
        T* yo = YO + ((size_t)w * S.N + i) * 2 * Nout;
        // tanh of the four value slots in one evaluation (lane lr < 4 of every row takes row group lr)
        T zsel = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const T v = row16_bcast<0>(acc[0][r]);
            zsel = lr == r ? v : zsel;
        }
        const T yall = ds_tanh(zsel);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + acc_row<T>(lane, r);
            T ss = 0;
#pragma unroll
            for (int t = 0; t < ST; ++t) ss += (16 * t + lr >= 2) ? acc[t][r] * acc[t][r] : T(0);
            ss = row16_sum(ss);
            const T y = row16_bcast_dyn<4>(yall, r), d1 = 1 - y * y, d2 = -2 * y * d1;
            const T oL = d1 * row16_bcast<1>(acc[0][r]) + d2 * ss;
            if (lr < 2) yo[2 * n + lr] = lr == 0 ? y : oL;
#pragma unroll
            for (int t = 0; t < ST; ++t) {
                T o = d1 * acc[t][r];
                if (t == 0) o = lr == 0 ? y : (lr == 1 ? oL : o);
                macc[t][r] += o;
            }
        }
    }
    const T inv = T(1) / T(ns);
    T* Mp = MEAN1 + ((size_t)w * S.nch + sp) * Nout * P + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + acc_row<T>(lane, r);
#pragma unroll
        for (int t = 0; t < ST; ++t) Mp[(size_t)n * P + 16 * t] = macc[t][r] * inv;
    }
}

// W0T[n][c] (c < NCW): the layer-0 weight of input row c for output feature n -- per-electron rows, then the shared rows, zero beyond
template <typename T>
__global__ void k_lr_w0t(const T* __restrict__ Wloc0, const T* __restrict__ Wsh0, int K0loc, int K0sh, int Kh, int NCW, T* __restrict__ W0T) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Kh * NCW) return;
    const int n = idx / NCW, c = idx - n * NCW;
    W0T[idx] = c < K0loc ? Wloc0[(size_t)c * Kh + n] : (c < K0loc + K0sh ? Wsh0[(size_t)(c - K0loc) * Kh + n] : T(0));
}

// Shared spin-mean term of a hidden layer, S[n][slot] = sum_sp sum_k W_sh[sp*Kh + k][n] * mean_{i in sp} G[i][k][slot]
// (network.py:327-330: the tiled spin means of h_one), with the means formed on the fly:
// a workgroup (one walker, NW waves of 16*NB features) sums the n_s electron rows of a 16-row K chunk into
// LDS (each thread sums 32-byte pieces of the rows, fully coalesced), then every wave runs 4 k-steps on it.
template <typename T, int NB, int ST>
__global__ void __launch_bounds__((ST > 10 ? 512 : 1024 / NB), (NB == 4 ? 2 : 1))      // (ST > 10: at most eight waves, so that a wave may hold 16 NB x 16 ST accumulators)
k_shared_term(SysDev<T> S, const T* __restrict__ G, const T* __restrict__ Wsh, int Kh, T* __restrict__ Sb, int Nout, int P,
              const T* __restrict__ bias, int bias_all_slots) {
    typedef typename Acc4<T>::type acc_t;
    constexpr int KC = 16;
    extern __shared__ __attribute__((aligned(32))) char smem_raw[];
    T* mbuf = reinterpret_cast<T*>(smem_raw);            // [2][KC][P]
    const int w = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lq = lane >> 4, n0 = (blockIdx.z * (nthr >> 6) + wave) * 16 * NB;
    const bool active = n0 < Nout;                        // inactive waves still help with the means
    const T* Gw = G + (size_t)w * S.N * S.ldk * P;
    const int nchunk = S.nch * Kh / KC;                   // Kh is a multiple of 64
    acc_t acc[NB][ST];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int s = 0; s < ST; ++s) acc[a][s] = acc_t{0, 0, 0, 0};
    auto fill = [&](int c, T* dst) {
        const int sp = (c * KC) / Kh, k0 = (c * KC) % Kh;
        const int i0 = sp == 0 ? 0 : S.n_up, ns = sp == 0 ? S.n_up : S.n_dn;
        const T inv = T(1) / T(ns);
        const T* g0 = Gw + ((size_t)i0 * S.ldk + k0) * P;
        typedef T vec4 __attribute__((ext_vector_type(4)));          // P is a multiple of 16: every row is 128-byte aligned
        for (int e = tid; e < KC * P / 4; e += nthr) {
            vec4 v = {0, 0, 0, 0};
#pragma unroll 6
            for (int i = 0; i < ns; ++i) v += *reinterpret_cast<const vec4*>(g0 + (size_t)i * S.ldk * P + 4 * e);
            *reinterpret_cast<vec4*>(dst + 4 * e) = v * inv;
        }
    };
    fill(0, mbuf);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const T* cur = mbuf + (c & 1) * KC * P;
        if (c + 1 < nchunk) fill(c + 1, mbuf + ((c + 1) & 1) * KC * P);
        if (active) {
            const T* Wp = Wsh + (size_t)(c * KC + lq) * Nout + n0 + lr;
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                T av[NB], bv[ST];
#pragma unroll
                for (int a = 0; a < NB; ++a) av[a] = Wp[(size_t)(4 * ks) * Nout + 16 * a];
#pragma unroll
                for (int s = 0; s < ST; ++s) bv[s] = cur[(4 * ks + lq) * P + 16 * s + lr];
#pragma unroll
                for (int a = 0; a < NB; ++a)
#pragma unroll
                    for (int s = 0; s < ST; ++s) acc[a][s] = mfma16(av[a], bv[s], acc[a][s]);
            }
        }
        __syncthreads();
    }
    if (!active) return;
    T* Sp = Sb + (size_t)w * Nout * P;
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + 16 * a + acc_row<T>(lane, r);
            const T bn = bias ? bias[n] : T(0);        // the layer bias rides on S (value slot of the jets / every walker column)
#pragma unroll
            for (int s = 0; s < ST; ++s) Sp[(size_t)n * P + 16 * s + lr] = acc[a][s][r] + ((bias_all_slots || (s == 0 && lr == 0)) ? bn : T(0));
        }
}

}  // namespace ds
