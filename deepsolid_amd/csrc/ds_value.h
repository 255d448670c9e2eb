// ds_value.h -- value-only instantiation of the chain (log psi / orbitals for Metropolis).
//
// Same buffers and the same GEMM as the forward-Laplacian chain, but the contiguous "slot" axis
// carries PV = 80 different WALKERS (a "group") instead of the 3N+2 jet components of one walker:
//   G    [group][electron][row k][PV]     MEAN [group][spin][k][PV]     S [group][n][PV]
//   H2   [group*16 + c/5][k2][c%5][NP]    (the pair-stream kernel handles five columns per lane)
//   Q    [group][electron][p][re,im][PV]  envelope * Bloch phase (values)
//   MOUT [group][spin][det][elec][orb][re,im][PV]
// Walker column c of group g is walker min(g*PV + c, B-1) (the tail of the last group repeats the
// last walker; its results are never copied out).
#pragma once
#include "ds_gemm.h"
#include <utility>

namespace ds {

constexpr int PV = 80;

// value of [sd, rel_x, rel_y, rel_z] (network.py:207-224), no derivatives
template <typename T>
__device__ __forceinline__ void nu_distance_val(const T r[3], const T* __restrict__ av, const T* __restrict__ bv, int L,
                                                T out[4]) {
    const T pi = T(DS_PI);
    T f[6], g[6];
    for (int l = 0; l < L; ++l) {
        T w = r[0] * bv[3 * l] + r[1] * bv[3 * l + 1] + r[2] * bv[3 * l + 2];
        w = w - ds_floor((w + pi) / (2 * pi)) * 2 * pi;
        const T aw = ds_abs(w / pi);
        f[l] = ds_abs(w) * (1 - aw * aw * aw / 4);
        g[l] = w * (1 - T(1.5) * aw + T(0.5) * aw * aw);
    }
    T s2 = 0;
    for (int l = 0; l < L; ++l) {
        const T n2 = av[3 * l] * av[3 * l] + av[3 * l + 1] * av[3 * l + 1] + av[3 * l + 2] * av[3 * l + 2];
        s2 += n2 * f[l] * f[l];
        for (int m = 0; m < L; ++m)
            if (m != l) s2 += (av[3 * l] * av[3 * m] + av[3 * l + 1] * av[3 * m + 1] + av[3 * l + 2] * av[3 * m + 2]) * g[l] * g[m];
    }
    out[0] = ds_sqrt(s2);
    for (int c = 0; c < 3; ++c) {
        T rc = 0;
        for (int l = 0; l < L; ++l) rc += av[3 * l + c] * g[l];
        out[1 + c] = rc;
    }
}

// value of the 'tri' features (network.py:227-246): [sd, sin-rel (3), cos-rel (3)]
template <typename T>
__device__ __forceinline__ void tri_distance_val(const T r[3], const T* __restrict__ av, const T* __restrict__ bv, int L,
                                                 T out[7]) {
    T sn[6], cs[6];
    for (int l = 0; l < L; ++l) ds_sincos(r[0] * bv[3 * l] + r[1] * bv[3 * l + 1] + r[2] * bv[3 * l + 2], &sn[l], &cs[l]);
    T s2 = 0;
    for (int l = 0; l < L; ++l)
        for (int m = 0; m < L; ++m)
            s2 += (av[3 * l] * av[3 * m] + av[3 * l + 1] * av[3 * m + 1] + av[3 * l + 2] * av[3 * m + 2]) *
                  ((1 - cs[l]) * (1 - cs[m]) + sn[l] * sn[m]);
    out[0] = ds_sqrt(s2);
    for (int c = 0; c < 3; ++c) {
        T a = 0, b = 0;
        for (int l = 0; l < L; ++l) { a += av[3 * l + c] * sn[l]; b += av[3 * l + c] * cs[l]; }
        out[1 + c] = a;
        out[4 + c] = b;
    }
}

// Two launches, grid (groups, FV_SPLIT), block 256: PHASE 0 writes the one-electron features and the pair
// features, PHASE 1 (next launch, so the rows are visible) the spin means and Q.
constexpr int FV_SPLIT = 16;
template <typename T, int PHASE>
__global__ void __launch_bounds__(256) k_features_val(SysDev<T> S, const T* __restrict__ x, long B, const T* __restrict__ env_pi0,
                                                      const T* __restrict__ env_sg0, const T* __restrict__ env_pi1,
                                                      const T* __restrict__ env_sg1, T* __restrict__ G, T* __restrict__ MEAN,
                                                      T* __restrict__ H2, T* __restrict__ Q) {
    const int g = blockIdx.x, tid = blockIdx.y * blockDim.x + threadIdx.x, nt = blockDim.x * gridDim.y;
    const int N = S.N, A = S.A, NP = S.NP, K1 = S.h1[0], nf = S.nf;
    T* Gw = G + (size_t)g * N * S.ldk * PV;
    auto walker = [&](int c) { long wi = (long)g * PV + c; return wi < B ? wi : B - 1; };
    if (PHASE == 0) {
    for (int idx = tid; idx < N * A * PV; idx += nt) {
        const int c = idx % PV, a = (idx / PV) % A, i = idx / (PV * A);
        const T* xp = x + (size_t)walker(c) * 3 * N + 3 * i;
        T r[3] = {xp[0], xp[1], xp[2]}, o[3], wr[3], f[7];
        wrap_point(r, S.prim_a, S.prim_ainv, o, wr);
        for (int k = 0; k < 3; ++k) o[k] -= S.atoms[3 * a + k];
        if (S.dist_type == 0) nu_distance_val(o, S.prim_AV, S.prim_BV, S.L, f);
        else tri_distance_val(o, S.prim_AV, S.prim_BV, S.L, f);
        for (int k = 0; k < nf; ++k) Gw[((size_t)i * S.ldk + nf * a + k) * PV + c] = f[k];
    }
    for (int idx = tid; idx < N * (K1 - nf * A) * PV; idx += nt) {       // zero padding rows
        const int c = idx % PV, k = nf * A + (idx / PV) % (K1 - nf * A), i = idx / (PV * (K1 - nf * A));
        Gw[((size_t)i * S.ldk + k) * PV + c] = 0;
    }
    for (int idx = tid; idx < PV * NP; idx += nt) {
        const int q = idx % NP, c = idx / NP;
        const int e = q / N, j = q % N;
        T f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q < N * N && e != j) {
            const T* xw = x + (size_t)walker(c) * 3 * N;
            T rj[3] = {xw[3 * j], xw[3 * j + 1], xw[3 * j + 2]}, re[3] = {xw[3 * e], xw[3 * e + 1], xw[3 * e + 2]}, oj[3], oe[3], wr[3];
            wrap_point(rj, S.sim_a, S.sim_ainv, oj, wr);
            wrap_point(re, S.sim_a, S.sim_ainv, oe, wr);
            for (int k = 0; k < 3; ++k) oj[k] -= oe[k];
            if (S.dist_type == 0) nu_distance_val(oj, S.sim_AV, S.sim_BV, S.L, f);
            else tri_distance_val(oj, S.sim_AV, S.sim_BV, S.L, f);
        }
        T* Hw = H2 + (size_t)(g * (PV / 5) + c / 5) * S.h2[0] * 5 * NP;
        for (int k = 0; k < S.h2[0]; ++k) Hw[(size_t)(k * 5 + c % 5) * NP + q] = f[k];
    }
    return;
    }
    T* Mw = MEAN + (size_t)g * S.nch * K1 * PV;
    for (int idx = tid; idx < S.nch * K1 * PV; idx += nt) {
        const int c = idx % PV, k = (idx / PV) % K1, s = idx / (PV * K1);
        const int i0 = s == 0 ? 0 : S.n_up, ns = s == 0 ? S.n_up : S.n_dn;
        T v = 0;
        for (int i = i0; i < i0 + ns; ++i) v += Gw[((size_t)i * S.ldk + k) * PV + c];
        Mw[idx] = v / T(ns);
    }
    T* Qw = Q + (size_t)g * N * S.nparam_max * 2 * PV;
    // q[i][p] = envelope(i, p) * exp(i k_m . x_i), p = (determinant, orbital m): the Bloch phase belongs to the orbital's k-point, one
    // sincos per (electron, m, walker) serves all determinants
    const int norb_max = S.norb[0] > S.norb[S.nch - 1] ? S.norb[0] : S.norb[S.nch - 1];
    for (int idx = tid; idx < N * norb_max * PV; idx += nt) {
        const int c = idx % PV, m = (idx / PV) % norb_max, i = idx / (PV * norb_max);
        const int s = spin_of(i, S.n_up), np = S.nparam[s], no = S.norb[s];
        if (m >= no) continue;
        const T* pi_ = s == 0 ? env_pi0 : env_pi1;
        const T* sg_ = s == 0 ? env_sg0 : env_sg1;
        const T* kv = S.klist[s] + 3 * m;
        const T* xp = x + (size_t)walker(c) * 3 * N + 3 * i;
        T sn, cs;
        ds_sincos(kv[0] * xp[0] + kv[1] * xp[1] + kv[2] * xp[2], &sn, &cs);
        for (int p = m; p < np; p += no) {
            T e = 0;
            for (int a = 0; a < A; ++a) {
                const T* f = Gw + ((size_t)i * S.ldk + nf * a) * PV + c;    // rows sd, rel_x, rel_y, rel_z (, cos-rel) of atom a
                T r;
                if (S.env_type == 0) r = ds_abs(f[0] * sg_[a * np + p]);
                else {
                    T r2 = 0;
                    for (int mm = 0; mm < 3; ++mm) {
                        T u = 0;
                        if (S.env_type == 1) u = sg_[(a * 3 + mm) * np + p] * f[(size_t)(1 + mm) * PV];
                        else
                            for (int k = 0; k < 3; ++k) u += sg_[((k * 3 + mm) * A + a) * np + p] * f[(size_t)(1 + k) * PV];
                        r2 += u * u;
                    }
                    r = ds_sqrt(r2);
                }
                e += pi_[a * np + p] * ds_exp(-r);
            }
            Qw[((size_t)(i * S.nparam_max + p) * 2) * PV + c] = e * cs;
            Qw[((size_t)(i * S.nparam_max + p) * 2 + 1) * PV + c] = e * sn;
        }
    }
}

// rows [row0, row0 + nch*K2) of G: mean over the partners j of spin s of h2[j][e] (values).
// Eight lanes share one (k, walker) row of N contiguous partners and reduce with shuffles.
template <typename T>
__global__ void __launch_bounds__(256) k_m2_expand_val(SysDev<T> S, const T* __restrict__ H2, int K2, T* __restrict__ G, int row0) {
    const int e = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, N = S.N, NP = S.NP;
    const int part = tid & 7;
    T* Ge = G + ((size_t)(g * N + e) * S.ldk + row0) * PV;
    for (int r0 = 0; r0 < K2 * PV; r0 += 32) {
        const int r = r0 + (tid >> 3);                       // K2 * PV is a multiple of 32
        const int c = r % PV, k = r / PV;
        const T* hp = H2 + ((size_t)(g * (PV / 5) + c / 5) * K2 * 5 + (size_t)(k * 5 + c % 5)) * NP + (size_t)e * N;
        T up = 0, dn = 0;
        for (int j = part; j < N; j += 8) {
            const T v = hp[j];
            if (j < S.n_up) up += v; else dn += v;
        }
        // 8-lane sums on DPP: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
        up += dpp_mov<0xB1>(up); up += dpp_mov<0x4E>(up); up += dpp_mov<0x141>(up);
        dn += dpp_mov<0xB1>(dn); dn += dpp_mov<0x4E>(dn); dn += dpp_mov<0x141>(dn);
        if (part == 0) Ge[(size_t)k * PV + c] = up / T(S.n_up);
        if (part == 1 && S.nch > 1) Ge[(size_t)(K2 + k) * PV + c] = dn / T(S.n_dn);
    }
}

// The whole two-electron stream of a log-psi forward in ONE launch (network.py:525-528 for every pair layer): the pair stream never
// reads the one-electron stream, so a wave that owns 16 pairs x 5 walkers can run layer after layer on them without the
// activations leaving its registers.  The accumulator layout of the 16x16x4 MFMA (rows = output features, columns = pairs) IS the
// layout of its B operand (k = features), so layer l's output feeds layer l + 1's product directly: step (a, r) contracts the four
// features 16 a + acc_row(lane, r) with the matching weight rows -- in float64 the same k order as k_two_layer, so the
// activations are bit-identical to the layer-by-layer kernels.  What leaves the kernel is only what the one-electron layers need:
// per layer the running sums of the tile's (electron, partner spin) segments, PM[l] laid out exactly as k_two_layer's PARTM
// ([5-walker block][tile][slot][feature][column]; k_m2_combine_val reads it).  The sums over the 16 pairs of a tile are taken
// on the matrix pipe: a layer's outputs are staged [column][feature][pair] through LDS, read back in the A-operand layout and
// multiplied with the indicator matrix [pair < end of segment q] (exact products; details at `ind` below) in place of 40
// sixteen-lane DPP prefix sums of ~16 instructions each; with no per-value control flow left the tanh chains of a column are one
// basic block (their constants are materialised once -- in k_two_layer every value re-materialises them: 75 instead of 45 VALU
// instructions per tanh -- and the chains interleave).
// The kernel is bound by the FP64 datapath, which the float64 MFMA SHARES with the float64 VALU on gfx950 (a 16x16x4 product
// occupies it for 64 cycles: SQ_VALU_MFMA_BUSY_CYCLES 34 % + SQ_ACTIVE_INST_VALU 53 % of the kernel's cycles, and removing the
// tanh, the sums or the stores from the kernel shortens it by exactly their own cycles -- nothing overlaps, EXPERIMENTS.md
// round 5): per tile 80 tanh x 45 instructions x 4 cycles = 14.4 k cycles, 100 product MFMAs = 6.4 k, 20 segment-sum products.
// Replaces k_two_layer<.., VAL> x n_double and the H2 round trips between them in the Metropolis forward: bcc-Li 4096 walkers,
// 0.372 + 0.239 ms -> 0.401 ms.  The gradient pass (activations kept) still runs the layer-by-layer kernels.
// grid ((NP / 16 + 3) / 4, 5-walker blocks), block 256 (a wave per pair tile; no workgroup barrier).
constexpr int PS_MAX_LAYERS = 8;
template <typename T> struct PairStreamArgs {
    const T* H0; int Kin0, nl;                                   // layer-0 input [block][Kin0][5][NP] (k_features_val), number of layers
    const T* W[PS_MAX_LAYERS]; const T* b[PS_MAX_LAYERS]; int res[PS_MAX_LAYERS];   // res[0] must be 0
    T* PM[PS_MAX_LAYERS];
};
template <typename T, int NT2>
__global__ void __launch_bounds__(256, 2) k_pair_stream_val(SysDev<T> S, PairStreamArgs<T> A) {
    typedef typename Acc4<T>::type acc_t;
    constexpr int Kout = 16 * NT2, LD = sizeof(T) == 8 ? 18 : 20;     // LD: conflict-free A-operand reads (rows 16 apart in lanes)
    constexpr int PMN = PM_SLOTS * Kout * 5;                     // segment sums of a tile: [slot][feature][column]
    // per wave: 80 staging rows [column][feature] x 16 pairs (one 16-feature half of a layer's output), the tile's sums, and a
    // dump area of a slot's extent (+ one element per lane) for the lanes that hold no slot
    constexpr int WLDS = 80 * LD + PMN + Kout * 5 + 64;
    __shared__ T ps_lds[4 * WLDS];
    const int w = blockIdx.y, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + wave, NP = S.NP, N = S.N;
    if (pt * 16 >= NP) return;
    const int lr = lane & 15, lq = lane >> 4, p0 = pt * 16;
    T* stage = ps_lds + wave * WLDS;
    T* pml = stage + 80 * LD;
    // ends (exclusive, relative to p0) of the first two (electron, partner spin) segments that meet this tile; the third runs to the end
    auto next_end = [&](int p) { const int e = p / N, j = p - e * N; return (S.nch > 1 && j < S.n_up) ? e * N + S.n_up : (e + 1) * N; };
    const int b1 = next_end(p0), b2 = next_end(b1);
    const int e1 = b1 - p0 < 16 ? b1 - p0 : 16, e2 = b2 - p0 < 16 ? b2 - p0 : 16;
    // Segment sums of a 16-feature x 16-pair tile V on the matrix pipe: PM[feature][slot q] = sum_pair V[feature][pair] * [pair < end_q].
    // float64: v_mfma_f64_4x4x4_4b (four 4x4x4 blocks = the four groups of four features; 16 cycles against the 64 of the
    // 16x16x4 form, whose 16 output columns would hold 3 slots -- the float64 MFMA runs on the same FP64 datapath as the tanh
    // polynomials, so its cycles are not free): a = V[feature lane & 15][pair 4 s + (lane >> 4)] (the same LDS read as a 16x16x4 A
    // operand), b = ind[pair 4 s + (lane >> 4)][slot lane & 3], d = PM[feature 4 ((lane >> 2) & 3) + (lane >> 4)][slot lane & 3]:
    // one value per lane (tools/probes/mfma44_layout.hip).  float32: the 16x16x4 form, slot = column lane & 15.
    constexpr bool M44 = sizeof(T) == 8;
    const int slot = M44 ? (lane & 3) : lr;
    T ind[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int pr = 4 * s4 + lq, lim = slot == 0 ? e1 : (slot == 1 ? e2 : 16);
        ind[s4] = (slot < PM_SLOTS && p0 + pr < N * N && pr < lim) ? T(1) : T(0);
    }
    // where this lane's sums go in pml: its slot, or (lanes without one) the dump area behind the sums, one element apart per lane
    T* pm_dst = pml + (slot < PM_SLOTS ? slot * Kout * 5 : PMN + lane);
    const int f44 = 4 * ((lane >> 2) & 3) + lq;                  // float64: the feature (within the half) of this lane's sum
    const T rs2 = T(0.70710678118654752440);
    // D: the layer's input / output, all Kout features (the B operand of the next layer).  A layer is produced one 16-feature
    // half at a time (acc: 20 accumulators instead of 40): half 0's output waits in the staging rows -- where the segment sums read
    // it anyway -- until half 1's product has consumed D, and only then replaces D[0].  120 live accumulator registers in place
    // of 160 leave the scheduler room to interleave the tanh chains of a unit with the LDS / MFMA latency of the unit before.
    acc_t D[NT2][5], acc[5];
    // One half of a layer's epilogue on `z` (pre-activations, features 16 a ..): v = tanh(z + b) (+ residual D[a]) -> `out` (z itself
    // or D[a]) and the staging rows; then the segment sums of the half: per column the 16 x 16 tile [feature][pair] is read back as the
    // A operand and multiplied with `ind` (four MFMAs), the sums land in pml.  No control flow and no fences inside: the LDS
    // operations of a wave execute in order and the compiler keeps the order of may-alias accesses.
    auto half_epilogue = [&](auto res_tag, int a, acc_t (&z)[5], acc_t (&resid)[5], acc_t (&out)[5], const T* __restrict__ bias) {
        constexpr bool RES = decltype(res_tag)::value;
        T bn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bn[r] = bias[16 * a + acc_row<T>(lane, r)];
        // software pipeline over the five columns, spelled out (the compiler keeps LDS operations in source order and would put each
        // unit's read -> MFMA -> write chain right behind its tanh): step c does the tanh of column c with the four MFMAs of column
        // c - 1 between them (operands read from LDS a step earlier) and stores the sums of column c - 2
        T ar[2][4];
        typename std::conditional<M44, T, acc_t>::type sm[2];
        auto seg_mfma = [&](int u, int r) {
            if constexpr (M44) sm[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(ar[u][r], ind[r], r == 0 ? 0.0 : sm[u], 0, 0, 0);
            else sm[u] = mfma16(ar[u][r], ind[r], r == 0 ? acc_t{0, 0, 0, 0} : sm[u]);
        };
        auto seg_store = [&](int u, int c) {
            if constexpr (M44) pm_dst[(16 * a + f44) * 5 + c] = sm[u];
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) pm_dst[(16 * a + acc_row<T>(lane, r)) * 5 + c] = sm[u][r];
            }
        };
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            if (c < 5) {
                T* st = stage + c * (16 * LD);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int r = 2 * h; r < 2 * h + 2; ++r) {
                        T v = ds_tanh(z[c][r] + bn[r]);
                        if (RES) v = (resid[c][r] + v) * rs2;
                        out[c][r] = v;
                        st[acc_row<T>(lane, r) * LD + lr] = v;
                        if (c >= 1) seg_mfma((c - 1) & 1, r);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (c == 5) {
#pragma unroll
                for (int r = 0; r < 4; ++r) seg_mfma(0, r);
            }
            if (c >= 2) seg_store((c - 2) & 1, c - 2);
            if (c < 5) {
                const T* st = stage + c * (16 * LD);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) ar[c & 1][s4] = st[lr * LD + 4 * s4 + lq];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto reload_half0 = [&]() {                                   // D[0] <- the staging rows (each lane reads back what it wrote)
#pragma unroll
        for (int c = 0; c < 5; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) D[0][c][r] = stage[(c * 16 + acc_row<T>(lane, r)) * LD + lr];
    };
    auto flush_sums = [&](T* __restrict__ pm) {
#pragma unroll
        for (int i = 0; i < (PMN + 63) / 64; ++i)
            if (64 * i + lane < PMN) pm[64 * i + lane] = pml[64 * i + lane];
    };
    for (int l = 0; l < A.nl; ++l) {
        const T* W = A.W[l];
        const T* bias = A.b[l];
        const bool res = A.res[l] != 0;
#pragma unroll
        for (int a2 = 0; a2 < NT2; ++a2) {
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[c] = acc_t{0, 0, 0, 0};
            if (l == 0) {
                const T* Hw = A.H0 + (size_t)w * A.Kin0 * 5 * NP + p0 + lr;
                for (int ks = 0; ks < A.Kin0 / 4; ++ks) {
                    const T av = W[(size_t)(4 * ks + lq) * Kout + 16 * a2 + lr];
                    T bv[5];
#pragma unroll
                    for (int c = 0; c < 5; ++c) bv[c] = Hw[(size_t)((4 * ks + lq) * 5 + c) * NP];
#pragma unroll
                    for (int c = 0; c < 5; ++c) acc[c] = mfma16(av, bv[c], acc[c]);
                }
            } else {
#pragma unroll
                for (int a = 0; a < NT2; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const T av = W[(size_t)(16 * a + acc_row<T>(lane, r)) * Kout + 16 * a2 + lr];
#pragma unroll
                        for (int c = 0; c < 5; ++c) acc[c] = mfma16(av, D[a][c][r], acc[c]);
                    }
            }
            // half a2 < NT2 - 1 leaves its output in acc (and the staging rows); the last half writes D[a2] in place
            if (NT2 == 2 && a2 == 0) {
                if (res) half_epilogue(std::true_type{}, 0, acc, D[0], acc, bias);
                else half_epilogue(std::false_type{}, 0, acc, D[0], acc, bias);
            } else {
                if (NT2 == 2) reload_half0();
                if (res) half_epilogue(std::true_type{}, a2, acc, D[a2], D[a2], bias);
                else half_epilogue(std::false_type{}, a2, acc, D[a2], D[a2], bias);
            }
        }
        flush_sums(A.PM[l] + ((size_t)w * (NP / 16) + pt) * PMN);
    }
}

// log-determinant of every (walker, det) matrix of one determinant channel WITHOUT the inverse (Metropolis / log psi only):
// LU with partial pivoting, FOUR LANES PER MATRIX -- lane p of a quad keeps rows R*p .. R*p+R-1 in registers (n <= 4R <= 16),
// so a wave factorises 16 walkers' matrices at once with no LDS and no barriers.  The 16 walkers of a wave are consecutive
// columns of one group: every load is four 128-byte segments.  Pivoting is implicit (no row exchanges): step k takes the
// largest |a[r][k]| among the rows not used yet, its owner broadcasts the row inside the quad, every other unused row is
// reduced; det = sgn(order) * prod pivots, the sign from the number of unused rows skipped at each step.
// Rows / columns n .. 4R-1 are padded with the identity.  Replaces jnp.linalg.slogdet at network.py:390 for the value chain.
// The determinant is carried as a complex product with its binary exponent apart (one rescale per pivot: no sqrt / log / division
// per step); log|det| and the phase are taken once at the end.
// grid (K, ceil(B / 64), channels), block 256 (4 waves x 16 walkers); channel sp0 + blockIdx.z uses offsets off.mout / off.dets [blockIdx.z].
struct DetOff2 { size_t mout[2], dets[2]; };
__device__ __forceinline__ int ds_frexp_exp(float x) { return __builtin_amdgcn_frexp_expf(x); }
__device__ __forceinline__ int ds_frexp_exp(double x) { return __builtin_amdgcn_frexp_exp(x); }
__device__ __forceinline__ float ds_ldexp(float x, int e) { return ldexpf(x, e); }
__device__ __forceinline__ double ds_ldexp(double x, int e) { return ldexp(x, e); }
template <typename T, int R>
__global__ void __launch_bounds__(256) k_det_lu_val(SysDev<T> S, const T* __restrict__ MOUT, size_t mout_stride, DetOff2 off, int sp0,
                                                    long B, T* __restrict__ DETS, size_t dets_stride) {
    constexpr int NC = 4 * R;
    const int sp = sp0 + blockIdx.z;
    const size_t mout_off = off.mout[blockIdx.z], dets_off = off.dets[blockIdx.z];
    const int kdet = blockIdx.x, lane = threadIdx.x & 63, p = lane & 3;
    const long w = ((long)blockIdx.y * 4 + (threadIdx.x >> 6)) * 16 + (lane >> 2);
    const long wc = w < B ? w : B - 1;                   // the tail repeats the last walker (never stored)
    const int n = S.det_n[sp];
    const T* Mw = MOUT + (size_t)(wc / PV) * mout_stride + mout_off + (size_t)kdet * n * n * 2 * PV + wc % PV;
    Cx<T> a[R][NC];
#pragma clang loop unroll(full)
    for (int rr = 0; rr < R; ++rr) {
        const int r = R * p + rr;
#pragma clang loop unroll(full)
        for (int m = 0; m < NC; ++m) {
            if (r < n && m < n) a[rr][m] = Cx<T>(Mw[(size_t)((r * n + m) * 2) * PV], Mw[(size_t)((r * n + m) * 2 + 1) * PV]);
            else a[rr][m] = Cx<T>(r == m ? T(1) : T(0), T(0));
        }
    }
    unsigned used = 0;                                     // bit r: row r has been a pivot (the same in the four lanes)
    Cx<T> ph(1, 0);                                        // det = ph * 2^pe
    int pe = 0;
    const int qbase = lane & ~3;
#pragma clang loop unroll(full)
    for (int k = 0; k < NC; ++k) {
        T best = -1;
        int bi = NC;
#pragma clang loop unroll(full)
        for (int rr = 0; rr < R; ++rr) {
            const int r = R * p + rr;
            const T m2 = cx_abs2(a[rr][k]);
            if (!((used >> r) & 1u) && m2 > best) { best = m2; bi = r; }
        }
#pragma clang loop unroll(full)
        for (int off = 1; off < 4; off <<= 1) {            // quad maximum, ties to the lower row
            const T ob = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        // every remaining candidate NaN (m2 > best is false for all of them): take the lowest unused row, so that the NaN
        // becomes the pivot and log|det| comes out NaN like slogdet / k_det_inverse -- a move into such a configuration is
        // then rejected (lp2 - lp1 > log u is false) instead of being accepted with a finite garbage value
        if (bi >= NC) bi = __ffs((int)(~used)) - 1;
        const int owner = qbase | (bi / R), orr = bi % R;
        if (__popc(~used & ((1u << bi) - 1u)) & 1) ph = Cx<T>(-ph.re, -ph.im);      // unused rows skipped: parity of the order
        used |= 1u << bi;
        Cx<T> pv[NC];
#pragma clang loop unroll(full)
        for (int m = k; m < NC; ++m) {
            Cx<T> mine = a[0][m];
#pragma clang loop unroll(full)
            for (int rr = 1; rr < R; ++rr)
                if (orr == rr) mine = a[rr][m];
            pv[m] = Cx<T>(__shfl(mine.re, owner), __shfl(mine.im, owner));
        }
        ph = ph * pv[k];
        {
            const T big = ds_abs(ph.re) > ds_abs(ph.im) ? ds_abs(ph.re) : ds_abs(ph.im);      // (NaN: the exponent is ignored, NaN stays)
            const int e = big > T(0) ? ds_frexp_exp(big) : 0;
            ph = Cx<T>(ds_ldexp(ph.re, -e), ds_ldexp(ph.im, -e));
            pe += e;
        }
        const Cx<T> dinv = cx_inv(pv[k]);
#pragma clang loop unroll(full)
        for (int rr = 0; rr < R; ++rr) {
            const int r = R * p + rr;
            if (!((used >> r) & 1u)) {
                const Cx<T> f = a[rr][k] * dinv;
                const Cx<T> nf(-f.re, -f.im);
#pragma clang loop unroll(full)
                for (int m = k + 1; m < NC; ++m) a[rr][m] = cx_fma(nf, pv[m], a[rr][m]);
            }
        }
    }
    if (p == 0 && w < B) {
        T* dw = DETS + (size_t)w * dets_stride + dets_off + (size_t)kdet * 4;
        dw[0] = T(0.5) * ds_log(cx_abs2(ph)) + T(pe) * T(0.69314718055994530942);
        dw[1] = ds_atan2(ph.im, ph.re);
    }
}

// The same for 16 < n <= NC <= 64: ONE LANE PER ROW -- a wave factorises one matrix, lane r keeps row r (NC complex values) in
// registers.  Step k: every unused lane offers |a[r][k]|^2, a wave arg-max picks the pivot row (ties to the lower row; implicit
// pivoting, no exchanges: the sign comes from the unused rows skipped), its lane broadcasts a[p][k..] through v_readlane -- the
// pivot row sits in SCALAR registers for the update --, every other unused lane reduces its row.  n^3 / 3 complex multiply-adds
// against the 2 n^3 of the Gauss-Jordan inverse on [M | 1] (k_det_inverse, one wave per matrix, LDS, a barrier pair per step)
// that a log-psi forward used to run for these sizes although it needs no inverse: 4.48 -> 0.x ms per launch at 48 x 48 (diamond).
// Rows / columns n .. NC-1 do not exist (lanes >= n are marked used from the start; the loop stops at n).
// grid (K, B), block 64.
template <typename T> __device__ __forceinline__ T wave_bcast(T x, int src);
template <> __device__ __forceinline__ float wave_bcast<float>(float x, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src));
}
template <> __device__ __forceinline__ double wave_bcast<double>(double x, int src) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// (the elimination steps are a function template over K, expanded with an integer sequence: a `for k` loop of NC steps x NC
//  columns is past the unroller's size limit, stays a loop, and the row array -- then indexed dynamically -- lands in scratch)
// det = (product of the pivots, kept as a complex number rescaled by its binary exponent at every step) * 2^esum: one log and
// one atan2 per matrix instead of a log, a square root and two divisions per pivot
template <typename T, int NC> struct LuState {
    unsigned long long used;
    int esum;
    Cx<T> ph;
};
template <typename T, int NC, int K>
__device__ __forceinline__ void lu_wave_step(Cx<T> (&a)[NC], LuState<T, NC>& st, int n, int lane) {
    if (K >= n) return;
    const bool free_row = !((st.used >> lane) & 1ull);
    const T m2 = cx_abs2(a[K]);
    // pivot = the unused row with the largest |a[r][K]|^2, the lowest such row on a tie: maximum over each 16-lane row on DPP,
    // over the four rows through v_readlane, then a ballot of the lanes that hold it (the butterfly of ds_bpermute shuffles
    // this replaces cost 18 trips through the LDS crossbar per step in float64)
    T best = free_row ? m2 : T(-1);
    best = best > T(-1) ? best : T(-1);                            // NaN candidates do not take part
    {
        T o;
        o = dpp_mov<0xB1>(best);  best = o > best ? o : best;
        o = dpp_mov<0x4E>(best);  best = o > best ? o : best;
        o = dpp_mov<0x141>(best); best = o > best ? o : best;
        o = dpp_mov<0x140>(best); best = o > best ? o : best;
        const T r0 = wave_bcast(best, 0), r1 = wave_bcast(best, 16), r2 = wave_bcast(best, 32), r3 = wave_bcast(best, 48);
        const T m01 = r1 > r0 ? r1 : r0, m23 = r3 > r2 ? r3 : r2;
        best = m23 > m01 ? m23 : m01;
    }
    const unsigned long long hit = __ballot(free_row && m2 == best);
    // every remaining candidate NaN (nobody holds the maximum -1): the lowest unused row becomes the pivot, log|det| comes out NaN
    int bi = hit ? __ffsll((long long)hit) - 1 : __ffsll((long long)(~st.used)) - 1;
    const int owner = __builtin_amdgcn_readfirstlane(bi);
    if (__popcll(~st.used & ((1ull << owner) - 1ull)) & 1) st.ph = Cx<T>(-st.ph.re, -st.ph.im);
    st.used |= 1ull << owner;
    const Cx<T> pk(wave_bcast(a[K].re, owner), wave_bcast(a[K].im, owner));
    st.ph = st.ph * pk;
    {
        const T big = ds_abs(st.ph.re) > ds_abs(st.ph.im) ? ds_abs(st.ph.re) : ds_abs(st.ph.im);      // (NaN: the exponent is ignored, NaN stays)
        const int e = big > T(0) ? ds_frexp_exp(big) : 0;
        st.ph = Cx<T>(ds_ldexp(st.ph.re, -e), ds_ldexp(st.ph.im, -e));
        st.esum += e;
    }
    const Cx<T> f = a[K] * cx_inv(pk);
    const Cx<T> nf(-f.re, -f.im);
    const bool upd = !((st.used >> lane) & 1ull);
#pragma unroll
    for (int m = K + 1; m < NC; ++m) {
        if (m < n) {
            const Cx<T> pm(wave_bcast(a[m].re, owner), wave_bcast(a[m].im, owner));
            const Cx<T> v = cx_fma(nf, pm, a[m]);
            a[m].re = upd ? v.re : a[m].re;        // (component-wise: a select between two 16-byte structs is lowered through
            a[m].im = upd ? v.im : a[m].im;        //  pointers and a memcpy, which keeps the whole row array in scratch memory)
        }
    }
}
template <typename T, int NC, int... Ks>
__device__ __forceinline__ void lu_wave_steps(std::integer_sequence<int, Ks...>, Cx<T> (&a)[NC], LuState<T, NC>& st, int n, int lane) {
    (lu_wave_step<T, NC, Ks>(a, st, n, lane), ...);
}
template <typename T, int NC>
__global__ void __launch_bounds__(64) k_det_lu_wave(SysDev<T> S, const T* __restrict__ MOUT, size_t mout_stride, size_t mout_off, int sp,
                                                    long B, T* __restrict__ DETS, size_t dets_stride, size_t dets_off) {
    const int kdet = blockIdx.x, lane = threadIdx.x;
    const long w = blockIdx.y;
    const int n = S.det_n[sp];
    const T* Mw = MOUT + (size_t)(w / PV) * mout_stride + mout_off + (size_t)kdet * n * n * 2 * PV + w % PV;
    Cx<T> a[NC];
#pragma unroll
    for (int m = 0; m < NC; ++m) {
        if (lane < n && m < n) a[m] = Cx<T>(Mw[(size_t)((lane * n + m) * 2) * PV], Mw[(size_t)((lane * n + m) * 2 + 1) * PV]);
        else a[m] = Cx<T>(T(0), T(0));
    }
    LuState<T, NC> st{n < 64 ? (~0ull << n) : 0ull, 0, Cx<T>(1, 0)};      // used: bit r = row r has been a pivot (or does not exist)
    lu_wave_steps<T, NC>(std::make_integer_sequence<int, NC>(), a, st, n, lane);
    if (lane == 0 && w < B) {
        T* dw = DETS + (size_t)w * dets_stride + dets_off + (size_t)kdet * 4;
        dw[0] = T(0.5) * ds_log(cx_abs2(st.ph)) + T(st.esum) * T(0.69314718055994530942);
        dw[1] = ds_atan2(st.ph.im, st.ph.re);
    }
}

// Inverse + log det by in-place Gauss-Jordan with one lane per matrix row (the forward-Laplacian chain: the determinant-trace
// kernels need M^-1).  Same pivot rule as k_det_lu_wave (largest |a[r][K]| among the unused rows, lowest row on a tie, no row
// exchange); at step K every other row is reduced and column K is overwritten by the new column of the inverse that belongs to
// the pivot row p_K.  In the end lane r = p_i holds row i of the inverse with its columns in pivot order: Minv[i][p_K] = a[K].
//   MOUT element (row, col, re/im) at ((row n + col) 2 + re/im) es  (es = 16: slot 0 of slot tile 0);  MINV [det][n][n][re,im].
template <typename T, int NC, int K>
__device__ __forceinline__ void gj_wave_step(Cx<T> (&a)[NC], LuState<T, NC>& st, int (&owners)[NC], int& mystep, int n, int lane) {
    if (K >= n) return;
    const bool free_row = !((st.used >> lane) & 1ull);
    const T m2 = cx_abs2(a[K]);
    T best = free_row ? m2 : T(-1);
    best = best > T(-1) ? best : T(-1);
    {
        T o;
        o = dpp_mov<0xB1>(best);  best = o > best ? o : best;
        o = dpp_mov<0x4E>(best);  best = o > best ? o : best;
        o = dpp_mov<0x141>(best); best = o > best ? o : best;
        o = dpp_mov<0x140>(best); best = o > best ? o : best;
        const T r0 = wave_bcast(best, 0), r1 = wave_bcast(best, 16), r2 = wave_bcast(best, 32), r3 = wave_bcast(best, 48);
        const T m01 = r1 > r0 ? r1 : r0, m23 = r3 > r2 ? r3 : r2;
        best = m23 > m01 ? m23 : m01;
    }
    const unsigned long long hit = __ballot(free_row && m2 == best);
    int bi = hit ? __ffsll((long long)hit) - 1 : __ffsll((long long)(~st.used)) - 1;
    const int owner = __builtin_amdgcn_readfirstlane(bi);
    if (__popcll(~st.used & ((1ull << owner) - 1ull)) & 1) st.ph = Cx<T>(-st.ph.re, -st.ph.im);
    st.used |= 1ull << owner;
    owners[K] = owner;
    mystep = lane == owner ? K : mystep;
    const Cx<T> pk(wave_bcast(a[K].re, owner), wave_bcast(a[K].im, owner));
    st.ph = st.ph * pk;
    {
        const T big = ds_abs(st.ph.re) > ds_abs(st.ph.im) ? ds_abs(st.ph.re) : ds_abs(st.ph.im);
        const int e = big > T(0) ? ds_frexp_exp(big) : 0;
        st.ph = Cx<T>(ds_ldexp(st.ph.re, -e), ds_ldexp(st.ph.im, -e));
        st.esum += e;
    }
    const Cx<T> dinv = cx_inv(pk);
    const bool piv = lane == owner;
    // pivot row: a[m] / d (column K: 1 / d);  other rows: a[m] - f (a_p[m] / d) (column K: -f / d), f = a[K]
    const Cx<T> f = a[K];
    const Cx<T> nf(-f.re, -f.im);
#pragma unroll
    for (int m = 0; m < NC; ++m) {
        if (m < n && m != K) {
            const Cx<T> pm = Cx<T>(wave_bcast(a[m].re, owner), wave_bcast(a[m].im, owner)) * dinv;       // scaled pivot row (uniform)
            const Cx<T> v = cx_fma(nf, pm, a[m]);
            a[m].re = piv ? pm.re : v.re;
            a[m].im = piv ? pm.im : v.im;
        }
    }
    const Cx<T> ck = nf * dinv;
    a[K].re = piv ? dinv.re : ck.re;
    a[K].im = piv ? dinv.im : ck.im;
}
template <typename T, int NC, int... Ks>
__device__ __forceinline__ void gj_wave_steps(std::integer_sequence<int, Ks...>, Cx<T> (&a)[NC], LuState<T, NC>& st, int (&owners)[NC], int& mystep,
                                              int n, int lane) {
    (gj_wave_step<T, NC, Ks>(a, st, owners, mystep, n, lane), ...);
}
template <typename T, int NC>
__global__ void __launch_bounds__(64) k_det_inv_wave(SysDev<T> S, const T* __restrict__ MOUT, size_t mout_stride, size_t mout_off, int ch, int es,
                                                     T* __restrict__ MINV, size_t minv_stride, size_t minv_off,
                                                     T* __restrict__ DETS, size_t dets_stride, size_t dets_off, int P) {
    const int kdet = blockIdx.x, lane = threadIdx.x;
    const long w = blockIdx.y;
    const int n = S.det_n[ch];
    const T* Mw = MOUT + (size_t)w * mout_stride + mout_off + (size_t)kdet * n * n * 2 * P;
    Cx<T> a[NC];
#pragma unroll
    for (int m = 0; m < NC; ++m) {
        if (lane < n && m < n) a[m] = Cx<T>(Mw[(size_t)((lane * n + m) * 2) * es], Mw[(size_t)((lane * n + m) * 2 + 1) * es]);
        else a[m] = Cx<T>(T(0), T(0));
    }
    LuState<T, NC> st{n < 64 ? (~0ull << n) : 0ull, 0, Cx<T>(1, 0)};
    int owners[NC], mystep = 0;
#pragma unroll
    for (int m = 0; m < NC; ++m) owners[m] = 0;
    gj_wave_steps<T, NC>(std::make_integer_sequence<int, NC>(), a, st, owners, mystep, n, lane);
    if (lane < n) {
        T* Iw = MINV + (size_t)w * minv_stride + minv_off + ((size_t)kdet * n + mystep) * n * 2;
#pragma unroll
        for (int m = 0; m < NC; ++m)
            if (m < n) { Iw[2 * owners[m]] = a[m].re; Iw[2 * owners[m] + 1] = a[m].im; }
    }
    if (lane == 0) {
        T* dw = DETS + (size_t)w * dets_stride + dets_off + (size_t)kdet * 4;
        dw[0] = T(0.5) * ds_log(cx_abs2(st.ph)) + T(st.esum) * T(0.69314718055994530942);
        dw[1] = ds_atan2(st.ph.im, st.ph.re);
    }
}

// M = phi * q (values).  PHI [group][elec in spin][ocols][PV], grid (n_s, groups), block 256
template <typename T>
__global__ void __launch_bounds__(256) k_orbital_epilogue_val(SysDev<T> S, const T* __restrict__ PHI, size_t phi_group_stride,
                                                              const T* __restrict__ Q, T* __restrict__ MOUT, int sp,
                                                              size_t mout_stride, size_t mout_off, const T* __restrict__ bias,
                                                              const T* __restrict__ Sb) {
    const int ii = blockIdx.x, g = blockIdx.y, N = S.N, OC = S.ocols[sp];
    const int i0 = sp == 0 ? 0 : S.n_up, nparam = S.nparam[sp], i = i0 + ii;
    const int norb = S.norb[sp], n = S.det_n[S.mat_ch[sp]], row = S.row_off[sp] + ii;
    const T* Pw = PHI + (size_t)g * phi_group_stride + (size_t)ii * OC * PV;
    const T* Qw = Q + ((size_t)(g * N + i) * S.nparam_max) * 2 * PV;
    T* Mw = MOUT + (size_t)g * mout_stride + mout_off;
    for (int idx = threadIdx.x; idx < nparam * PV; idx += blockDim.x) {
        const int c = idx % PV, p = idx / PV;
        Cx<T> phi(Pw[(size_t)orb_col<T>(p, 0) * PV + c], Pw[(size_t)orb_col<T>(p, 1) * PV + c]);
        if (Sb) { phi.re += Sb[((size_t)g * OC + orb_col<T>(p, 0)) * PV + c]; phi.im += Sb[((size_t)g * OC + orb_col<T>(p, 1)) * PV + c]; }
        if (bias) { phi.re += bias[p]; phi.im += bias[nparam + p]; }
        const Cx<T> q(Qw[(size_t)(p * 2) * PV + c], Qw[(size_t)(p * 2 + 1) * PV + c]);
        const Cx<T> v = phi * q;
        T* mo = Mw + (((size_t)((p / norb) * n + row) * n + p % norb) * 2) * PV + c;
        mo[0] = v.re;
        mo[PV] = v.im;
    }
}

// MOUT (value chain) -> dense (B, K, n, n, 2) orbital matrices (network.py:601 'eval_mats')
template <typename T>
__global__ void k_gather_val(const T* __restrict__ MOUT, size_t mout_stride, size_t mout_off, size_t per, long w0, long nw,
                             T* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const long w = blockIdx.y;                       // walker within this chunk
    if (idx >= per || w >= nw) return;
    out[(size_t)(w0 + w) * per + idx] = MOUT[(size_t)(w / PV) * mout_stride + mout_off + idx * PV + w % PV];
}

}  // namespace ds
