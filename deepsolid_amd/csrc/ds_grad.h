// ds_grad.h -- reverse sweep of the value chain: parameter gradient of
//     L = sum_b [ cot_re[b] * log|psi_b| + cot_im[b] * arg psi_b ]
// the vector-Jacobian product behind the energy gradient of reference train.py:91-142
// (tangents_dot = mean(Re(clip_diff * conj(d log psi)))).
//
// Runs on the buffers of the value chain (ds_value.h: the contiguous axis carries PV walkers of a
// "group"), with every layer's activations kept.  Cotangent buffers mirror the forward ones:
//   HB   [group][electron][n][PV]      cotangent of the one-electron stream after a layer
//   ZBAR [group][electron][n][PV]      cotangent of the layer's pre-activation
//   SBAR [group][n][PV]                ... summed over electrons (cotangent of the shared spin-mean term)
//   GBAR [group][electron][rows][PV]   W * ZBAR: cotangent of the per-electron layer input rows [h | pair means]
//   H2BAR/Z2BAR [group*16 + c/5][k2][c%5][NP]   two-electron stream
// Weight gradients are contractions over (electron, walker) -- k_outer_gemm (MFMA) -- written as one
// partial per group and summed by k_reduce_partials in a fixed order (bit-reproducible, no atomics).
#pragma once
#include "ds_value.h"

namespace ds {

// WT[c][r] = W[r][c] for r < rows, 0 for rows <= r < ldt   (W: rows x cols, WT: cols x ldt)
template <typename T>
__global__ void k_transpose_pad(const T* __restrict__ W, int rows, int cols, T* __restrict__ WT, int ldt) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)cols * ldt) return;
    const int c = (int)(idx / ldt), r = (int)(idx % ldt);
    WT[idx] = r < rows ? W[(size_t)r * cols + c] : T(0);
}

// CW[group][k][re,im][PV] = conj(cot_w) * w_k,  w_k = D_k / sum_k' D_k'  (network.py:395-427); 0 for padding columns
template <typename T>
__global__ void __launch_bounds__(64) k_det_weights(SysDev<T> S, const T* __restrict__ DETS, size_t dets_stride, size_t dets_off1,
                                                    const T* __restrict__ cot, long Bc, T* __restrict__ CW) {
    const long w = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long ng = (Bc + PV - 1) / PV;
    if (w >= ng * PV) return;
    const int K = S.K;
    T* out = CW + (size_t)(w / PV) * K * 2 * PV + w % PV;
    if (w >= Bc) {
        for (int k = 0; k < 2 * K; ++k) out[(size_t)k * PV] = 0;
        return;
    }
    const T* Dw = DETS + (size_t)w * dets_stride;
    T la[DS_MAX_DETS], ar[DS_MAX_DETS];
    T mx = -1e300;
    for (int k = 0; k < K; ++k) {
        la[k] = Dw[4 * k]; ar[k] = Dw[4 * k + 1];
        if (S.n_detch > 1) { la[k] += Dw[dets_off1 + 4 * k]; ar[k] += Dw[dets_off1 + 4 * k + 1]; }
        mx = la[k] > mx ? la[k] : mx;
    }
    Cx<T> sum(0, 0);
    for (int k = 0; k < K; ++k) {
        T sn, cs;
        ds_sincos(ar[k], &sn, &cs);
        const T e = ds_exp(la[k] - mx);
        la[k] = e * cs; ar[k] = e * sn;
        sum = sum + Cx<T>(la[k], ar[k]);
    }
    const Cx<T> sinv = cx_inv(sum), cc(cot[2 * w], -cot[2 * w + 1]);
    for (int k = 0; k < K; ++k) {
        const Cx<T> v = cc * (Cx<T>(la[k], ar[k]) * sinv);
        out[(size_t)(2 * k) * PV] = v.re;
        out[(size_t)(2 * k + 1) * PV] = v.im;
    }
}

// Orbital matrices -> orbital head.  With A = conj(c) w_k (M_k^-1)[m][row]  (dL = Re sum A dM, M = phi * q):
//   PHIBAR[elec][packed col (p, Re/Im)][PV] = (Re, -Im)(A q)       cotangent of the real GEMM outputs
//   QBAR  [elec][p][re,im][PV]             = A phi                 (dL = Re(QBAR dq), for the envelope parameters)
// grid (n_s, groups), block 256
template <typename T>
__global__ void __launch_bounds__(256) k_orbital_bwd(SysDev<T> S, const T* __restrict__ PHI, size_t phi_group_stride,
                                                     const T* __restrict__ Q, const T* __restrict__ MINV, size_t minv_stride,
                                                     size_t minv_off, const T* __restrict__ CW, int sp, long Bc,
                                                     const T* __restrict__ bias, const T* __restrict__ Sb, T* __restrict__ PHIBAR,
                                                     T* __restrict__ QBAR) {
    const int ii = blockIdx.x, g = blockIdx.y, N = S.N, OC = S.ocols[sp];
    const int i0 = sp == 0 ? 0 : S.n_up, nparam = S.nparam[sp], i = i0 + ii;
    const int norb = S.norb[sp], n = S.det_n[S.mat_ch[sp]], row = S.row_off[sp] + ii;
    const T* Pw = PHI + (size_t)g * phi_group_stride + (size_t)ii * OC * PV;
    const T* Qw = Q + ((size_t)(g * N + i) * S.nparam_max) * 2 * PV;
    const T* Iw = MINV + (size_t)g * minv_stride + minv_off;
    const T* Cw = CW + (size_t)g * S.K * 2 * PV;
    T* Pb = PHIBAR + (size_t)g * phi_group_stride + (size_t)ii * OC * PV;
    T* Qb = QBAR + ((size_t)(g * N + i) * S.nparam_max) * 2 * PV;
    for (int idx = threadIdx.x; idx < nparam * PV; idx += blockDim.x) {
        const int c = idx % PV, p = idx / PV;
        const int cr = orb_col<T>(p, 0), ci = orb_col<T>(p, 1);
        Cx<T> a(0, 0), qb(0, 0);
        if ((long)g * PV + c < Bc) {
            const int kdet = p / norb, m = p % norb;
            const T* mi = Iw + ((size_t)kdet * n * n + (size_t)m * n + row) * 2 * PV + c;
            const Cx<T> A = Cx<T>(Cw[(size_t)(2 * kdet) * PV + c], Cw[(size_t)(2 * kdet + 1) * PV + c]) * Cx<T>(mi[0], mi[PV]);
            Cx<T> phi(Pw[(size_t)cr * PV + c], Pw[(size_t)ci * PV + c]);
            if (Sb) { phi.re += Sb[((size_t)g * OC + cr) * PV + c]; phi.im += Sb[((size_t)g * OC + ci) * PV + c]; }   // use_last_layer
            if (bias) { phi.re += bias[p]; phi.im += bias[nparam + p]; }
            const Cx<T> q(Qw[(size_t)(p * 2) * PV + c], Qw[(size_t)(p * 2 + 1) * PV + c]);
            a = A * q;
            qb = A * phi;
        }
        Pb[(size_t)cr * PV + c] = a.re;
        Pb[(size_t)ci * PV + c] = -a.im;
        Qb[(size_t)(p * 2) * PV + c] = qb.re;
        Qb[(size_t)(p * 2 + 1) * PV + c] = qb.im;
    }
}

// Envelope parameters (network.py:335-364): q = e * exp(i k.x), e = sum_a pi[a,p] exp(-r_a),
//   isotropic r = |sd sigma[a,p]|;  diagonal r = |(sigma[a,m,p] rel_m)_m|;  full r = |(sum_k sigma[k,m,a,p] rel_k)_m|
//   d pi[a,p] = sum_{i,b} ge exp(-r),   d sigma = sum_{i,b} ge pi exp(-r) (-dr/dsigma),   ge = Re(QBAR * exp(i k.x)).
// grid (groups, spin channels), block 256: a thread owns ONE sigma entry (a, p, j), j < 1 / 3 / 9 (and pi[a,p] when j = 0)
template <typename T>
__global__ void __launch_bounds__(256) k_env_grad(SysDev<T> S, const T* __restrict__ x, long Bc, const T* __restrict__ G0,
                                                  const T* __restrict__ QBAR, const T* __restrict__ env_pi0,
                                                  const T* __restrict__ env_sg0, const T* __restrict__ env_pi1,
                                                  const T* __restrict__ env_sg1, T* __restrict__ part, size_t part_stride,
                                                  long off_pi0, long off_sg0, long off_pi1, long off_sg1) {
    const int g = blockIdx.x, sp = blockIdx.y, N = S.N, A = S.A;
    const int i0 = sp == 0 ? 0 : S.n_up, ns = sp == 0 ? S.n_up : S.n_dn, np = S.nparam[sp];
    const int nsig = S.env_type == 0 ? 1 : (S.env_type == 1 ? 3 : 9);
    const T* pi_ = sp == 0 ? env_pi0 : env_pi1;
    const T* sg_ = sp == 0 ? env_sg0 : env_sg1;
    T* dpi = part + (size_t)g * part_stride + (sp == 0 ? off_pi0 : off_pi1);
    T* dsg = part + (size_t)g * part_stride + (sp == 0 ? off_sg0 : off_sg1);
    for (int idx = threadIdx.x; idx < A * np * nsig; idx += blockDim.x) {
        const int j = idx / (A * np), ap = idx % (A * np), a = ap / np, p = ap % np;
        const int jk = j / 3, jm = j % 3;            // full: sigma[k = jk][m = jm];  diagonal: m = j
        const T pv = pi_[ap];
        // sigma entries feeding the three components u_m of this (a, p)
        T s00 = 0, s01 = 0, s02 = 0, s10 = 0, s11 = 0, s12 = 0, s20 = 0, s21 = 0, s22 = 0;
        if (S.env_type == 0) s00 = sg_[ap];
        else if (S.env_type == 1) { s00 = sg_[(a * 3 + 0) * np + p]; s11 = sg_[(a * 3 + 1) * np + p]; s22 = sg_[(a * 3 + 2) * np + p]; }
        else {
            s00 = sg_[(0 * A + a) * np + p]; s01 = sg_[(1 * A + a) * np + p]; s02 = sg_[(2 * A + a) * np + p];
            s10 = sg_[(3 * A + a) * np + p]; s11 = sg_[(4 * A + a) * np + p]; s12 = sg_[(5 * A + a) * np + p];
            s20 = sg_[(6 * A + a) * np + p]; s21 = sg_[(7 * A + a) * np + p]; s22 = sg_[(8 * A + a) * np + p];
        }
        const T* kv = S.klist[sp] + 3 * (p % S.norb[sp]);
        T gp = 0, gs = 0;
        for (int ii = 0; ii < ns; ++ii) {
            const int i = i0 + ii;
            const T* fp = G0 + ((size_t)(g * N + i) * S.ldk + S.nf * a) * PV;     // rows sd, rel_x, rel_y, rel_z of atom a
            const T* qb = QBAR + ((size_t)(g * N + i) * S.nparam_max + p) * 2 * PV;
            for (int c = 0; c < PV; ++c) {
                const long wi = (long)g * PV + c;
                if (wi >= Bc) break;
                const T* xp = x + (size_t)wi * 3 * N + 3 * i;
                T sn, cs;
                ds_sincos(kv[0] * xp[0] + kv[1] * xp[1] + kv[2] * xp[2], &sn, &cs);
                const T ge = qb[c] * cs - qb[PV + c] * sn;
                if (S.env_type == 0) {
                    const T sd = fp[c], u = sd * s00, ex = ds_exp(-ds_abs(u));
                    gp += ge * ex;
                    gs -= ge * pv * ex * sd * ds_sign(u);
                } else {
                    const T r0 = fp[(size_t)PV + c], r1 = fp[(size_t)2 * PV + c], r2 = fp[(size_t)3 * PV + c];
                    // u_m = sum_k sigma[k][m] rel_k  (diagonal: only k = m)
                    const T u0 = s00 * r0 + s10 * r1 + s20 * r2, u1 = s01 * r0 + s11 * r1 + s21 * r2, u2 = s02 * r0 + s12 * r1 + s22 * r2;
                    const T r = ds_sqrt(u0 * u0 + u1 * u1 + u2 * u2), ex = ds_exp(-r);
                    gp += ge * ex;
                    if (r > T(0)) {
                        const T um = jm == 0 ? u0 : (jm == 1 ? u1 : u2);
                        const int kk = S.env_type == 1 ? jm : jk;
                        const T rk = kk == 0 ? r0 : (kk == 1 ? r1 : r2);
                        gs -= ge * pv * ex * um * rk / r;
                    }
                }
            }
        }
        if (j == 0) dpi[ap] = gp;
        if (S.env_type == 0) dsg[ap] = gs;
        else if (S.env_type == 1) dsg[(a * 3 + jm) * np + p] = gs;
        else dsg[(j * A + a) * np + p] = gs;
    }
}

// Weight gradient dW[k][n] = sum_t sum_j X[t][k][j] * Z[t][n][j]  (t: electron / pair-block tiles, j: the contiguous
// walker or pair axis).  Both MFMA operands run along j, so a lane loads FOUR consecutive j (32 B) of its row
// and feeds them to four k-steps: any assignment of j to (step, lane group) is a valid contraction order as
// long as X and Z use the same one.  A wave owns a 32 x 32 block of dW; grid (blocks of 4 waves, groups).
template <typename T>
__global__ void __launch_bounds__(256) k_outer_gemm(const T* __restrict__ X, size_t x_group_stride, size_t x_tile_stride, int ldx,
                                                    const T* __restrict__ Z, size_t z_group_stride, size_t z_tile_stride, int ldz,
                                                    int n_tiles, int J, int K, int Nc, T* __restrict__ part, size_t part_stride,
                                                    int nsplit) {
    typedef typename Acc4<T>::type acc_t;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    const int g = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    // nsplit > 1: the tiles are dealt to nsplit waves per 32 x 32 block, each writing its own partial
    // (part index g * nsplit + split) -- for small matrices with a long contraction (pair-stream weights)
    const int nbn = (Nc + 31) / 32, nwt = ((K + 31) / 32) * nbn, wall = blockIdx.x * 4 + wave;
    const int wt = wall % nwt, split = wall / nwt;
    if (split >= nsplit) return;
    const int k0 = (wt / nbn) * 32, n0 = (wt % nbn) * 32;
    const int tps = (n_tiles + nsplit - 1) / nsplit, t_lo = split * tps, t_hi = t_lo + tps < n_tiles ? t_lo + tps : n_tiles;
    const T* Xg = X + (size_t)g * x_group_stride;
    const T* Zg = Z + (size_t)g * z_group_stride;
    int kr[2], nr[2];
    bool kv[2], nv[2];
    for (int a = 0; a < 2; ++a) {
        kr[a] = k0 + 16 * a + lr; kv[a] = kr[a] < K; if (!kv[a]) kr[a] = 0;
        nr[a] = n0 + 16 * a + lr; nv[a] = nr[a] < Nc; if (!nv[a]) nr[a] = 0;
    }
    acc_t acc[2][2];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) acc[a][b] = acc_t{0, 0, 0, 0};
    const vec4 zero = {0, 0, 0, 0};
    for (int t = t_lo; t < t_hi; ++t) {
        const T* xa[2]; const T* zb[2];
        for (int a = 0; a < 2; ++a) {
            xa[a] = Xg + (size_t)t * x_tile_stride + (size_t)kr[a] * ldx + 4 * lq;
            zb[a] = Zg + (size_t)t * z_tile_stride + (size_t)nr[a] * ldz + 4 * lq;
        }
        for (int j0 = 0; j0 < J; j0 += 16) {
            vec4 av[2], bv[2];
            for (int a = 0; a < 2; ++a) {
                av[a] = kv[a] ? *reinterpret_cast<const vec4*>(xa[a] + j0) : zero;
                bv[a] = nv[a] ? *reinterpret_cast<const vec4*>(zb[a] + j0) : zero;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = mfma16(av[a][s], bv[b][s], acc[a][b]);
        }
    }
    T* out = part + ((size_t)g * nsplit + split) * part_stride;
    for (int a = 0; a < 2; ++a)
        for (int r = 0; r < 4; ++r) {
            const int k = k0 + 16 * a + acc_row<T>(lane, r);
            if (k >= K) continue;
            for (int b = 0; b < 2; ++b) {
                const int n = n0 + 16 * b + lr;
                if (n < Nc) out[(size_t)k * Nc + n] = acc[a][b][r];
            }
        }
}

// out[n] = sum_t sum_j Z[t][n][j]   (bias gradients).  grid (rows, groups), block 256
template <typename T>
__global__ void __launch_bounds__(256) k_row_sums(const T* __restrict__ Z, size_t z_group_stride, size_t z_tile_stride, int ldz,
                                                  int n_tiles, int J, T* __restrict__ part, size_t part_stride) {
    __shared__ T red[256];
    const int n = blockIdx.x, g = blockIdx.y;
    const T* Zg = Z + (size_t)g * z_group_stride + (size_t)n * ldz;
    T v = 0;
    for (int t = 0; t < n_tiles; ++t)
        for (int j = threadIdx.x; j < J; j += blockDim.x) v += Zg[(size_t)t * z_tile_stride + j];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)g * part_stride + n] = red[0];
}

// orbital bias gradient (network.py:546 bias_orbitals): d b[part * nparam + p] = sum over electrons and walkers of the
// packed column (p, part) of PHIBAR.  grid (2 * nparam, groups), block 256
template <typename T>
__global__ void __launch_bounds__(256) k_orb_bias_grad(const T* __restrict__ PHIBAR, size_t group_stride, int ns, int OC, int nparam,
                                                       T* __restrict__ part, size_t part_stride) {
    __shared__ T red[256];
    const int q = blockIdx.x, g = blockIdx.y, p = q % nparam, pt = q / nparam;
    const T* Pg = PHIBAR + (size_t)g * group_stride + (size_t)orb_col<T>(p, pt) * PV;
    T v = 0;
    for (int idx = threadIdx.x; idx < ns * PV; idx += blockDim.x) v += Pg[(size_t)(idx / PV) * OC * PV + idx % PV];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)g * part_stride + q] = red[0];
}

// out[group][col][PV] = sum over the tiles (electrons) of in[group][tile][col][PV]: cotangent of a shared term
template <typename T>
__global__ void k_sum_tiles(const T* __restrict__ in, size_t group_stride, int n_tiles, int n, T* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
    if (idx >= n) return;
    T v = 0;
    for (int t = 0; t < n_tiles; ++t) v += in[(size_t)g * group_stride + (size_t)t * n + idx];
    out[(size_t)g * n + idx] = v;
}

// MEAN[group][spin][k][PV] = mean over the spin's electrons of G rows (network.py:327-330)
template <typename T>
__global__ void __launch_bounds__(256) k_spin_mean(SysDev<T> S, const T* __restrict__ G, int Kh, T* __restrict__ MEAN) {
    const int g = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S.nch * Kh * PV) return;
    const int c = idx % PV, k = (idx / PV) % Kh, s = idx / (PV * Kh);
    const int i0 = s == 0 ? 0 : S.n_up, ns = s == 0 ? S.n_up : S.n_dn;
    const T* Gw = G + (size_t)g * S.N * S.ldk * PV;
    T v = 0;
    for (int i = i0; i < i0 + ns; ++i) v += Gw[((size_t)i * S.ldk + k) * PV + c];
    MEAN[(size_t)g * S.nch * Kh * PV + idx] = v / T(ns);
}

// One-electron layer, element-wise part of the reverse sweep (network.py:517-528):
//   hbar = D1[e][n] (W * ZBAR of the layer above, or of the orbital head) + MB[spin(e)][n] / n_spin (its spin-mean rows;
//          MB2: second contribution, the other spin's orbital head with use_last_layer)
//          + CARRY[e][n] / sqrt2 (residual bypass of the layer above)
//   HB = hbar;   y = tanh(z) recovered from the stored activations;   ZBAR = (RES ? hbar / sqrt2 : hbar) * (1 - y^2)
//   SBAR[n] = sum_e ZBAR[e][n];   bias partial[n] = sum_{e,c} ZBAR
// grid (Nout / 4, groups), block 4 * PV: thread (n, c) walks the electrons.
template <typename T, bool RES>
__global__ void __launch_bounds__(4 * PV) k_layer_bwd_prep(SysDev<T> S, const T* __restrict__ D1, int ld1, const T* __restrict__ MB,
                                                           const T* __restrict__ MB2,
                                                           const T* __restrict__ CARRY, const T* __restrict__ Gout,
                                                           const T* __restrict__ Gin, int Nout, T* __restrict__ HB,
                                                           T* __restrict__ ZBAR, T* __restrict__ SBAR, T* __restrict__ part,
                                                           size_t part_stride) {
    __shared__ T red[4][PV];
    const int g = blockIdx.y, nl = threadIdx.x / PV, c = threadIdx.x % PV, n = blockIdx.x * 4 + nl, N = S.N;
    const T rs2 = T(0.70710678118654752440), s2 = T(1.41421356237309504880);
    T sb = 0;
    for (int e = 0; e < N; ++e) {
        const int sp = spin_of(e, S.n_up);
        T hb = D1[((size_t)(g * N + e) * ld1 + n) * PV + c];
        if (MB) hb += MB[((size_t)(g * S.nch + sp) * Nout + n) * PV + c] / T(sp == 0 ? S.n_up : S.n_dn);
        if (MB2) hb += MB2[((size_t)(g * S.nch + sp) * Nout + n) * PV + c] / T(sp == 0 ? S.n_up : S.n_dn);
        if (CARRY) hb += CARRY[((size_t)(g * N + e) * Nout + n) * PV + c] * rs2;
        HB[((size_t)(g * N + e) * Nout + n) * PV + c] = hb;
        const T ho = Gout[((size_t)(g * N + e) * S.ldk + n) * PV + c];
        T y = ho;
        if (RES) y = s2 * ho - Gin[((size_t)(g * N + e) * S.ldk + n) * PV + c];
        const T zb = (RES ? hb * rs2 : hb) * (1 - y * y);
        ZBAR[((size_t)(g * N + e) * Nout + n) * PV + c] = zb;
        sb += zb;
    }
    SBAR[((size_t)g * Nout + n) * PV + c] = sb;
    red[nl][c] = sb;
    __syncthreads();
    if (c == 0) {
        T v = 0;
        for (int j = 0; j < PV; ++j) v += red[nl][j];
        part[(size_t)g * part_stride + n] = v;
    }
}

// pair-mean rows of GBAR -> cotangent of the pair stream (network.py:331: mean over the partners of each spin):
//   pm[k][c][q = e*N + j] = GBAR[e][row0 + spin(j)*K2 + k][walker c] / n_spin(j)
template <typename T>
__device__ __forceinline__ T pair_mean_bar(const SysDev<T>& S, const T* __restrict__ GB, int ldg, int row0, int K2, int g, int col,
                                           int k, int q) {
    const int N = S.N;
    if (q >= N * N) return T(0);
    const int e = q / N, j = q % N, sp = spin_of(j, S.n_up);
    return GB[((size_t)(g * N + e) * ldg + row0 + sp * K2 + k) * PV + col] / T(sp == 0 ? S.n_up : S.n_dn);
}

// top of the pair stream: H2BAR = pair-mean scatter only.  grid (ceil(NP/256), K2 * 5, groups * 16)
template <typename T>
__global__ void __launch_bounds__(256) k_pair_scatter(SysDev<T> S, const T* __restrict__ GB, int ldg, int row0, int K2,
                                                      T* __restrict__ H2BAR) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x, kc = blockIdx.y, w = blockIdx.z;
    if (q >= S.NP) return;
    const int k = kc / 5, c = kc % 5, g = w / (PV / 5), col = (w % (PV / 5)) * 5 + c;
    H2BAR[((size_t)w * K2 * 5 + kc) * S.NP + q] = pair_mean_bar(S, GB, ldg, row0, K2, g, col, k, q);
}

// Two-electron layer  h2' = res(h2, tanh(h2 W + b))  (network.py:525-528), reverse sweep.  Same tiling as the
// forward k_two_layer: a wave owns 16 pairs x 5 walkers of one 5-walker block.
//   z2bar[n] = (RES ? hb[n] / sqrt2 : hb[n]) * (1 - y[n]^2)         written to Z2BAR (for dW2 / db2)
//   DX: H2BAR_in[k] = sum_n W[k][n] z2bar[n]  (MFMA, B operand formed on the fly)  + RES hb[k] / sqrt2 + pair-mean scatter
template <typename T, int NTI, int NTO, bool RES, bool DX>
__global__ void __launch_bounds__(256, 3) k_two_bwd(SysDev<T> S, const T* __restrict__ HBn, const T* __restrict__ Hout,
                                                 const T* __restrict__ Hin, const T* __restrict__ W, const T* __restrict__ GB, int ldg,
                                                 int row0, T* __restrict__ Z2BAR, T* __restrict__ HBi, int kin_rows) {
    typedef typename Acc4<T>::type acc_t;
    constexpr int Kin = 16 * NTI, Kout = 16 * NTO;
    const int w = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + wave, NP = S.NP;
    if (pt * 16 >= NP) return;
    const int lr = lane & 15, lq = lane >> 4, q = pt * 16 + lr;
    const T rs2 = T(0.70710678118654752440), s2 = T(1.41421356237309504880);
    // kin_rows: rows of Hin in memory (= Kin, except for the first pair layer with a residual -- RES without DX --, whose input is the
    // nf-row feature block: rows beyond it have no residual, and their cotangent is zero anyway: zero-padded weights downstream)
    const size_t ob = (size_t)w * Kout * 5 * NP + q, ib = (size_t)w * kin_rows * 5 * NP + q;
    acc_t acc[NTI][5];
    for (int a = 0; a < NTI; ++a)
        for (int c = 0; c < 5; ++c) acc[a][c] = acc_t{0, 0, 0, 0};
    for (int ks = 0; ks < Kout / 4; ++ks) {
        const int n = 4 * ks + lq;
        T bv[5];
        for (int c = 0; c < 5; ++c) {
            const size_t o = ob + (size_t)(n * 5 + c) * NP;
            const T hb = HBn[o], ho = Hout[o];
            T y = ho;
            if (RES) y = s2 * ho - ((DX || n < kin_rows) ? Hin[ib + (size_t)(n * 5 + c) * NP] : T(0));
            bv[c] = (RES ? hb * rs2 : hb) * (1 - y * y);
            Z2BAR[o] = bv[c];
        }
        if (DX) {
            T av[NTI];
            for (int a = 0; a < NTI; ++a) av[a] = W[(size_t)(16 * a + lr) * Kout + n];
            for (int a = 0; a < NTI; ++a)
                for (int c = 0; c < 5; ++c) acc[a][c] = mfma16(av[a], bv[c], acc[a][c]);
        }
    }
    if (!DX) return;
    const int g = w / (PV / 5), cb = (w % (PV / 5)) * 5;
    for (int a = 0; a < NTI; ++a)
        for (int r = 0; r < 4; ++r) {
            const int k = 16 * a + acc_row<T>(lane, r);
            for (int c = 0; c < 5; ++c) {
                T v = acc[a][c][r] + pair_mean_bar(S, GB, ldg, row0, Kin, g, cb + c, k, q);
                if (RES) v += HBn[ob + (size_t)(k * 5 + c) * NP] * rs2;
                HBi[ib + (size_t)(k * 5 + c) * NP] = v;
            }
        }
}

// out[g][p] = sum_z scratch[g * nz + z][p]   (second stage of a split k_outer_gemm).  grid (ceil(n / 256), groups)
template <typename T>
__global__ void k_reduce_splits(const T* __restrict__ scratch, int nz, int n, T* __restrict__ part, size_t part_stride) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
    if (p >= n) return;
    T v = 0;
    for (int z = 0; z < nz; ++z) v += scratch[((size_t)g * nz + z) * n + p];
    part[(size_t)g * part_stride + p] = v;
}

// grad[p] = (accumulate ? grad[p] : 0) + sum_g part[g][p], groups in index order
template <typename T>
__global__ void k_reduce_partials(const T* __restrict__ part, size_t part_stride, long ng, long n, int accumulate, T* __restrict__ grad) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    T v = accumulate ? grad[p] : T(0);
    for (long g = 0; g < ng; ++g) v += part[(size_t)g * part_stride + p];
    grad[p] = v;
}

}  // namespace ds
