// ds_kernels.h -- gfx950 kernels of the forward-Laplacian local-energy chain.
//
// Data layout (all in HBM, element type T, one "walker" = one MCMC configuration):
//   jet tensors carry, for every scalar, P = roundup16(3N+2) "slots" contiguous in memory:
//       slot 0 value | slot 1 Laplacian | slot 2+3j+c = d/dx_{j,c} | zero padding
//   G    [walker][electron i][row k][P]   rows 0..h1-1 : one-electron stream h_i
//                                         rows h1..    : mean_j h2_ji over spin-up j, then spin-down j
//   MEAN [walker][spin][k][P]             spin means of h (the shared part of the layer input)
//   H2   [walker][k2][5][NP]              two-electron stream, 5 = (value, d/dr_x, d/dr_y, d/dr_z, Laplacian)
//                                         h2[j][e] as a function of r = x_j - x_e, stored at pair = e*N + j;
//                                         NP = roundup16(N*N)
//   MOUT [walker][channel][det k][slot tile][elec i][orb m][re/im][16]   orbital matrices with all slots, slot-tile
//                                         major so that the determinant kernels stream one tile contiguously
//
// Every dense contraction is computed TRANSPOSED, C[n][slot] = sum_k W[k][n] * X[k][slot], so that
// the MFMA A operand is a row of the weight matrix (n contiguous), the B operand is a row of the
// jet tensor (slots contiguous) and the accumulator writes rows of the output jet tensor: every
// global access is a 128-byte segment per 16 lanes and the layout is the same at every layer.
#pragma once
#include "ds_device.h"

namespace ds {

#define DS_MAXL 8

template <typename T> struct SysDev {
    int N, n_up, n_dn, A, L, K, nch;
    int D, P, NP;                 // 3N+2, padded slots, padded pairs
    int n_layers, n_double;
    int h1[DS_MAXL + 1];          // h1[0] = 4A, h1[l+1] = hidden_single[l]
    int h2[DS_MAXL + 1];          // h2[0] = 4,  h2[l+1] = hidden_double[l]
    int ldk;                      // rows per electron in G
    int full_det, env_type, bias_orb;   // network options (network.py:609-621)
    int dist_type, nf;            // 0 'nu' / 1 'tri'; features per atom or pair: 4 / 7 (rows padded to h1[0], h2[0])
    int norb[2];                  // orbitals per determinant seen by spin s: n_s, or N with full_det
    int n_detch;                  // determinant channels: one per active spin, or 1 with full_det
    int det_n[2];                 // matrix size of channel ch
    int mat_ch[2], row_off[2];    // channel and first row that spin s's electrons fill
    int nparam[2];                // norb * K
    int nparam_max;
    int ocols[2];                 // packed orbital columns (2*nparam rounded up to 64)
    const T *prim_a, *prim_ainv, *sim_a, *sim_ainv, *prim_AV, *prim_BV, *sim_AV, *sim_BV, *atoms;
    const T* klist[2];            // per spin: (norb[s], 3); with full_det both point to the concatenated list
    // Ewald
    int As, NG, dist_mode;
    const T *sim_atoms, *sim_charges, *disp27, *shift27, *gpoints, *gweight, *ion_re, *ion_im;
    T alpha, ee_const, ei_const, ii_total;
    // G = 2 pi (n1, n2, n3) . recvec: gidx (NG,3) = the three table positions of a G point (n_j - nmin_j + g_off[j], stored as T),
    // g_len = table entries per electron (0: no tables, direct sincos per (G, electron))
    const T* gidx;
    int g_nmin[3], g_off[3], g_len;
};
// dynamic LDS of k_ewald: walker coordinates + two reduction arrays + the per-electron phase tables
template <typename T> inline size_t ewald_lds_bytes(const SysDev<T>& S) {
    return (size_t)(3 * S.N + 512 + 2 * (size_t)S.N * S.g_len) * sizeof(T);
}

__device__ __forceinline__ int spin_of(int i, int n_up) { return i < n_up ? 0 : 1; }

// =====================================================================================
// 1. input features (reference network.py:249-302) + envelope*phase jets (network.py:335-337,449-458)
// =====================================================================================
template <typename T>
__global__ void __launch_bounds__(256) k_features(SysDev<T> S, const T* __restrict__ x, const T* __restrict__ env_pi0,
                                                  const T* __restrict__ env_sg0, const T* __restrict__ env_pi1,
                                                  const T* __restrict__ env_sg1, T* __restrict__ G, int ldg, T* __restrict__ MEAN,
                                                  T* __restrict__ H2, T* __restrict__ Q) {
    // ldg = rows per electron tile of G (S.ldk, or the layer-0 input width when the tiles go to their own buffer)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* xs = reinterpret_cast<T*>(smem_raw);        // [N][3] raw
    T* px = xs + 3 * S.N;                          // [N][3] wrapped into the primitive cell
    T* sx = px + 3 * S.N;                          // [N][3] wrapped into the simulation cell
    Jet5<T>* jea = reinterpret_cast<Jet5<T>*>(sx + 3 * S.N);   // [N*A][nf]
    const int w = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int N = S.N, A = S.A, P = S.P, NP = S.NP;
    const T* xw = x + (size_t)w * 3 * N;
    for (int i = tid; i < N; i += nt) {
        T r[3] = {xw[3 * i], xw[3 * i + 1], xw[3 * i + 2]}, o[3], wr[3];
        for (int c = 0; c < 3; ++c) xs[3 * i + c] = r[c];
        wrap_point(r, S.prim_a, S.prim_ainv, o, wr);
        for (int c = 0; c < 3; ++c) px[3 * i + c] = o[c];
        wrap_point(r, S.sim_a, S.sim_ainv, o, wr);
        for (int c = 0; c < 3; ++c) sx[3 * i + c] = o[c];
    }
    __syncthreads();
    // electron-atom jets
    const int nf = S.nf;
    for (int ia = tid; ia < N * A; ia += nt) {
        const int i = ia / A, a = ia % A;
        T r[3];
        for (int c = 0; c < 3; ++c) r[c] = px[3 * i + c] - S.atoms[3 * a + c];
        Jet5<T> o[7];
        distance_jet(S.dist_type, r, S.prim_AV, S.prim_BV, S.L, o);
        for (int f = 0; f < nf; ++f) jea[nf * ia + f] = o[f];
    }
    __syncthreads();
    const int K1 = S.h1[0];      // nf * A rounded up to a multiple of 4 (zero rows pad)
    // one-electron stream rows of G: row k = nf*a + f = [sd, rel...] per atom (network.py:503-504)
    T* Gw = G + (size_t)w * N * ldg * P;
    for (int idx = tid; idx < N * K1 * P; idx += nt) {
        const int slot = idx % P, k = (idx / P) % K1, i = idx / (P * K1);
        T v = 0;
        if (k < nf * A) {
            const Jet5<T>& j = jea[nf * (i * A + k / nf) + (k % nf)];
            if (slot == 0) v = j.v;
            else if (slot == 1) v = j.l;
            else if (slot < S.D && (slot - 2) / 3 == i) v = j.g[(slot - 2) % 3];
        }
        Gw[((size_t)i * ldg + k) * P + slot] = v;
    }
    // spin means of the one-electron stream
    T* Mw = MEAN + (size_t)w * S.nch * K1 * P;
    for (int idx = tid; idx < S.nch * K1 * P; idx += nt) {
        const int slot = idx % P, k = (idx / P) % K1, s = idx / (P * K1);
        const int i0 = s == 0 ? 0 : S.n_up, ns = s == 0 ? S.n_up : S.n_dn;
        const int a = k / nf, f = k % nf;
        T v = 0;
        if (k < nf * A) {
            if (slot < 2) {
                for (int i = i0; i < i0 + ns; ++i) v += (slot == 0 ? jea[nf * (i * A + a) + f].v : jea[nf * (i * A + a) + f].l);
            } else if (slot < S.D) {
                const int j = (slot - 2) / 3;
                if (j >= i0 && j < i0 + ns) v = jea[nf * (j * A + a) + f].g[(slot - 2) % 3];
            }
        }
        Mw[idx] = v / T(ns);
    }
    // two-electron stream (pair features of r = x_i - x_j in the simulation cell; diagonal masked,
    // network.py:294-300); rows nf..h2[0]-1 are zero padding
    T* Hw = H2 + (size_t)w * S.h2[0] * 5 * NP;
    // stored pair index q = e*N + j holds h2[j][e] (first electron j, second electron e), r = x_j - x_e:
    // network.py:323-328 averages over the FIRST index, so electron e's partners are contiguous.
    for (int pr = tid; pr < NP; pr += nt) {
        const int e = pr / N, j = pr % N;
        Jet5<T> o[8];
        for (int f = 0; f < 8; ++f) o[f] = jet_zero<T>();
        if (pr < N * N && e != j) {
            T r[3];
            for (int c = 0; c < 3; ++c) r[c] = sx[3 * j + c] - sx[3 * e + c];
            distance_jet(S.dist_type, r, S.sim_AV, S.sim_BV, S.L, o);
        }
        for (int f = 0; f < S.h2[0]; ++f) {
            Hw[(size_t)(f * 5 + 0) * NP + pr] = o[f].v;
            Hw[(size_t)(f * 5 + 1) * NP + pr] = o[f].g[0];
            Hw[(size_t)(f * 5 + 2) * NP + pr] = o[f].g[1];
            Hw[(size_t)(f * 5 + 3) * NP + pr] = o[f].g[2];
            Hw[(size_t)(f * 5 + 4) * NP + pr] = 2 * o[f].l;    // Laplacian over x_j AND x_e
        }
    }
    // q_i[p] = envelope_i[p] * exp(i k_m . x_i), p = det*n_s + m, as a complex 5-jet in x_i
    T* Qw = Q + (size_t)w * N * S.nparam_max * 10;
    for (int idx = tid; idx < N * S.nparam_max; idx += nt) {
        const int i = idx / S.nparam_max, p = idx % S.nparam_max;
        const int s = spin_of(i, S.n_up);
        if (p >= S.nparam[s]) continue;
        const T* pi_ = s == 0 ? env_pi0 : env_pi1;
        const T* sg_ = s == 0 ? env_sg0 : env_sg1;
        const int np = S.nparam[s];
        Jet5<T> e = jet_zero<T>();
        for (int a = 0; a < A; ++a) {
            const T pw = pi_[a * np + p];
            if (S.env_type == 0) {            // isotropic (network.py:335-337): exp(-|sd sigma|)
                const Jet5<T>& sd = jea[nf * (i * A + a)];
                const T sg = sg_[a * np + p];
                const T u = sd.v * sg;
                const T ex = pw * ds_exp(-ds_abs(u));
                e = jet_add(e, jet_fn(sd, ex, -ds_sign(u) * sg * ex, sg * sg * ex));
            } else {                           // diagonal (:340-343) / full (:346-364): exp(-|Sigma rel|)
                Jet5<T> r2 = jet_zero<T>();
                for (int m = 0; m < 3; ++m) {
                    Jet5<T> u = jet_zero<T>();
                    if (S.env_type == 1) u = jet_scale(sg_[(a * 3 + m) * np + p], jea[nf * (i * A + a) + 1 + m]);
                    else
                        for (int k = 0; k < 3; ++k) u = jet_add(u, jet_scale(sg_[((k * 3 + m) * A + a) * np + p], jea[nf * (i * A + a) + 1 + k]));
                    r2 = jet_add(r2, jet_mul(u, u));
                }
                const T r = ds_sqrt(r2.v);
                const Jet5<T> rj = jet_fn(r2, r, T(0.5) / r, T(-0.25) / (r * r2.v));
                const T ex = pw * ds_exp(-r);
                e = jet_add(e, jet_fn(rj, ex, -ex, ex));
            }
        }
        const int m = p % S.norb[s];
        const T* kv = S.klist[s] + 3 * m;
        const T kx = kv[0] * xs[3 * i] + kv[1] * xs[3 * i + 1] + kv[2] * xs[3 * i + 2];
        T sn, cs;
        ds_sincos(kx, &sn, &cs);
        const T k2 = kv[0] * kv[0] + kv[1] * kv[1] + kv[2] * kv[2];
        T* q = Qw + (size_t)idx * 10;
        q[0] = e.v * cs; q[1] = e.v * sn;
        T lre = e.l * cs - e.v * k2 * cs, lim = e.l * sn - e.v * k2 * sn;
        for (int c = 0; c < 3; ++c) {
            // ph_g = i k ph = (-k sn, k cs)
            q[2 + 2 * c] = e.g[c] * cs - e.v * kv[c] * sn;
            q[3 + 2 * c] = e.g[c] * sn + e.v * kv[c] * cs;
            lre += 2 * e.g[c] * (-kv[c] * sn);
            lim += 2 * e.g[c] * (kv[c] * cs);
        }
        q[8] = lre; q[9] = lim;
    }
}

// =====================================================================================
// 2a. spin means of the two-electron stream -> rows [h1, h1 + nch*h2) of G   (network.py:305-332)
//     expands the 5-slot pair jets to dense slots:  d/dx_i = +d/dr, d/dx_j = -d/dr
// =====================================================================================
// grid.z splits the K2 pair features (the pair jets of a split then fit several workgroups' worth of LDS per CU)
// skip (round 6): the rows are exactly zero outside slot tile 0, the electron's own tile(s) and the tiles of the partner spin's slots.
//   0: every slot of every row is written;  1: the structurally zero tiles are NOT written -- only for consumers that never use them
//   (k_layer1_lr and the float64 k_jet_gemm<.., 2> instances: all pair-mean k-steps run under the tile masks).  What is left
//   unwritten holds stale numbers of an earlier layer: loaded by the operand ring, never multiplied.
template <typename T>
__global__ void __launch_bounds__(256) k_m2_expand(SysDev<T> S, const T* __restrict__ H2, int K2, T* __restrict__ G, int row0, int ldg, int skip) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int e = blockIdx.x, w = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    const int N = S.N, P = S.P, NP = S.NP;
    const int Kc = K2 / gridDim.z, k0 = blockIdx.z * Kc;      // this workgroup's features
    T* hs = reinterpret_cast<T*>(smem_raw);     // [Kc*5][N]  pair jets h2[j][e], j = 0..N-1
    T* sums = hs + Kc * 5 * N;                  // [nch][Kc][5]
    const T* Hw = H2 + ((size_t)w * K2 + k0) * 5 * NP + (size_t)e * N;
    for (int idx = tid; idx < Kc * 5 * N; idx += nt) hs[idx] = Hw[(size_t)(idx / N) * NP + idx % N];
    __syncthreads();
    for (int idx = tid; idx < S.nch * Kc * 5; idx += nt) {
        const int kc = idx % (5 * Kc), s = idx / (5 * Kc);
        const int j0 = s == 0 ? 0 : S.n_up, ns = s == 0 ? S.n_up : S.n_dn;
        T v = 0;
        for (int j = j0; j < j0 + ns; ++j) v += hs[kc * N + j];
        sums[idx] = v / T(ns);
    }
    __syncthreads();
    // h2[j][e] depends on r = x_j - x_e: d/dx_j = +d/dr, d/dx_e = -d/dr
    T* Ge = G + ((size_t)(w * N + e) * ldg + row0 + k0) * P;
    // a thread produces four consecutive slots of one row (32-byte store; one division per four elements)
    typedef T vec4 __attribute__((ext_vector_type(4)));
    const int QP = P / 4;
    for (int idx = tid; idx < S.nch * Kc * QP; idx += nt) {
        const int row = idx / QP, sq = idx - row * QP, k = row % Kc, s = row / Kc;
        const int j0 = s == 0 ? 0 : S.n_up, ns = s == 0 ? S.n_up : S.n_dn;
        if (skip) {
            const int t = sq >> 2, lo = (2 + 3 * j0) >> 4, hi = (4 + 3 * (j0 + ns - 1)) >> 4;
            if (!(t == 0 || t == ((2 + 3 * e) >> 4) || t == ((4 + 3 * e) >> 4) || (t >= lo && t <= hi))) continue;
        }
        const T inv = T(1) / T(ns);
        const T* sm = sums + (s * Kc + k) * 5;
        vec4 v;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int slot = 4 * sq + u;
            T x = 0;
            if (slot == 0) x = sm[0];
            else if (slot == 1) x = sm[4];
            else if (slot < S.D) {
                const int j = (slot - 2) / 3, c = (slot - 2) - 3 * j;
                if (j == e) x = -sm[1 + c];
                else if (j >= j0 && j < j0 + ns) x = hs[(k * 5 + 1 + c) * N + j] * inv;
            }
            v[u] = x;
        }
        *reinterpret_cast<vec4*>(Ge + ((size_t)s * K2 + k) * P + 4 * sq) = v;
    }
}

// Residual of a FIRST layer as wide as its input features (network.py:525: hidden_single[0] == nf x atoms; round 6).  The residual
// kernels take the residual rows from the operand ring (K % 16 == 0, as many residual rows as device features); the layer-0 input has
// K = features + pair-mean rows (64 k + 4 or + 8) and -- for widths that are not multiples of 64 -- fewer feature rows than the
// zero-padded layer has outputs.  So layer 0 runs WITHOUT its residual (EPI 1 / 3) and this kernel finishes it in place:
//   out[tile][n][:] = ((n < n_res ? in[tile][n][:] : 0) + out[tile][n][:]) / sqrt 2,   n < Nout
// (a rare architecture: bandwidth-bound, one extra pass over the layer's output).  grid (tiles, walkers | groups), block 256.
template <typename T>
__global__ void __launch_bounds__(256) k_layer_res_add(const T* __restrict__ Xin, size_t x_ws, size_t x_ts, T* __restrict__ Gout, size_t g_ws,
                                                       size_t g_ts, int n_res, int Nout, int P) {
    const T* xi = Xin + (size_t)blockIdx.y * x_ws + (size_t)blockIdx.x * x_ts;
    T* go = Gout + (size_t)blockIdx.y * g_ws + (size_t)blockIdx.x * g_ts;
    const T rs2 = T(0.70710678118654752440);
    for (int idx = threadIdx.x; idx < Nout * P; idx += blockDim.x) {
        const T r = idx < n_res * P ? xi[idx] : T(0);
        go[idx] = (r + go[idx]) * rs2;
    }
}

// The same for a first PAIR layer as wide as the pair features (network.py:527: hidden_double[0] == nf): the pair layer 0 runs without
// its residual and this kernel finishes it in place, on the 5-jets (energy chain) or on five walker columns (value chain) alike:
//   out[blk][n][c][pair] = ((n < n_res ? in[blk][n][c][pair] : 0) + out[blk][n][c][pair]) / sqrt 2,   n < Kout
// grid (ceil(Kout * 5 * NP / 256), walkers | 5-walker blocks), block 256.
template <typename T>
__global__ void __launch_bounds__(256) k_pair_res_add(const T* __restrict__ Hin, int Kin, T* __restrict__ Hout, int Kout, int n_res, int NP) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)Kout * 5 * NP) return;
    const T* hi = Hin + (size_t)blockIdx.y * Kin * 5 * NP;
    T* ho = Hout + (size_t)blockIdx.y * Kout * 5 * NP;
    const T r = idx < (size_t)n_res * 5 * NP ? hi[idx] : T(0);
    ho[idx] = (r + ho[idx]) * T(0.70710678118654752440);
}

// =====================================================================================
// 2b. two-electron stream layer  h2 <- res(h2, tanh(h2 W + b))   (network.py:525-528)
//     MFMA, C[n][(c,pair)] ; the five jet components of a pair sit in five accumulator tiles of the
//     same lane, so the tanh chain rule is lane-local.
// =====================================================================================
// (value chain only) PARTM != nullptr: the wave also leaves, per output feature / walker column, the running sums of its 16
// pairs at the end of every (electron, partner spin) segment -- up to PM_SLOTS segments meet a 16-pair tile when every spin has
// >= 8 electrons -- in PARTM[w][tile][slot][feature][column] (one 16-lane prefix sum per value, staged in LDS, written as one
// contiguous block); k_m2_combine_val takes differences and adds the (at most three) tiles of a segment in index order, so the
// partner means of the NEXT one-electron layer need no second pass over H2 and stay bit-reproducible.  Hout == nullptr:
// the layer output itself is not needed (last pair layer of a log-psi forward).
constexpr int PM_SLOTS = 3;
template <typename T, int NT2, bool RES, bool VAL>
__global__ void __launch_bounds__(256, ((RES && !VAL && sizeof(T) == 8 && NT2 == 2) ? 2 : 3)) k_two_layer(SysDev<T> S, const T* __restrict__ Hin, int Kin, const T* __restrict__ W,
                                                   const T* __restrict__ bias, T* __restrict__ Hout, T* __restrict__ PARTM = nullptr) {
    typedef typename Acc4<T>::type acc_t;
    const int w = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + wave, NP = S.NP, Kout = 16 * NT2;
    if (pt * 16 >= NP) return;
    const int lr = lane & 15, lq = lane >> 4;
    const T* Hw = Hin + (size_t)w * Kin * 5 * NP + pt * 16 + lr;
    acc_t acc[NT2][5];
#pragma unroll
    for (int a = 0; a < NT2; ++a)
#pragma unroll
        for (int c = 0; c < 5; ++c) acc[a][c] = acc_t{0, 0, 0, 0};
    // float64 residual layers of the energy chain (Kin == Kout): the residual H[n][c][pair] of accumulator element (a, c, r) -- row
    // n = 16 a + lq + 4 r -- IS the B operand this very lane loads at k-step n / 4 = 4 a + r (operand row n % 4 = lq): all 4 NT2 k-steps'
    // operands are requested up front and kept (5 x 4 NT2 numbers), the epilogue reads no H a second time (6.0 -> 3.0 GB per launch,
    // profiles/r05_pmc_traffic.json).  (float32: the accumulator row map 4 lq + r puts that operand into another lane.)
    constexpr bool KEEP = RES && !VAL && sizeof(T) == 8;
    T bk[KEEP ? 4 * NT2 : 1][5];
    if constexpr (KEEP) {
#pragma unroll
        for (int ks = 0; ks < 4 * NT2; ++ks)
#pragma unroll
            for (int c = 0; c < 5; ++c) bk[ks][c] = Hw[(size_t)((4 * ks + lq) * 5 + c) * NP];
#pragma unroll
        for (int ks = 0; ks < 4 * NT2; ++ks) {
            T av[NT2];
#pragma unroll
            for (int a = 0; a < NT2; ++a) av[a] = W[(size_t)(4 * ks + lq) * Kout + 16 * a + lr];
#pragma unroll
            for (int a = 0; a < NT2; ++a)
#pragma unroll
                for (int c = 0; c < 5; ++c) acc[a][c] = mfma16(av[a], bk[ks][c], acc[a][c]);
        }
    } else
    for (int ks = 0; ks < Kin / 4; ++ks) {
        T av[NT2], bv[5];
#pragma unroll
        for (int a = 0; a < NT2; ++a) av[a] = W[(size_t)(4 * ks + lq) * Kout + 16 * a + lr];
#pragma unroll
        for (int c = 0; c < 5; ++c) bv[c] = Hw[(size_t)((4 * ks + lq) * 5 + c) * NP];
#pragma unroll
        for (int a = 0; a < NT2; ++a)
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[a][c] = mfma16(av[a], bv[c], acc[a][c]);
    }
    const T rs2 = T(0.70710678118654752440);
    T* Ho = Hout + (size_t)w * Kout * 5 * NP + pt * 16 + lr;
    __shared__ T pm_stage[VAL ? 4 * PM_SLOTS * 16 * NT2 * 5 : 1];
    T* pm_lds = pm_stage + (VAL ? wave * (PM_SLOTS * Kout * 5) : 0);
    bool pm_valid = false, pm_last = false;
    int pm_slot = 0;
    if (VAL && PARTM) {
        const int N = S.N, nch = S.nch;
        auto seg_of = [&](int p) { const int e = p / N, j = p - e * N; return p < N * N ? e * nch + (nch > 1 && j >= S.n_up ? 1 : 0) : -1; };
        const int p = pt * 16 + lr, sg = seg_of(p);
        pm_valid = sg >= 0;
        pm_slot = sg - seg_of(pt * 16);
        pm_last = pm_valid && (lr == 15 || seg_of(p + 1) != sg);
    }
#pragma unroll
    for (int a = 0; a < NT2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 16 * a + acc_row<T>(lane, r);
            const T z0 = acc[a][0][r] + bias[n];
            const T y = ds_tanh(z0), d1 = 1 - y * y, d2 = -2 * y * d1;
            const T z1 = acc[a][1][r], z2 = acc[a][2][r], z3 = acc[a][3][r], z4 = acc[a][4][r];
            T o[5] = {y, d1 * z1, d1 * z2, d1 * z3, d1 * z4 + d2 * 2 * (z1 * z1 + z2 * z2 + z3 * z3)};
            if (VAL) {   // value chain: the five columns are five independent walkers
                o[1] = ds_tanh(z1 + bias[n]); o[2] = ds_tanh(z2 + bias[n]); o[3] = ds_tanh(z3 + bias[n]); o[4] = ds_tanh(z4 + bias[n]);
            }
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                T v = o[c];
                if (RES) v = ((KEEP ? bk[KEEP ? 4 * a + r : 0][c] : Hw[(size_t)(n * 5 + c) * NP]) + v) * rs2;
                if (Hout) Ho[(size_t)(n * 5 + c) * NP] = v;
                if (VAL && PARTM) {
                    // running sum over the tile's pairs; the last lane of every (electron, spin) segment keeps its prefix:
                    // slot q of the tile holds the sum of segments 0..q (k_m2_combine_val takes differences)
                    const T pre = row16_prefix(pm_valid ? v : T(0));
                    if (pm_last) pm_lds[(pm_slot * Kout + n) * 5 + c] = pre;
                }
            }
        }
    if (VAL && PARTM) {
        // PARTM[w][tile][slot][feature][column]: PM_SLOTS * Kout * 5 contiguous values per wave, written coalesced
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        T* pm = PARTM + ((size_t)w * (NP / 16) + pt) * (PM_SLOTS * Kout * 5);
        for (int i = lane; i < PM_SLOTS * Kout * 5; i += 64) pm[i] = pm_lds[i];
    }
}

// 2c. (round 6) pair layer + the spin means of its OUTPUT in one kernel (energy chain): k_two_layer followed by k_m2_expand of the
// next level read the layer's output back from memory (4.1 GB per 4096 bcc-Li walkers) -- and the last pair layer's output has no
// other reader at all.  Here a workgroup owns the EW * N pairs of EW consecutive electrons (one wave per 16 of them, the last one
// possibly partly idle: the products are a small part of this kernel), runs the layer exactly like k_two_layer, leaves the output jets in LDS and expands them
// into the pair-mean rows of the next one-electron layer exactly like k_m2_expand (same sums in the same order: bit-identical rows).
// Hout == nullptr: nobody reads the layer output itself.  grid (N / EW, walkers, feature splits), block 64 * ceil(EW * N / 16),
// LDS (Kout * 5 * EW * N + EW * nch * Kout * 5) * sizeof(T) with Kout = 16 NT2 the features of ONE split: a layer without residual (layer 0:
// a handful of input features) runs its 32 output features as two workgroups of 16 -- half the LDS, twice the workgroups in flight
// for a kernel whose phases (load, products, tanh, sums, expand) follow each other behind barriers.  EW = 2 where the layer output is
// written and N is an odd multiple of 8: a workgroup's stretch of every output row then begins and ends on a 128-byte line (with one
// electron per workgroup neighbouring workgroups -- on different XCDs -- each wrote part of a line).
template <typename T, int NT2, bool RES>
__global__ void __launch_bounds__(512) k_two_layer_expand(SysDev<T> S, const T* __restrict__ Hin, int Kin, const T* __restrict__ W,
                                                          const T* __restrict__ bias, T* __restrict__ Hout, T* __restrict__ G, int row0,
                                                          int ldg, int skip, int EW) {
    typedef typename Acc4<T>::type acc_t;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int w = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x, nt = blockDim.x;
    const int N = S.N, NP = S.NP, P = S.P, Kout = 16 * NT2, e0 = blockIdx.x * EW, PW = EW * N;
    const int KT = Kout * gridDim.z, n0 = Kout * blockIdx.z;      // all output features of the layer; this workgroup's first one
    const int lr = lane & 15, lq = lane >> 4;
    const bool valid = wave * 16 + lr < PW;
    const int pl = valid ? wave * 16 + lr : PW - 1;              // (idle lanes load the last pair again and store nothing)
    const T* Hw = Hin + (size_t)w * Kin * 5 * NP + e0 * N + pl;
    T* hs = reinterpret_cast<T*>(smem_raw);     // [Kout * 5][PW]  output jets of the workgroup's pairs
    T* sums = hs + Kout * 5 * PW;               // [EW][nch][Kout][5]
    acc_t acc[NT2][5];
#pragma unroll
    for (int a = 0; a < NT2; ++a)
#pragma unroll
        for (int c = 0; c < 5; ++c) acc[a][c] = acc_t{0, 0, 0, 0};
    constexpr bool KEEP = RES && sizeof(T) == 8;       // (see k_two_layer: the residual is the operand the lane loaded)
    T bk[KEEP ? 4 * NT2 : 1][5];
    if constexpr (KEEP) {
#pragma unroll
        for (int ks = 0; ks < 4 * NT2; ++ks)
#pragma unroll
            for (int c = 0; c < 5; ++c) bk[ks][c] = Hw[(size_t)((4 * ks + lq) * 5 + c) * NP];
#pragma unroll
        for (int ks = 0; ks < 4 * NT2; ++ks) {
            T av[NT2];
#pragma unroll
            for (int a = 0; a < NT2; ++a) av[a] = W[(size_t)(4 * ks + lq) * KT + n0 + 16 * a + lr];
#pragma unroll
            for (int a = 0; a < NT2; ++a)
#pragma unroll
                for (int c = 0; c < 5; ++c) acc[a][c] = mfma16(av[a], bk[ks][c], acc[a][c]);
        }
    } else
    for (int ks = 0; ks < Kin / 4; ++ks) {
        T av[NT2], bv[5];
#pragma unroll
        for (int a = 0; a < NT2; ++a) av[a] = W[(size_t)(4 * ks + lq) * KT + n0 + 16 * a + lr];
#pragma unroll
        for (int c = 0; c < 5; ++c) bv[c] = Hw[(size_t)((4 * ks + lq) * 5 + c) * NP];
#pragma unroll
        for (int a = 0; a < NT2; ++a)
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[a][c] = mfma16(av[a], bv[c], acc[a][c]);
    }
    const T rs2 = T(0.70710678118654752440);
    T* Ho = (Hout && valid) ? Hout + ((size_t)w * KT + n0) * 5 * NP + e0 * N + pl : nullptr;
#pragma unroll
    for (int a = 0; a < NT2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = 16 * a + acc_row<T>(lane, r);
            const T z0 = acc[a][0][r] + bias[n0 + n];
            const T y = ds_tanh(z0), d1 = 1 - y * y, d2 = -2 * y * d1;
            const T z1 = acc[a][1][r], z2 = acc[a][2][r], z3 = acc[a][3][r], z4 = acc[a][4][r];
            const T o[5] = {y, d1 * z1, d1 * z2, d1 * z3, d1 * z4 + d2 * 2 * (z1 * z1 + z2 * z2 + z3 * z3)};
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                T v = o[c];
                if (RES) v = ((KEEP ? bk[KEEP ? 4 * a + r : 0][c] : Hw[(size_t)((n0 + n) * 5 + c) * NP]) + v) * rs2;
                if (Ho) Ho[(size_t)(n * 5 + c) * NP] = v;
                if (valid) hs[(n * 5 + c) * PW + pl] = v;
            }
        }
    __syncthreads();
    const int nch = S.nch;
    for (int idx = tid; idx < EW * nch * Kout * 5; idx += nt) {
        const int kc = idx % (5 * Kout), es = idx / (5 * Kout), s = es % nch, el = es / nch;
        const int j0 = s == 0 ? 0 : S.n_up, ns = s == 0 ? S.n_up : S.n_dn;
        T v = 0;
        for (int j = j0; j < j0 + ns; ++j) v += hs[kc * PW + el * N + j];
        sums[idx] = v / T(ns);
    }
    __syncthreads();
    // (a thread per four consecutive slots of one row, rows walked in index order: a per-lane slot descriptor with two rows per
    //  wave pass measured slower, EXPERIMENTS.md)
    typedef T vec4 __attribute__((ext_vector_type(4)));
    const int QP = P / 4;
    for (int idx = tid; idx < EW * nch * Kout * QP; idx += nt) {
        const int row = idx / QP, sq = idx - row * QP, k = row % Kout, es = row / Kout, s = es % nch, el = es / nch, e = e0 + el;
        const int j0 = s == 0 ? 0 : S.n_up, ns = s == 0 ? S.n_up : S.n_dn;
        if (skip) {
            const int t = sq >> 2, lo = (2 + 3 * j0) >> 4, hi = (4 + 3 * (j0 + ns - 1)) >> 4;
            if (!(t == 0 || t == ((2 + 3 * e) >> 4) || t == ((4 + 3 * e) >> 4) || (t >= lo && t <= hi))) continue;
        }
        const T inv = T(1) / T(ns);
        const T* sm = sums + (es * Kout + k) * 5;
        vec4 v;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int slot = 4 * sq + u;
            T x = 0;
            if (slot == 0) x = sm[0];
            else if (slot == 1) x = sm[4];
            else if (slot < S.D) {
                const int j = (slot - 2) / 3, c = (slot - 2) - 3 * j;
                if (j == e) x = -sm[1 + c];
                else if (j >= j0 && j < j0 + ns) x = hs[(k * 5 + 1 + c) * PW + el * N + j] * inv;
            }
            v[u] = x;
        }
        *reinterpret_cast<vec4*>(G + ((size_t)(w * N + e) * ldg + row0 + s * KT + n0 + k) * P + 4 * sq) = v;
    }
}

// rows [row0, row0 + nch*K2) of G (value chain) from the per-tile segment sums of k_two_layer: mean over the partners j of
// spin s of h2[j][e] = (sum over the tiles that meet segment (e, s), in tile order) / n_s.   grid (N, groups), block 256.
// Reads walk (feature, column-in-5) fastest -- contiguous in PARTM --, the result is transposed through LDS so that the
// G rows are written column-contiguous.
// grid.z: chunks of RC rows (RC divides K2: a chunk lies inside one partner spin) -- with few walker groups a workgroup per
// (electron, group) was a 20-iteration chain of dependent loads on 168 CUs.
template <typename T>
__global__ void __launch_bounds__(256) k_m2_combine_val(SysDev<T> S, const T* __restrict__ PARTM, int K2, T* __restrict__ G, int row0, int RC) {
    constexpr int PV_ = 80;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* out = reinterpret_cast<T*>(smem_raw);                      // [RC][PV]
    const int e = blockIdx.x, g = blockIdx.y, N = S.N, nt = S.NP / 16, nch = S.nch;
    const int r0 = blockIdx.z * RC, sp = r0 / K2, k0 = r0 - sp * K2;
    const size_t blk = (size_t)PM_SLOTS * K2 * 5;
    const int a = e * N + (sp == 0 ? 0 : S.n_up), ns = sp == 0 ? S.n_up : S.n_dn, b = a + ns;      // pairs [a, b)
    const int seg = e * nch + sp;
    for (int idx = threadIdx.x; idx < RC * PV_; idx += blockDim.x) {
        const int cc = idx % 5, kk = (idx / 5) % RC, w5 = idx / (5 * RC), k = k0 + kk;
        const T* pw = PARTM + ((size_t)g * (PV_ / 5) + w5) * nt * blk;
        T v = 0;
        for (int pt = a / 16; pt <= (b - 1) / 16; ++pt) {
            const int p0 = pt * 16, e0 = p0 / N, j0 = p0 - e0 * N, seg0 = e0 * nch + (nch > 1 && j0 >= S.n_up ? 1 : 0);
            const int q = seg - seg0;                       // slots hold running sums over the tile's segments
            v += pw[(size_t)pt * blk + ((size_t)q * K2 + k) * 5 + cc] - (q > 0 ? pw[(size_t)pt * blk + ((size_t)(q - 1) * K2 + k) * 5 + cc] : T(0));
        }
        out[kk * PV_ + w5 * 5 + cc] = v / T(ns);
    }
    __syncthreads();
    T* Ge = G + ((size_t)(g * N + e) * S.ldk + row0 + r0) * PV_;
    for (int idx = threadIdx.x; idx < RC * PV_; idx += blockDim.x) Ge[idx] = out[idx];
}

// (3. one-electron stream layer and 4. orbital head: ds_gemm.h)

// =====================================================================================
// 5a. inverse + log det of every (walker, spin, det) value matrix: Gauss-Jordan, partial pivoting,
//     one wave per matrix, augmented matrix in LDS.   (replaces jnp.linalg.slogdet, network.py:375-392)
// =====================================================================================
template <typename T>
__global__ void __launch_bounds__(64) k_det_inverse(SysDev<T> S, const T* __restrict__ MOUT, size_t mout_stride, size_t mout_off,
                                                    int sp, T* __restrict__ MINV, size_t minv_stride, size_t minv_off,
                                                    T* __restrict__ DETS, size_t dets_stride, size_t dets_off, int P,
                                                    int es, int cols_per_group, int ms) {
    // P = slots per matrix element (sets the per-determinant block size), es = stride between consecutive
    // (elec, orb, re/im) entries of the value:
    //   forward-Laplacian chain: P = S.P, es = 16 (slot 0 of slot tile 0), cols_per_group = 1;
    //   value chain: P = es = PV, cols_per_group = PV (walker w = column w % PV of group w / PV)
    // ms = element stride of the inverse: 1 (per-walker blocks, minv_stride apart) or PV (interleaved like MOUT)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Cx<T>* aug = reinterpret_cast<Cx<T>*>(smem_raw);   // [n][2n]
    const int kdet = blockIdx.x, w = blockIdx.y, lane = threadIdx.x;
    // MINV == nullptr: log det only (value chain, matrices beyond the register LU kernels): plain LU on the n x n matrix -- no identity
    // block, only the rows below the pivot are reduced (a sixth of the Gauss-Jordan work)
    const bool lu_only = MINV == nullptr;
    const int n = S.det_n[sp], n2 = lu_only ? n : 2 * n;      // sp = determinant channel
    int* piv_p = reinterpret_cast<int*>(aug + n * n2);  // all LDS in the one dynamic region (16-B aligned base)
    const T* Mw = MOUT + (size_t)(w / cols_per_group) * mout_stride + mout_off + (size_t)kdet * n * n * 2 * P + w % cols_per_group;
    for (int idx = lane; idx < n * n; idx += 64) {
        const int r = idx / n, c = idx % n;
        aug[r * n2 + c] = Cx<T>(Mw[(size_t)(idx * 2) * es], Mw[(size_t)(idx * 2 + 1) * es]);
        if (!lu_only) aug[r * n2 + n + c] = Cx<T>(r == c ? T(1) : T(0), T(0));
    }
    __syncthreads();
    T logabs = 0;
    Cx<T> ph(1, 0);
    for (int j = 0; j < n; ++j) {
        // pivot search over rows j..n-1 (first maximum, like LAPACK)
        T best = -1; int bi = j;
        for (int r = j + lane; r < n; r += 64) {
            const T m = cx_abs2(aug[r * n2 + j]);
            if (m > best) { best = m; bi = r; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const T ob = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) *piv_p = bi;
        __syncthreads();
        const int pv = *piv_p;
        if (pv != j) {
            for (int c = lane; c < n2; c += 64) {
                const Cx<T> t = aug[j * n2 + c]; aug[j * n2 + c] = aug[pv * n2 + c]; aug[pv * n2 + c] = t;
            }
            ph = Cx<T>(-ph.re, -ph.im);
        }
        __syncthreads();
        const Cx<T> d = aug[j * n2 + j];
        const T ad = ds_sqrt(cx_abs2(d));
        logabs += ds_log(ad);
        ph = ph * Cx<T>(d.re / ad, d.im / ad);
        const Cx<T> dinv = cx_inv(d);
        __syncthreads();
        for (int c = lane; c < n2; c += 64) aug[j * n2 + c] = aug[j * n2 + c] * dinv;
        __syncthreads();
        // eliminate column j from every other row (log det only: from the rows below, columns to the right)
        if (lu_only) {
            const int m = n - 1 - j;
            for (int idx = lane; idx < m * m; idx += 64) {
                const int r = j + 1 + idx / m, c = j + 1 + idx % m;
                aug[r * n2 + c] = aug[r * n2 + c] - aug[r * n2 + j] * aug[j * n2 + c];
            }
            __syncthreads();
            continue;
        }
        for (int idx = lane; idx < n * n2; idx += 64) {
            const int r = idx / n2, c = idx % n2;
            if (r == j || c == j) continue;
            aug[idx] = aug[idx] - aug[r * n2 + j] * aug[j * n2 + c];
        }
        __syncthreads();
        for (int r = lane; r < n; r += 64)
            if (r != j) aug[r * n2 + j] = Cx<T>(0, 0);
        __syncthreads();
    }
    if (MINV) {
        T* Iw = MINV + (size_t)(w / cols_per_group) * minv_stride + minv_off + (size_t)kdet * n * n * 2 * ms + w % cols_per_group;
        for (int idx = lane; idx < n * n; idx += 64) {
            const int r = idx / n, c = idx % n;
            Iw[(size_t)(2 * idx) * ms] = aug[r * n2 + n + c].re;
            Iw[(size_t)(2 * idx + 1) * ms] = aug[r * n2 + n + c].im;
        }
    }
    if (lane == 0) {
        T* dw = DETS + (size_t)w * dets_stride + dets_off + (size_t)kdet * 4;
        dw[0] = logabs;
        dw[1] = ds_atan2(ph.im, ph.re);
    }
}

// =====================================================================================
// 5b. traces:  Y_d = d_dM . M^-1 ;  TR[d] = tr Y_d  (d = Laplacian slot and every direction),
//              trY2 = sum_{d>=2} tr(Y_d Y_d)
//     block (walker, spin, det): SP slots x 256/SP matrix rows per pass, Y rows exchanged through LDS.
// =====================================================================================
template <typename T, int NMAX, int SP>
__global__ void __launch_bounds__(256) k_det_trace(SysDev<T> S, const T* __restrict__ MOUT, size_t mout_stride, size_t mout_off,
                                                   int sp, const T* __restrict__ MINV, size_t minv_stride, size_t minv_off,
                                                   T* __restrict__ TR, size_t tr_stride, size_t tr_off,
                                                   T* __restrict__ DETS, size_t dets_stride, size_t dets_off) {
    // blockDim.x = SP * RP with RP = rows per pass (n rounded up so that the block is whole waves):
    // thread (dl, il) owns slot d0 + dl and matrix row ib + il.
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int kdet = blockIdx.x, w = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x, RP = nthr / SP;
    const int P = S.P, n = S.det_n[sp];                  // sp = determinant channel
    Cx<T>* minv = reinterpret_cast<Cx<T>*>(smem_raw);   // [n][n]
    Cx<T>* ybuf = minv + n * n;                          // [SP slots][n*n + 1]: +1 complex spreads the slots over the LDS banks
    const int ys = n * n + 1;
    Cx<T>* red = ybuf + SP * ys;                         // [nthr]
    const T* Iw = MINV + (size_t)w * minv_stride + minv_off + (size_t)kdet * n * n * 2;
    for (int idx = tid; idx < n * n; idx += nthr) minv[idx] = Cx<T>(Iw[2 * idx], Iw[2 * idx + 1]);
    __syncthreads();
    const T* Mw = MOUT + (size_t)w * mout_stride + mout_off + (size_t)kdet * n * n * 2 * P;
    T* Tw = TR + (size_t)w * tr_stride + tr_off + (size_t)kdet * 2 * P;
    const int dl = tid % SP, il = tid / SP;
    Cx<T> y2(0, 0);
    // slot 0 (the value: Y = 1) and the zero padding beyond D are skipped: start at slot 1
    for (int d0 = 1; d0 < S.D; d0 += SP) {
        const int d = d0 + dl;
        const bool live = d < S.D;
        for (int ib = 0; ib < n; ib += RP) {
            const int i = ib + il;
            if (i < n) {
                Cx<T> yv[NMAX];
#pragma unroll
                for (int e = 0; e < NMAX; ++e) yv[e] = Cx<T>(0, 0);
                if (live) {
                    const T* mp = Mw + ((size_t)(d >> 4) * n * n * 2 + (size_t)i * n * 2) * 16 + (d & 15);   // slot-tile major MOUT
                    // all loads of this matrix row are issued before the first FMA (one exposed latency per row)
                    T mre[NMAX], mim[NMAX];
#pragma unroll
                    for (int m = 0; m < NMAX; ++m)
                        if (m < n) { mre[m] = mp[(size_t)(2 * m) * 16]; mim[m] = mp[(size_t)(2 * m + 1) * 16]; }
#pragma unroll
                    for (int m = 0; m < NMAX; ++m) {
                        if (m < n) {
                            const Cx<T> dm(mre[m], mim[m]);
#pragma unroll
                            for (int e = 0; e < NMAX; ++e)
                                if (e < n) yv[e] = cx_fma(dm, minv[m * n + e], yv[e]);
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < NMAX; ++e)
                    if (e < n) ybuf[dl * ys + i * n + e] = yv[e];
            }
        }
        __syncthreads();
        Cx<T> trc(0, 0);
        for (int ib = 0; ib < n; ib += RP) {
            const int i = ib + il;
            if (i < n) {
                trc = trc + ybuf[dl * ys + i * n + i];
                if (d >= 2)
                    for (int e = 0; e < n; ++e) y2 = cx_fma(ybuf[dl * ys + i * n + e], ybuf[dl * ys + e * n + i], y2);
            }
        }
        // reduce the trace over the row-groups holding the same slot
        red[tid] = trc;
        __syncthreads();
        if (il == 0 && live) {
            Cx<T> t(0, 0);
            for (int g = 0; g < RP; ++g) t = t + red[g * SP + dl];
            Tw[d] = t.re;
            Tw[P + d] = t.im;
        }
        __syncthreads();
    }
    red[tid] = y2;
    __syncthreads();
    if (tid == 0) {
        Cx<T> t(0, 0);
        for (int g = 0; g < nthr; ++g) t = t + red[g];
        T* dw = DETS + (size_t)w * dets_stride + dets_off + (size_t)kdet * 4;
        dw[2] = t.re;
        dw[3] = t.im;
    }
}

// =====================================================================================
// 5b'. the same traces on the matrix cores (used when 2n is a multiple of 4 and the Y chunk fits LDS):
//     Y_d[i][e] = sum_m d_dM[i][m] Minv[m][e] is, in real arithmetic, C[(e,ri')][slot] = sum_{(m,ri)} A * X with
//     X = the MOUT rows of electron i ([m][re,im][P], already k-major / slot-contiguous) and A the 2n x 2n real
//     expansion of Minv^T, built on the fly from MINV.  One workgroup per (walker, channel, det) walks the slot
//     tiles; the four waves split the electrons, park Y of one slot tile in LDS, then all threads form
//     tr Y_d and sum_{i,e} Y_d[i][e] Y_d[e][i] with conflict-free LDS reads.
// =====================================================================================
// SW = slots of a 16-slot MFMA tile parked in LDS per pass: 16, or 8 (two passes over the tile, the products are
// recomputed) when n * 2n * 16 elements do not fit the LDS.
// NW = waves per workgroup: 8 for the matrices whose Y tile leaves room for one workgroup per CU only (two waves per SIMD).
// NFIX = the matrix size when it is known at compile time (12 and 24: the 24- and 48-electron cells of the benchmark configurations),
// 0 = read from the descriptor: with a constant n the index arithmetic of the pair sums (a division and a remainder per term) and
// the LDS addresses fold into constants.
template <typename T, int NT, int SW, int NW = 4, int NFIX = 0>
__global__ void __launch_bounds__(64 * NW) k_det_trace_mfma(SysDev<T> S, const T* __restrict__ MOUT, size_t mout_stride, size_t mout_off,
                                                        int ch, const T* __restrict__ MINV, size_t minv_stride, size_t minv_off,
                                                        T* __restrict__ TR, size_t tr_stride, size_t tr_off,
                                                        T* __restrict__ DETS, size_t dets_stride, size_t dets_off,
                                                        unsigned long long* __restrict__ tl = nullptr) {
    typedef typename Acc4<T>::type acc_t;
    // tl (kernel development): per-wave cycle totals of workgroup (0, 0), as in k_det_trace_mfma_split
    const bool stamp = tl && blockIdx.x == 0 && blockIdx.y == 0;
    long long c_setup = 0, c_prod = 0, c_bar1 = 0, c_pairs = 0, c_bar2 = 0, c_red = 0, c_t = 0;
    const long long c_begin = stamp ? clock64() : 0;
    constexpr int KSMAX = 4 * NT;                         // 2n <= 16 NT  ->  2n / 4 <= 4 NT k-steps
    constexpr int NTHR = 64 * NW, NG = NTHR / SW;         // NG thread groups in the trace phase
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int kdet = blockIdx.x, w = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const int P = S.P, n = NFIX ? NFIX : S.det_n[ch], n2 = 2 * n, nks = n2 / 4;
    T* Y = reinterpret_cast<T*>(smem_raw);               // [n][2n][SW]
    Cx<T>* red = reinterpret_cast<Cx<T>*>(Y + (size_t)n * n2 * SW);   // [NTHR]
    const T* Iw = MINV + (size_t)w * minv_stride + minv_off + (size_t)kdet * n * n * 2;
    const T* Mw = MOUT + (size_t)w * mout_stride + mout_off + (size_t)kdet * n * n * 2 * P;
    T* Tw = TR + (size_t)w * tr_stride + tr_off + (size_t)kdet * 2 * P;
    // A fragments (slot independent): A[n' = (e, ri')][k' = (m, ri)]
    T af[NT][KSMAX];
    if constexpr (NFIX > 0) {
        // compile-time size: every fragment element is ONE unconditional 2-element load (clamped index, selected afterwards) -- all in
        // flight together instead of a branch and a wait per element (setup: 10 k of the 70 k cycles a workgroup lives, round 6)
        typedef T vec2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int ks = 0; ks < KSMAX; ++ks) {
                if (4 * ks >= n2) { af[nt][ks] = 0; continue; }
                const int np = 16 * nt + lr, kp = 4 * ks + lq, npc = np < n2 ? np : n2 - 1;
                const int e = npc >> 1, rip = npc & 1, m = kp >> 1, ri = kp & 1;
                const vec2 c = *reinterpret_cast<const vec2*>(Iw + (m * n + e) * 2);
                const T v = (ri == rip) ? c[0] : (rip == 0 ? -c[1] : c[1]);
                af[nt][ks] = np < n2 ? v : T(0);
            }
    } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            const int np = 16 * nt + lr, kp = 4 * ks + lq;
            T v = 0;
            if (np < n2 && ks < nks) {
                const int e = np >> 1, rip = np & 1, m = kp >> 1, ri = kp & 1;
                const T re = Iw[(m * n + e) * 2], im = Iw[(m * n + e) * 2 + 1];
                v = (ri == rip) ? re : (rip == 0 ? -im : im);
            }
            af[nt][ks] = v;
        }
    }
    Cx<T> y2(0, 0);
    const int d = tid % SW, g = tid / SW;
    // The wave's rows over ALL slot tiles form one sequence q = (pass, row): the B operands of row q + 2 are requested before
    // the MFMAs of row q, across the tile boundaries too, so the trace phase of a tile hides the latency of the next tile's
    // first rows (large matrices run one workgroup per CU: nothing else would).
    const int nsp = (P / 16) * (16 / SW), R = wave < n ? (n - wave + NW - 1) / NW : 0, total = R * nsp;
    // Known matrix size with the same number of rows for every wave (n = 12 on 4 waves, 24 on 8: three rows each): ALL rows of
    // the next slot tile are requested before the products of the current one -- a whole tile (products, pair sums, two barriers)
    // of latency cover.  With the rows requested two ahead, three rows of 12 MFMAs per tile left the products phase waiting for
    // memory: 11 k cycles per tile for 2.3 k cycles of MFMAs (tools/trace_timeline.py).
    // (only while the two operand tiles stay within ~96 VGPRs: at n = 24 in float64 they take 144 next to 72 of A fragments, the
    //  kernel spills and the products phase becomes 2.5 times slower)
    constexpr bool TILE_PF = NFIX > 0 && NFIX % NW == 0 && SW == 16 && 2 * (NFIX / NW) * KSMAX * ((int)sizeof(T) / 4) <= 96;
    constexpr int RW = TILE_PF ? NFIX / NW : 1;
    T cur[RW][KSMAX], nxt[RW][KSMAX];
    auto load_tile = [&](T (&b)[RW][KSMAX], int st) {
        if (st < nsp) {
#pragma unroll
            for (int rr = 0; rr < RW; ++rr) {
                const T* xp = Mw + ((size_t)st * n * n2 + (size_t)(wave + NW * rr) * n2 + lq) * 16 + lr;
#pragma unroll
                for (int ks = 0; ks < KSMAX; ++ks) b[rr][ks] = ks < nks ? xp[(size_t)(4 * ks) * 16] : T(0);
            }
        }
    };
    T bc[KSMAX], bn[KSMAX], bm[KSMAX];
    auto load_q = [&](T (&b)[KSMAX], int q) {
        if (q < total) {
            const int sp = q / R, i = wave + NW * (q - sp * R), st = sp / (16 / SW);
            const T* xp = Mw + ((size_t)st * n * n2 + (size_t)i * n2 + lq) * 16 + lr;      // contiguous n*2n*16 chunk per slot tile
#pragma unroll
            for (int ks = 0; ks < KSMAX; ++ks) b[ks] = ks < nks ? xp[(size_t)(4 * ks) * 16] : T(0);
        }
    };
    if (TILE_PF) load_tile(cur, 0);
    else { load_q(bc, 0); load_q(bn, 1); }
    int q = 0;
    // pair terms of this thread (compile-time even n, full slot tiles): pi = g, g + NG, ... -> LDS offsets of Y[i][e] and Y[e][i]
    // (measured: graphene's n = 24 instance 4.05 -> 3.79 ms per 512 walkers, bcc-Li's n = 12 instance 3.2 -> 3.75 ms per 4096 -- the offsets
    //  cost it a wave of occupancy: used from n = 24 on)
    constexpr bool FASTP = NFIX >= 24 && (NFIX % 2) == 0 && SW == 16;
    constexpr int NPAIR = FASTP ? (NFIX / 2) * (NFIX + 1) : 1, NPT = FASTP ? (NPAIR + NG - 1) / NG : 1;
    int poA[NPT], poB[NPT];
    if constexpr (FASTP) {
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int pi = g + NG * k;
            const int r = pi / (NFIX + 1), t = pi - r * (NFIX + 1);
            const int i = t < NFIX - r ? r : NFIX - 1 - r, e = t < NFIX - r ? r + t : i + (t - (NFIX - r));
            poA[k] = pi < NPAIR ? (i * n2 + 2 * e) * SW + d : -1;
            poB[k] = (e * n2 + 2 * i) * SW + d;
        }
    }
    // Compile-time even n below the register-table sizes (n = 12, the 24-electron benchmark cell): the (i, e) -> LDS offsets of the
    // pair terms in a TABLE IN LDS (behind the reduction buffer), one broadcast read per term instead of a division and a remainder
    // per term and tile; tr Y_d straight from the diagonal by one thread per slot -- no reduction buffer, one barrier less per tile
    // (round 6: pairs 17 k -> ... of the 78 k cycles a workgroup lives, tools/trace_timeline.py)
    constexpr bool TABP = NFIX > 0 && !FASTP && (NFIX % 2) == 0 && SW == 16;
    constexpr int TNPAIR = TABP ? (NFIX / 2) * (NFIX + 1) : 1, TNPT = TABP ? (TNPAIR + NG - 1) / NG : 1;
    unsigned* ptab = reinterpret_cast<unsigned*>(red);      // (table instances: the table sits where the per-tile reduction buffer would; the final
                                                             //  reduction of y2 uses the then dead Y -- 37.2 KB per workgroup, four per CU)
    if constexpr (TABP) {
        for (int pi = tid; pi < TNPAIR; pi += NTHR) {
            const int r = pi / (NFIX + 1), t = pi - r * (NFIX + 1);
            const int i = t < NFIX - r ? r : NFIX - 1 - r, e = t < NFIX - r ? r + t : i + (t - (NFIX - r));
            ptab[pi] = (unsigned)((i * n2 + 2 * e) * SW) | ((unsigned)((e * n2 + 2 * i) * SW) << 16);
        }
        // (visible to all waves behind the first tile's product barrier)
    }
    if (stamp) c_setup = clock64() - c_begin;
    for (int sp = 0; sp < nsp; ++sp) {
        const int st = sp / (16 / SW), half = sp % (16 / SW);
        if (stamp) c_t = clock64();
        if (TILE_PF) {
            // (table instances: no second operand tile -- the next tile's rows are requested into `cur` itself right behind this tile's
            //  products, with the pair sums and two barriers as cover: 48 registers less, a fourth workgroup per CU)
            if constexpr (!TABP) load_tile(nxt, sp + 1);
#pragma unroll
            for (int rr = 0; rr < RW; ++rr) {
                const int i = wave + NW * rr;
                acc_t acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = acc_t{0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KSMAX; ++ks) {
                    if (ks < nks) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(af[nt][ks], cur[rr][ks], acc[nt]);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int np = 16 * nt + acc_row<T>(lane, r);
                        if (np < n2) Y[((size_t)i * n2 + np) * SW + lr] = acc[nt][r];
                    }
            }
            if constexpr (TABP) load_tile(cur, sp + 1);
        } else
        for (int i = wave; i < n; i += NW, ++q) {
            load_q(bm, q + 2);
            acc_t acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = acc_t{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KSMAX; ++ks) {
                if (ks < nks) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(af[nt][ks], bc[ks], acc[nt]);
                }
            }
#pragma unroll
            for (int ks = 0; ks < KSMAX; ++ks) { bc[ks] = bn[ks]; bn[ks] = bm[ks]; }
            if (SW == 16 || (lr / SW) == half) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int np = 16 * nt + acc_row<T>(lane, r);
                        if (np < n2) Y[((size_t)i * n2 + np) * SW + lr % SW] = acc[nt][r];
                    }
            }
        }
        if (stamp) { const long long c = clock64(); c_prod += c - c_t; c_t = c; }
        __syncthreads();
        if (stamp) { const long long c = clock64(); c_bar1 += c - c_t; c_t = c; }
        const int slot = 16 * st + SW * half + d;
        const bool live = slot >= 1 && slot < S.D;
        if constexpr (FASTP) {
            // compile-time matrix size: the (i, e) of a thread's pair terms were worked out before the tile loop; tr Y_d is summed by
            // ONE thread per slot (group 0) straight from the diagonal -- no reduction buffer, one barrier less per tile
#pragma unroll
            for (int k = 0; k < NPT; ++k)
                if (poA[k] >= 0 && slot >= 2) {
                    const Cx<T> yie(Y[poA[k]], Y[poA[k] + SW]), yei(Y[poB[k]], Y[poB[k] + SW]);
                    y2 = cx_fma((poA[k] == poB[k] ? T(1) : T(2)) * yie, yei, y2);
                }
            if (g == 0 && live) {
                Cx<T> t(0, 0);
#pragma unroll
                for (int i = 0; i < (NFIX > 0 ? NFIX : 1); ++i) t = t + Cx<T>(Y[((size_t)i * n2 + 2 * i) * SW + d], Y[((size_t)i * n2 + 2 * i + 1) * SW + d]);
                Tw[slot] = t.re;
                Tw[P + slot] = t.im;
            }
            if (stamp) { const long long c = clock64(); c_pairs += c - c_t; c_t = c; }
            __syncthreads();
            if (stamp) { const long long c = clock64(); c_bar2 += c - c_t; c_t = c; }
            if (TILE_PF) {
#pragma unroll
                for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                    for (int ks = 0; ks < KSMAX; ++ks) cur[rr][ks] = nxt[rr][ks];
            }
            continue;
        }
        if constexpr (TABP) {
#pragma unroll
            for (int k = 0; k < TNPT; ++k) {
                const int pi = g + NG * k;
                if (pi < TNPAIR && slot >= 2) {
                    const unsigned ent = ptab[pi];
                    const int oa = (int)(ent & 0xffffu) + d, ob = (int)(ent >> 16) + d;
                    const Cx<T> yie(Y[oa], Y[oa + SW]), yei(Y[ob], Y[ob + SW]);
                    y2 = cx_fma((oa == ob ? T(1) : T(2)) * yie, yei, y2);
                }
            }
            if (g == NG - 1 && live) {      // (the last group has the fewest pair terms)
                Cx<T> t(0, 0);
#pragma unroll
                for (int i = 0; i < (NFIX > 0 ? NFIX : 1); ++i) t = t + Cx<T>(Y[((size_t)i * n2 + 2 * i) * SW + d], Y[((size_t)i * n2 + 2 * i + 1) * SW + d]);
                Tw[slot] = t.re;
                Tw[P + slot] = t.im;
            }
            if (stamp) { const long long c = clock64(); c_pairs += c - c_t; c_t = c; }
            __syncthreads();
            if (stamp) { const long long c = clock64(); c_bar2 += c - c_t; c_t = c; }
            continue;
        }
        Cx<T> trc(0, 0);
        // sum_{i,e} Y[i][e] Y[e][i] = sum_i Y[i][i]^2 + 2 sum_{i<e} Y[i][e] Y[e][i]: upper triangle only.  Rows r and
        // n-1-r together hold n+1 entries (n even); odd n walks the full square.
        const bool tri = (n & 1) == 0;
        const int npair = tri ? (n / 2) * (n + 1) : n * n;
#pragma unroll 4
        for (int pi = g; pi < npair; pi += NG) {
            int i, e;
            if (tri) {
                const int r = pi / (n + 1), t = pi - r * (n + 1);
                if (t < n - r) { i = r; e = r + t; } else { i = n - 1 - r; e = i + (t - (n - r)); }
            } else { i = pi / n; e = pi - i * n; }
            const Cx<T> yie(Y[((size_t)i * n2 + 2 * e) * SW + d], Y[((size_t)i * n2 + 2 * e + 1) * SW + d]);
            if (i == e) trc = trc + yie;
            if (slot >= 2) {
                const Cx<T> yei(Y[((size_t)e * n2 + 2 * i) * SW + d], Y[((size_t)e * n2 + 2 * i + 1) * SW + d]);
                const T wgt = (tri && i != e) ? T(2) : T(1);
                y2 = cx_fma(wgt * yie, yei, y2);
            }
        }
        if (stamp) { const long long c = clock64(); c_pairs += c - c_t; c_t = c; }
        red[tid] = trc;
        __syncthreads();
        if (stamp) { const long long c = clock64(); c_bar2 += c - c_t; c_t = c; }
        if (g == 0 && live) {
            Cx<T> t(0, 0);
            for (int u = 0; u < NG; ++u) t = t + red[u * SW + d];
            Tw[slot] = t.re;
            Tw[P + slot] = t.im;
        }
        __syncthreads();
        if (stamp) { const long long c = clock64(); c_red += c - c_t; c_t = c; }
        if (TILE_PF) {      // (here, not after the products: the wait for the next tile's rows then has the whole tile as cover)
#pragma unroll
            for (int rr = 0; rr < RW; ++rr)
#pragma unroll
                for (int ks = 0; ks < KSMAX; ++ks) cur[rr][ks] = nxt[rr][ks];
        }
    }
    if (stamp && lane == 0) {
        unsigned long long* o = tl + wave * 8;
        o[0] = c_setup; o[1] = c_prod; o[2] = c_bar1; o[3] = c_pairs; o[4] = c_bar2; o[5] = c_red; o[6] = clock64() - c_begin; o[7] = 0;
    }
    Cx<T>* fred = TABP ? reinterpret_cast<Cx<T>*>(Y) : red;      // (the last tile's barrier has passed: Y is dead)
    fred[tid] = y2;
    __syncthreads();
    if (tid == 0) {
        Cx<T> t(0, 0);
        for (int u = 0; u < NTHR; ++u) t = t + fred[u];
        T* dw = DETS + (size_t)w * dets_stride + dets_off + (size_t)kdet * 4;
        dw[2] = t.re;
        dw[3] = t.im;
    }
}

// =====================================================================================
// 5b''. the same for matrices whose 16-slot tile of Y does not fit the LDS (n = 32 in float64, n = 48 in float32), without
//     computing anything twice: the rows are split in halves I1 | I2 and the columns in C1 | C2 (n a multiple of 16):
//       pass A  rows I1, all columns           -> sum over (i, e) in I1 x I1 and the diagonal of I1
//       pass B  rows I2, columns C1, written over the (dead) I1 x C1 block -> 2 * sum over i in I1, e in I2
//       pass C  rows I2, columns C2, written over the (dead) I1 x C2 block -> sum over I2 x I2 and the diagonal of I2
//     LDS holds n/2 rows of a full 16-slot tile (the footprint of the half-slot mode it replaces); rows of I2 are loaded twice.
// =====================================================================================
// PF = rows requested ahead (2, or 1 where the A fragments leave no registers for a third operand set)
template <typename T, int NT, int NW = 4, int PF = 2>
__global__ void __launch_bounds__(64 * NW) k_det_trace_mfma_split(SysDev<T> S, const T* __restrict__ MOUT, size_t mout_stride, size_t mout_off,
                                                              int ch, const T* __restrict__ MINV, size_t minv_stride, size_t minv_off,
                                                              T* __restrict__ TR, size_t tr_stride, size_t tr_off,
                                                              T* __restrict__ DETS, size_t dets_stride, size_t dets_off,
                                                              unsigned long long* __restrict__ tl = nullptr) {
    // tl (kernel development): workgroup (0, 0) writes per-wave cycle totals of its phases -- [wave][0..7] = fragment setup, products,
    // wait at the barrier after them, pair sums, wait after them, per-tile trace reduction, whole kernel, 0
    typedef typename Acc4<T>::type acc_t;
    constexpr int KSMAX = 4 * NT, NTHR = 64 * NW, NG = NTHR / 16;
    const bool stamp = tl && blockIdx.x == 0 && blockIdx.y == 0;
    long long c_setup = 0, c_prod = 0, c_bar1 = 0, c_pairs = 0, c_bar2 = 0, c_red = 0, c_t = 0;
    const long long c_begin = stamp ? clock64() : 0;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int kdet = blockIdx.x, w = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, lq = lane >> 4;
    // n = 8 NT exactly (the launcher's condition for this kernel): compile-time, so that the index arithmetic of the pair sums
    // (a division and a remainder by h or h + 1 per term) and every LDS address fold into constants -- with n read from the
    // descriptor the pair sums cost 320 cycles per term and 18 % of the kernel
    constexpr int n = 8 * NT, n2 = 2 * n, nks = n2 / 4, h = n / 2;
    const int P = S.P;
    T* Y = reinterpret_cast<T*>(smem_raw);               // [h][2n][16]
    Cx<T>* red = reinterpret_cast<Cx<T>*>(Y + (size_t)h * n2 * 16);   // [NTHR]
    const T* Iw = MINV + (size_t)w * minv_stride + minv_off + (size_t)kdet * n * n * 2;
    const T* Mw = MOUT + (size_t)w * mout_stride + mout_off + (size_t)kdet * n * n * 2 * P;
    T* Tw = TR + (size_t)w * tr_stride + tr_off + (size_t)kdet * 2 * P;
    T af[NT][KSMAX];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KSMAX; ++ks) {
            const int np = 16 * nt + lr, kp = 4 * ks + lq;
            T v = 0;
            if (np < n2 && ks < nks) {
                const int e = np >> 1, rip = np & 1, m = kp >> 1, ri = kp & 1;
                const T re = Iw[(m * n + e) * 2], im = Iw[(m * n + e) * 2 + 1];
                v = (ri == rip) ? re : (rip == 0 ? -im : im);
            }
            af[nt][ks] = v;
        }
    if (stamp) c_setup = clock64() - c_begin;
    Cx<T> y2(0, 0);
    const int d = tid & 15, g = tid >> 4;
    // The wave's rows over all slot tiles and passes form one sequence q = ((tile, pass), row): the operands of row q + PF are
    // requested before the MFMAs of row q, across pass and tile boundaries (one workgroup per CU: nothing else hides the latency).
    const int R = wave < h ? (h - wave + NW - 1) / NW : 0, total = R * 3 * (P / 16);
    T bc[KSMAX], bn[KSMAX], bm[KSMAX];
    auto load_q = [&](T (&b)[KSMAX], int q) {
        if (q < total) {
            const int tp = q / R, ii = wave + NW * (q - tp * R), st = tp / 3, i = (tp - 3 * st == 0 ? 0 : h) + ii;
            const T* xp = Mw + ((size_t)st * n * n2 + (size_t)i * n2 + lq) * 16 + lr;
#pragma unroll
            for (int ks = 0; ks < KSMAX; ++ks) b[ks] = ks < nks ? xp[(size_t)(4 * ks) * 16] : T(0);
        }
    };
    load_q(bc, 0);
    if (PF == 2) load_q(bn, 1);
    int q = 0;
    for (int st = 0; st < P / 16; ++st) {
        const int slot = 16 * st + d;
        const bool live = slot >= 1 && slot < S.D;
        Cx<T> trc(0, 0);
        for (int pass = 0; pass < 3; ++pass) {
            // products of this pass: rows i0..i0+h-1, accumulator tiles [NT0, NT1) (compile-time), stored at row (i - i0) of Y
            auto products = [&](auto t0, auto t1) {
                constexpr int NT0 = decltype(t0)::value, NT1 = decltype(t1)::value, NTP = NT1 - NT0;
                for (int ii = wave; ii < h; ii += NW, ++q) {
                    if (PF == 2) load_q(bm, q + 2); else load_q(bn, q + 1);
                    acc_t acc[NTP];
#pragma unroll
                    for (int nt = 0; nt < NTP; ++nt) acc[nt] = acc_t{0, 0, 0, 0};
#pragma unroll
                    for (int ks = 0; ks < KSMAX; ++ks) {
                        if (ks < nks) {
#pragma unroll
                            for (int nt = 0; nt < NTP; ++nt) acc[nt] = mfma16(af[NT0 + nt][ks], bc[ks], acc[nt]);
                        }
                    }
#pragma unroll
                    for (int ks = 0; ks < KSMAX; ++ks) { bc[ks] = bn[ks]; if (PF == 2) bn[ks] = bm[ks]; }
#pragma unroll
                    for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) Y[((size_t)ii * n2 + 16 * (NT0 + nt) + acc_row<T>(lane, r)) * 16 + lr] = acc[nt][r];
                }
            };
            // (the launcher guarantees 2 * nth == NT: n = 8 NT)
            if (stamp) c_t = clock64();
            if (pass == 0) products(std::integral_constant<int, 0>(), std::integral_constant<int, NT>());
            else if (pass == 1) products(std::integral_constant<int, 0>(), std::integral_constant<int, NT / 2>());
            else products(std::integral_constant<int, NT / 2>(), std::integral_constant<int, NT>());
            if (stamp) { const long long c = clock64(); c_prod += c - c_t; c_t = c; }
            __syncthreads();
            if (stamp) { const long long c = clock64(); c_bar1 += c - c_t; c_t = c; }
            // pairs of this pass.  Y row index = electron - (h if the electron is in I2); the column keeps its global index.
            // A lane takes one (i, e) term for FOUR consecutive slots (16-byte LDS reads): a quarter of the iterations of the
            // one-slot-per-lane form, which was bound by the latency of its reads with one wave per SIMD (280 k of 1.89 M cycles
            // per workgroup).  Slots 0 and 1 (value, Laplacian) do not enter the sum; padding slots hold zeros.
            {
                typedef T vec4 __attribute__((ext_vector_type(4), aligned(16)));
                const int d4 = tid & 3, g4 = tid >> 2;
                constexpr int NG4 = NTHR / 4;
                const bool skip01 = st == 0 && d4 == 0;
                auto ld4 = [&](int row, int col) { return *reinterpret_cast<const vec4*>(Y + ((size_t)row * n2 + col) * 16 + 4 * d4); };
                auto term = [&](int ri, int ci, int re_, int ce, T wgt) {      // w * Y[ri][ci] * Y[re_][ce], complex, four slots
                    const vec4 ar = ld4(ri, 2 * ci), ai = ld4(ri, 2 * ci + 1), br = ld4(re_, 2 * ce), bi = ld4(re_, 2 * ce + 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (!(skip01 && j < 2)) y2 = cx_fma(wgt * Cx<T>(ar[j], ai[j]), Cx<T>(br[j], bi[j]), y2);
                };
                if (pass == 1) {      // i in I1 (rows still from pass A, columns C2), e in I2 (rows from pass B, columns C1): weight 2
#pragma unroll 3
                    for (int pi = g4; pi < h * h; pi += NG4) {
                        const int i = pi / h, e = h + pi % h;
                        term(i, e, e - h, i, T(2));
                    }
                } else {              // upper triangle of the diagonal block (electrons off .. off + h - 1)
                    const int off = pass == 0 ? 0 : h;
                    for (int r = g; r < h; r += NG)      // the trace of this block, one slot per lane as the reduction below expects
                        trc = trc + Cx<T>(Y[((size_t)r * n2 + 2 * (off + r)) * 16 + d], Y[((size_t)r * n2 + 2 * (off + r) + 1) * 16 + d]);
                    // rows r and h-1-r of the triangle hold h+1 entries together: a rectangle (h/2) x (h+1), no search
#pragma unroll 3
                    for (int pi = g4; pi < (h / 2) * (h + 1); pi += NG4) {
                        const int rr = pi / (h + 1), tt = pi - rr * (h + 1);
                        const int r = tt < h - rr ? rr : h - 1 - rr, t = tt < h - rr ? tt : tt - (h - rr);
                        const int i = off + r, e = off + r + t;
                        term(r, e, e - off, i, i != e ? T(2) : T(1));
                    }
                }
            }
            if (stamp) { const long long c = clock64(); c_pairs += c - c_t; c_t = c; }
            __syncthreads();
            if (stamp) { const long long c = clock64(); c_bar2 += c - c_t; c_t = c; }
        }
        red[tid] = trc;
        __syncthreads();
        if (g == 0 && live) {
            Cx<T> t(0, 0);
            for (int u = 0; u < NG; ++u) t = t + red[u * 16 + d];
            Tw[slot] = t.re;
            Tw[P + slot] = t.im;
        }
        __syncthreads();
        if (stamp) { const long long c = clock64(); c_red += c - c_t; c_t = c; }
    }
    if (stamp && lane == 0) {
        unsigned long long* o = tl + wave * 8;
        o[0] = c_setup; o[1] = c_prod; o[2] = c_bar1; o[3] = c_pairs; o[4] = c_bar2; o[5] = c_red; o[6] = clock64() - c_begin; o[7] = 0;
    }
    red[tid] = y2;
    __syncthreads();
    if (tid == 0) {
        Cx<T> t(0, 0);
        for (int u = 0; u < NTHR; ++u) t = t + red[u];
        T* dw = DETS + (size_t)w * dets_stride + dets_off + (size_t)kdet * 4;
        dw[2] = t.re;
        dw[3] = t.im;
    }
}

// =====================================================================================
// 5b-3. the same for ANY matrix size 16 < n <= 64 (odd sizes, and sizes whose slot tile of Y fits the LDS in neither of the modes
//     above: n > 34 in float64): the electrons are cut into blocks of 16, and for every pair of blocks (a <= b) the two blocks
//     Y[I_a][C_b] and Y[I_b][C_a] of one slot tile sit in the LDS together -- all that the pair sums of that block pair need.
//     Nothing is computed twice; the rows of MOUT are read once per column block (n / 16 times, from the L2 after the first).
//       loop order: block pairs outside (their two sets of A fragments stay in registers: four waves per workgroup, one per SIMD,
//       for the 512-register budget), slot tiles inside; tr Y_d is collected per slot in the LDS by one owner thread per slot.
// =====================================================================================
template <typename T, int KSM>      // KSM >= ceil(2 n / 4) k-steps
__global__ void __launch_bounds__(256) k_det_trace_blocked(SysDev<T> S, const T* __restrict__ MOUT, size_t mout_stride, size_t mout_off,
                                                           int ch, const T* __restrict__ MINV, size_t minv_stride, size_t minv_off,
                                                           T* __restrict__ TR, size_t tr_stride, size_t tr_off,
                                                           T* __restrict__ DETS, size_t dets_stride, size_t dets_off) {
    typedef typename Acc4<T>::type acc_t;
    constexpr int NTB = 2;                                 // a block of 16 electrons = 32 real columns = two accumulator tiles
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int kdet = blockIdx.x, w = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
    const int P = S.P, n = S.det_n[ch], n2 = 2 * n, nks = (n2 + 3) / 4, R = (n + 15) / 16, ntile = P / 16;
    T* Y0 = reinterpret_cast<T*>(smem_raw);               // [16 rows][32 columns][16 slots]: rows of I_a, columns of C_b
    T* Y1 = Y0 + 16 * 32 * 16;                            //                                  rows of I_b, columns of C_a
    Cx<T>* trl = reinterpret_cast<Cx<T>*>(Y1 + 16 * 32 * 16);     // [P] tr Y_d
    Cx<T>* red = trl + P;                                 // [256]
    const T* Iw = MINV + (size_t)w * minv_stride + minv_off + (size_t)kdet * n * n * 2;
    const T* Mw = MOUT + (size_t)w * mout_stride + mout_off + (size_t)kdet * n * n * 2 * P;
    for (int p = tid; p < P; p += 256) trl[p] = Cx<T>(0, 0);
    // A fragments of a column block: rows (e, ri2) of the real expansion of Minv^T, columns (m, ri); zero beyond the matrix
    auto load_A = [&](T (&af)[NTB][KSM], int b) {
#pragma unroll
        for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
            for (int ks = 0; ks < KSM; ++ks) {
                const int np = 32 * b + 16 * nt + lr, kp = 4 * ks + lq;
                const int e = np >> 1, rip = np & 1, m = kp >> 1, ri = kp & 1;
                T v = 0;
                if (e < n && m < n) {
                    v = Iw[(m * n + e) * 2 + (ri != rip ? 1 : 0)];
                    if (ri != rip && rip == 0) v = -v;
                }
                af[nt][ks] = v;
            }
    };
    // Y[I_a][C_.] of slot tile st with the A fragments of the column block: a wave takes rows wave, wave + 4, ...; the operands of
    // its next row are requested before the products of the current one
    auto block = [&](const T (&af)[NTB][KSM], int a, int st, T* Y) {
        T bx[2][KSM];
        auto load_row = [&](int u, int r) {
            const int i = 16 * a + r;
            const T* xp = Mw + ((size_t)st * n * n2 + (size_t)(i < n ? i : n - 1) * n2 + lq) * 16 + lr;
#pragma unroll
            for (int ks = 0; ks < KSM; ++ks) bx[u][ks] = (ks < nks && 4 * ks + lq < n2) ? xp[(size_t)(4 * ks) * 16] : T(0);
        };
        load_row(0, wave);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wave + 4 * j;
            if (j + 1 < 4) load_row((j + 1) & 1, r + 4);
            if (16 * a + r < n) {
                acc_t acc[NTB];
#pragma unroll
                for (int nt = 0; nt < NTB; ++nt) acc[nt] = acc_t{0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KSM; ++ks)
                    if (ks < nks) {
#pragma unroll
                        for (int nt = 0; nt < NTB; ++nt) acc[nt] = mfma16(af[nt][ks], bx[j & 1][ks], acc[nt]);
                    }
#pragma unroll
                for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) Y[(r * 32 + 16 * nt + acc_row<T>(lane, q)) * 16 + lr] = acc[nt][q];
            }
        }
    };
    Cx<T> y2(0, 0);
    const int d = tid & 15, g = tid >> 4;                 // pair sums: 16 thread groups x 16 slots
    for (int a = 0; a < R; ++a)
        for (int b = a; b < R; ++b) {
            T afa[NTB][KSM], afb[NTB][KSM];
            load_A(afb, b);
            if (a != b) load_A(afa, a);
            const T wgt = a == b ? T(1) : T(2);
            for (int st = 0; st < ntile; ++st) {
                block(afb, a, st, Y0);
                if (a != b) block(afa, b, st, Y1);
                __syncthreads();
                const T* Yt = a == b ? Y0 : Y1;
                const int slot = 16 * st + d;
                if (slot >= 2 && slot < S.D) {
                    for (int t = g; t < 256; t += 16) {
                        const int r = t >> 4, c = t & 15;
                        if (16 * a + r < n && 16 * b + c < n) {
                            const Cx<T> yrc(Y0[(r * 32 + 2 * c) * 16 + d], Y0[(r * 32 + 2 * c + 1) * 16 + d]);
                            const Cx<T> ycr(Yt[(c * 32 + 2 * r) * 16 + d], Yt[(c * 32 + 2 * r + 1) * 16 + d]);
                            y2 = cx_fma(wgt * yrc, ycr, y2);
                        }
                    }
                }
                if (a == b && g == 0 && slot >= 1 && slot < S.D) {      // (one owner thread per slot: no atomics, a fixed order)
                    Cx<T> t = trl[slot];
                    for (int r = 0; r < 16 && 16 * a + r < n; ++r) t = t + Cx<T>(Y0[(r * 32 + 2 * r) * 16 + d], Y0[(r * 32 + 2 * r + 1) * 16 + d]);
                    trl[slot] = t;
                }
                __syncthreads();
            }
        }
    T* Tw = TR + (size_t)w * tr_stride + tr_off + (size_t)kdet * 2 * P;
    for (int p = tid; p < S.D; p += 256)
        if (p >= 1) { Tw[p] = trl[p].re; Tw[P + p] = trl[p].im; }
    red[tid] = y2;
    __syncthreads();
    if (tid == 0) {
        Cx<T> t(0, 0);
        for (int u = 0; u < 256; ++u) t = t + red[u];
        T* dw = DETS + (size_t)w * dets_stride + dets_off + (size_t)kdet * 4;
        dw[2] = t.re;
        dw[3] = t.im;
    }
}

// =====================================================================================
// 5c. combine determinants (network.py:395-427 log-sum-exp) and assemble
//        E_kin = -1/2 sum_k w_k [ lap log D_k + sum_d (d_d log D_k)^2 ],  w_k = D_k / sum D
//     (equals the -1/2 sum_d [d_d^2 f + (d_d f)^2] of hamiltonian.py:59-68)
// =====================================================================================
template <typename T>
__global__ void __launch_bounds__(64) k_combine(SysDev<T> S, const T* __restrict__ TR, size_t tr_stride, size_t tr_off1,
                                                const T* __restrict__ DETS, size_t dets_stride, size_t dets_off1,
                                                T* __restrict__ out_ke, T* __restrict__ out_logabs, T* __restrict__ out_phase,
                                                T* __restrict__ out_grad) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const int P = S.P, K = S.K;
    const T* Tw = TR + (size_t)w * tr_stride;
    const T* Dw = DETS + (size_t)w * dets_stride;
    // log D_k
    T la[DS_MAX_DETS], ar[DS_MAX_DETS];
    T mx = -1e300;
    for (int k = 0; k < K; ++k) {
        la[k] = Dw[4 * k]; ar[k] = Dw[4 * k + 1];
        if (S.n_detch > 1) { la[k] += Dw[dets_off1 + 4 * k]; ar[k] += Dw[dets_off1 + 4 * k + 1]; }
        mx = la[k] > mx ? la[k] : mx;
    }
    Cx<T> sum(0, 0);
    Cx<T> wk[DS_MAX_DETS];
    for (int k = 0; k < K; ++k) {
        T sn, cs;
        ds_sincos(ar[k], &sn, &cs);
        const T e = ds_exp(la[k] - mx);
        wk[k] = Cx<T>(e * cs, e * sn);
        sum = sum + wk[k];
    }
    const Cx<T> sinv = cx_inv(sum);
    Cx<T> ke(0, 0);
    for (int k = 0; k < K && out_ke; ++k) {
        // sum_d (d_d log D_k)^2 over direction slots, complex square
        Cx<T> g2(0, 0);
        for (int d = 2 + lane; d < S.D; d += 64) {
            Cx<T> g(Tw[(size_t)(k * 2) * P + d], Tw[(size_t)(k * 2 + 1) * P + d]);
            if (S.n_detch > 1) g = g + Cx<T>(Tw[tr_off1 + (size_t)(k * 2) * P + d], Tw[tr_off1 + (size_t)(k * 2 + 1) * P + d]);
            g2 = g2 + g * g;
        }
        g2.re = wave_sum(g2.re); g2.im = wave_sum(g2.im);
        Cx<T> lap(Tw[(size_t)(k * 2) * P + 1] - Dw[4 * k + 2], Tw[(size_t)(k * 2 + 1) * P + 1] - Dw[4 * k + 3]);
        if (S.n_detch > 1)
            lap = lap + Cx<T>(Tw[tr_off1 + (size_t)(k * 2) * P + 1] - Dw[dets_off1 + 4 * k + 2],
                              Tw[tr_off1 + (size_t)(k * 2 + 1) * P + 1] - Dw[dets_off1 + 4 * k + 3]);
        ke = ke + (wk[k] * sinv) * (lap + g2);
    }
    if (out_grad) {   // d log psi / d x_d = sum_k w_k d_d log D_k  (complex: Re = grad log|psi|, Im = grad arg psi)
        for (int d = 2 + lane; d < S.D; d += 64) {
            Cx<T> acc(0, 0);
            for (int k = 0; k < K; ++k) {
                Cx<T> g(Tw[(size_t)(k * 2) * P + d], Tw[(size_t)(k * 2 + 1) * P + d]);
                if (S.n_detch > 1) g = g + Cx<T>(Tw[tr_off1 + (size_t)(k * 2) * P + d], Tw[tr_off1 + (size_t)(k * 2 + 1) * P + d]);
                acc = acc + (wk[k] * sinv) * g;
            }
            out_grad[((size_t)w * (S.D - 2) + (d - 2)) * 2] = acc.re;
            out_grad[((size_t)w * (S.D - 2) + (d - 2)) * 2 + 1] = acc.im;
        }
    }
    if (lane == 0) {
        if (out_ke) { out_ke[2 * w] = T(-0.5) * ke.re; out_ke[2 * w + 1] = T(-0.5) * ke.im; }
        const T as = ds_sqrt(cx_abs2(sum));
        if (out_logabs) out_logabs[w] = ds_log(as) + mx;
        if (out_phase) { out_phase[2 * w] = sum.re / as; out_phase[2 * w + 1] = sum.im / as; }
    }
}

// The log-sum-exp alone (value chain: log|psi| and phase from the log-determinants), one THREAD per walker and no private arrays
// (k_combine keeps 2 x DS_MAX_DETS numbers per lane for the energy terms and runs 64 lanes per walker: 53 us per 4096-walker
// forward for sixteen logarithms per walker).  Same operations in the same order as k_combine: bit-identical log|psi| / phase.
template <typename T>
__global__ void __launch_bounds__(256) k_combine_val(SysDev<T> S, const T* __restrict__ DETS, size_t dets_stride, size_t dets_off1, long B,
                                                     T* __restrict__ out_logabs, T* __restrict__ out_phase) {
    const long w = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= B) return;
    const int K = S.K;
    const T* Dw = DETS + (size_t)w * dets_stride;
    const bool two = S.n_detch > 1;
    T mx = -1e300;
    for (int k = 0; k < K; ++k) {
        T la = Dw[4 * k];
        if (two) la += Dw[dets_off1 + 4 * k];
        mx = la > mx ? la : mx;
    }
    Cx<T> sum(0, 0);
    for (int k = 0; k < K; ++k) {
        T la = Dw[4 * k], ar = Dw[4 * k + 1];
        if (two) { la += Dw[dets_off1 + 4 * k]; ar += Dw[dets_off1 + 4 * k + 1]; }
        T sn, cs;
        ds_sincos(ar, &sn, &cs);
        const T e = ds_exp(la - mx);
        sum = sum + Cx<T>(e * cs, e * sn);
    }
    const T as = ds_sqrt(cx_abs2(sum));
    if (out_logabs) out_logabs[w] = ds_log(as) + mx;
    if (out_phase) { out_phase[2 * w] = sum.re / as; out_phase[2 * w + 1] = sum.im / as; }
}

// =====================================================================================
// 6. Ewald energy per walker (ewaldsum.py:138-191) with the minimal-image dispatch of distance.py:32-141
// =====================================================================================
template <typename T>
__device__ __forceinline__ void min_image(const SysDev<T>& S, const T d[3], T out[3]) {
    if (S.dist_mode == 0) {           // diagonal_dist_i  distance.py:110-128
        for (int c = 0; c < 3; ++c) {
            const T Lc = S.sim_a[4 * c];
            out[c] = ds_pymod(d[c] + Lc / 2, Lc) - Lc / 2;
        }
    } else if (S.dist_mode == 1) {    // orthogonal_dist_i  distance.py:91-108
        T fr[3];
        for (int c = 0; c < 3; ++c) {
            const T f = d[0] * S.sim_ainv[c] + d[1] * S.sim_ainv[3 + c] + d[2] * S.sim_ainv[6 + c];
            fr[c] = ds_pymod(f + T(0.5), T(1)) - T(0.5);
        }
        for (int c = 0; c < 3; ++c) out[c] = fr[0] * S.sim_a[c] + fr[1] * S.sim_a[3 + c] + fr[2] * S.sim_a[6 + c];
    } else {                          // general_dist_i  distance.py:70-89 (first minimum on ties)
        T best = 0;
        for (int s = 0; s < 27; ++s) {
            const T v0 = d[0] + S.shift27[3 * s], v1 = d[1] + S.shift27[3 * s + 1], v2 = d[2] + S.shift27[3 * s + 2];
            const T nn = ds_sqrt(v0 * v0 + v1 * v1 + v2 * v2);
            if (s == 0 || nn < best) { best = nn; out[0] = v0; out[1] = v1; out[2] = v2; }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) k_ewald(SysDev<T> S, const T* __restrict__ x, T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* xs = reinterpret_cast<T*>(smem_raw);   // [N][3]
    T* red = xs + 3 * S.N;                    // [2][256]
    const int w = blockIdx.x, tid = threadIdx.x, N = S.N;
    for (int idx = tid; idx < 3 * N; idx += 256) xs[idx] = x[(size_t)w * 3 * N + idx];
    __syncthreads();
    T ee = 0, ei = 0;
    // real space: electron-ion pairs then electron-electron pairs i<j, 27 lattice images each
    const int n_ei = N * S.As, n_ee = N * (N - 1) / 2;
    for (int it = tid; it < n_ei + n_ee; it += 256) {
        T d[3], mi[3];
        T q;
        if (it < n_ei) {
            const int i = it / S.As, a = it % S.As;
            for (int c = 0; c < 3; ++c) d[c] = xs[3 * i + c] - S.sim_atoms[3 * a + c];
            q = -S.sim_charges[a];
        } else {
            // unrank (i<j)
            int r = it - n_ei, i = 0;
            while (r >= N - 1 - i) { r -= N - 1 - i; ++i; }
            const int j = i + 1 + r;
            for (int c = 0; c < 3; ++c) d[c] = xs[3 * i + c] - xs[3 * j + c];
            q = 1;
        }
        min_image(S, d, mi);
        T acc = 0;
        for (int s = 0; s < 27; ++s) {
            const T v0 = mi[0] + S.disp27[3 * s], v1 = mi[1] + S.disp27[3 * s + 1], v2 = mi[2] + S.disp27[3 * s + 2];
            const T r = ds_sqrt(v0 * v0 + v1 * v1 + v2 * v2);
            acc += ds_erfc(S.alpha * r) / r;
        }
        if (it < n_ei) ei += q * acc; else ee += acc;
    }
    // reciprocal space (ewaldsum.py:176-182): sum_G w_G |sum_e e^{iG.r_e}|^2 and the electron-ion term
    if (S.g_len > 0) {
        // every G is an integer combination of the reciprocal vectors, so e^{iG.r} = E_1[n1] E_2[n2] E_3[n3] with
        // E_j[n] = exp(i n theta_j), theta_j = 2 pi frac_j(r): N * g_len sincos per walker instead of N * NG
        T* tab = red + 512;                      // [N][g_len][re, im]
        const int L = S.g_len;
        for (int idx = tid; idx < N * L; idx += 256) {
            const int i = idx / L, t = idx - i * L;
            const int j = t >= S.g_off[2] ? 2 : (t >= S.g_off[1] ? 1 : 0);
            const T nn = (T)(S.g_nmin[j] + (t - S.g_off[j]));
            const T frac = xs[3 * i] * S.sim_ainv[j] + xs[3 * i + 1] * S.sim_ainv[3 + j] + xs[3 * i + 2] * S.sim_ainv[6 + j];
            T sn, cs;
            ds_sincos(nn * (T(6.283185307179586476925286766559) * frac), &sn, &cs);
            tab[2 * idx] = cs; tab[2 * idx + 1] = sn;
        }
        __syncthreads();
        for (int g = tid; g < S.NG; g += 256) {
            const int t0 = (int)S.gidx[3 * g], t1 = (int)S.gidx[3 * g + 1], t2 = (int)S.gidx[3 * g + 2];
            T ss = 0, sc = 0;
            for (int i = 0; i < N; ++i) {
                const T* ti = tab + 2 * (size_t)i * L;
                const T ar = ti[2 * t0], ai = ti[2 * t0 + 1], br = ti[2 * t1], bi = ti[2 * t1 + 1], cr = ti[2 * t2], ci = ti[2 * t2 + 1];
                const T pr = ar * br - ai * bi, pi = ar * bi + ai * br;
                sc += pr * cr - pi * ci;
                ss += pr * ci + pi * cr;
            }
            const T wg = S.gweight[g];
            ee += wg * (ss * ss + sc * sc);
            ei += 2 * wg * (-S.ion_re[g] * sc - S.ion_im[g] * ss);
        }
    } else
    for (int g = tid; g < S.NG; g += 256) {
        const T g0 = S.gpoints[3 * g], g1 = S.gpoints[3 * g + 1], g2 = S.gpoints[3 * g + 2];
        T ss = 0, sc = 0;
        for (int i = 0; i < N; ++i) {
            T sn, cs;
            ds_sincos(xs[3 * i] * g0 + xs[3 * i + 1] * g1 + xs[3 * i + 2] * g2, &sn, &cs);
            ss += sn; sc += cs;
        }
        const T wg = S.gweight[g];
        ee += wg * (ss * ss + sc * sc);
        ei += 2 * wg * (-S.ion_re[g] * sc - S.ion_im[g] * ss);
    }
    red[tid] = ee; red[256 + tid] = ei;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) { red[tid] += red[tid + off]; red[256 + tid] += red[256 + tid + off]; }
        __syncthreads();
    }
    if (tid == 0) {
        out[3 * w] = red[0] + S.ee_const;
        out[3 * w + 1] = red[256] + S.ei_const;
        out[3 * w + 2] = S.ii_total;
    }
}

// =====================================================================================
// 7. walker wrap and Metropolis pieces (distance.py:144-163, qmc.py:190-196 / 217-222)
// =====================================================================================
template <typename T> struct Lattice { T a[9], ainv[9]; };

template <typename T>
__global__ void k_enforce_pbc(Lattice<T> lat, const T* __restrict__ x, size_t n_elec, T* __restrict__ out,
                              T* __restrict__ wrap_out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elec) return;
    T r[3] = {x[3 * e], x[3 * e + 1], x[3 * e + 2]}, o[3], wr[3];
    wrap_point(r, lat.a, lat.ainv, o, wr);
    for (int c = 0; c < 3; ++c) out[3 * e + c] = o[c];
    if (wrap_out)
        for (int c = 0; c < 3; ++c) wrap_out[3 * e + c] = wr[c];
}

template <typename T>
__global__ void k_mh_propose(const T* __restrict__ a, const T* __restrict__ ainv, const T* __restrict__ x1,
                             const T* __restrict__ normal, T width, size_t n_elec, T* __restrict__ x2) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elec) return;
    T r[3], o[3], wr[3];
    for (int c = 0; c < 3; ++c) r[c] = x1[3 * e + c] + width * normal[3 * e + c];
    wrap_point(r, a, ainv, o, wr);
    for (int c = 0; c < 3; ++c) x2[3 * e + c] = o[c];
}

template <typename T>
__global__ void k_mh_accept(T* __restrict__ x1, T* __restrict__ lp1, const T* __restrict__ x2, const T* __restrict__ lp2,
                            const T* __restrict__ uniform, int n3, T* __restrict__ n_accept) {
    // one block per walker
    const int w = blockIdx.x;
    const bool cond = (lp2[w] - lp1[w]) > ds_log(uniform[w]);
    if (cond)
        for (int c = threadIdx.x; c < n3; c += blockDim.x) x1[(size_t)w * n3 + c] = x2[(size_t)w * n3 + c];
    __syncthreads();
    if (threadIdx.x == 0 && cond) {
        lp1[w] = lp2[w];
        atomicAdd(n_accept, T(1));
    }
}

// out[w] = t[3w] + t[3w+1] + t[3w+2]   (hamiltonian.py:175-177 sums the Ewald triple)
template <typename T> __global__ void k_sum3(const T* __restrict__ t, long n, T* __restrict__ out) {
    const long w = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n) out[w] = t[3 * w] + t[3 * w + 1] + t[3 * w + 2];
}

// HBM counter calibration: a plain 8-byte-per-lane copy (the access width of every jet tensor load / store)
static __global__ void k_calib_copy(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// fp64 MFMA issue-rate probe: NACC independent accumulators per wave.  The (256, 2) launch bound makes the compiler keep
// the accumulators in VGPRs: on gfx950 v_mfma_f64_16x16x4_f64 issues every 64 cycles with VGPR accumulators but only every
// ~97 cycles when they live in AGPRs (tools/probes/mfma16_nacc.hip vs mfma44_probe.hip, profiles/r01_mfma44_probe.txt).
template <int NACC>
__global__ void __launch_bounds__(256, 2) k_mfma_peak(long iters, double* out) {
    typedef Acc4<double>::type acc_t;
    acc_t acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = acc_t{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    // shader-clock cycles (s_memtime) and constant-rate ticks (s_memrealtime) across the loop: their ratio is
    // the core clock the MFMA pipe actually ran at
    const long long c0 = clock64(), r0 = wall_clock64();
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = mfma16(a, b, acc[j]);
    }
    double s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    const long long c1 = clock64(), r1 = wall_clock64();
    if (s == 12345.678) out[0] = s;   // keep the chain alive
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) { out[1] = (double)(c1 - c0); out[2] = (double)(r1 - r0); }
}

}  // namespace ds
