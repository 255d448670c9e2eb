// ds_layer.h -- one-electron-stream layer of the forward-Laplacian chain, "electron group" formulation (round 3).
//
//   k_layer_unit    z_i = W_loc^T [h_i ; mean_j h2_ji] + S  ->  tanh chain rule on the jets  ->  residual  ->  G_out
//                   for ONE 16-slot tile of a GROUP of up to LG_GE electrons of one spin per workgroup (network.py:305-332, 521-528).
//   k_layer_fin     the Laplacian slot of the layer output (needs the sum of squares over ALL slot tiles of an electron).
//   k_m2_means      the partner sums of the pair stream as 5-jets (value, d/dr, Laplacian) per (electron, spin, feature).
//   k_group_fold    partial spin means of several groups -> one mean per spin.
//
// Why groups.  A workgroup of k_jet_gemm owns ONE electron, so the spin means of a layer's output (the next layer's shared
// term) needed a second pass over all of G (k_shared_term: 4 GB per 1024-walker launch at 24 electrons), and the pair-mean
// rows of the layer input had to be expanded to dense jet rows in HBM first (k_m2_expand: 1.4 GB per launch).  Here the
// four waves of a workgroup (64 output features each) walk the group's electrons four at a time for one slot tile: a "pass"
// is one 16-slot tile of four electrons (4 x 4 accumulator tiles), and
//   * the sum over the group's electrons of the output tile is a lane-local sum over accumulator tiles, carried across the
//     passes and written once: MEANP[walker][group][n][slot] (already divided by the spin's electron count) -- the input of
//     the next layer's shared term, no second pass over G;
//   * the rows k >= Kh of the jet operand (pair means) are generated in the operand load from the pair stream's 5-jets:
//     slot (j, c) of row (spin s, k2) of electron i is  H2[k2][1+c][i*N+j] / n_s  (j in s, j != i),  -mean_c (j == i),
//     the mean value / Laplacian for slots 0 / 1, zero otherwise (network.py:323-328; d/dx_j = +d/dr, d/dx_i = -d/dr);
//   * what ties the slot tiles of an electron together goes through three small per-electron arrays instead of a second
//     pass: y = tanh(z_0) (written by the launch of slot tile 0, read by the launch of the other tiles), the partial sums
//     of squares per tile, and y' z_L; k_layer_fin assembles the Laplacian slot  y' z_L + y'' sum_d z_d^2  from them.
#pragma once
#include "ds_gemm.h"

namespace ds {

constexpr int LG_GE = 12;   // electrons per group (three passes of four per slot tile)

template <typename T> struct LayerArgs {
    const T* Gin;          // [walker][electron][ldk rows][P]   rows 0..Kh-1: h_i
    T* Gout;               // same geometry, rows 0..Nout-1 written
    size_t g_ws, g_ts;     // walker / electron stride of G (elements)
    const T* W;            // W_loc [(Kh + nch*K2)][Nout]
    int Kh, K2, Nout;
    const T* Sb;           // [walker][Nout][P]   shared spin-mean term + bias
    const T* H2;           // [walker][K2][5][NP] pair stream
    size_t h2_ws;
    const T* M2V;          // [walker][N][nch][K2][5] signed partner SUMS of the pair stream (k_m2_means)
    size_t m2v_ws;
    T* MEANP;              // [walker][group][Nout][P] partial spin means of the output
    size_t mp_ws;
    // per-electron quantities that tie the slot tiles of an electron together (lay_ws = walker stride of each, elements):
    T* Y;                  // [walker][electron][Nout]         y = tanh(z_0): written by the slot-tile-0 launch, read by the others
    T* ZLD;                // [walker][electron][Nout]         y' z_L (the linear part of the Laplacian slot), from slot tile 0
    T* SSP;                // [walker][slot tile][electron][Nout]  sum over the tile's gradient slots of z_d^2
    size_t lay_ws;
    int t0, nt;            // slot tiles [t0, t0 + nt) are handled by this launch
    const T* zero;         // one element holding 0 (the B operand of lanes that contribute nothing in a pair sub-phase)
    unsigned long long* clk;   // optional in-kernel clock probe (ds_profile_*): clk[0] += shader cycles, clk[1] += 100 MHz ticks per workgroup
    int dbg;               // timing experiments only (DS_LG_DBG)
};
template <typename T> inline size_t layer_unit_lds_bytes(unsigned threads) { return (size_t)(threads / 64) * 2 * 16 * 64 * sizeof(T); }

// M2V[w][e][sp][k2][comp] = sgn(comp) sum_{j in sp} H2[w][k2][comp][e*N + j]  with sgn = -1 for the gradient components 1..3
// (d/dx_e of h2[j][e](x_j - x_e) = -d/dr) and +1 for value / Laplacian (comp 4 already holds the full Laplacian).  SUMS, not
// means: the factor 1 / n_sp is common to a whole B-operand row of the layer product and rides on the weight operand there.
// grid (N, walkers), block 256: a 16-lane row owns one (k2, comp) line of the pair stream.
template <typename T>
__global__ void __launch_bounds__(256) k_m2_means(SysDev<T> S, const T* __restrict__ H2, int K2, T* __restrict__ M2V, size_t m2v_ws) {
    const int e = blockIdx.x, w = blockIdx.y, row = threadIdx.x >> 4, lr = threadIdx.x & 15;
    const int N = S.N, NP = S.NP, nch = S.nch;
    const T* Hw = H2 + (size_t)w * K2 * 5 * NP + (size_t)e * N;
    T* out = M2V + (size_t)w * m2v_ws + (size_t)e * nch * K2 * 5;
    for (int item = row; item < K2 * 5; item += 16) {
        const T* hp = Hw + (size_t)item * NP;
        T su = 0, sd = 0;
        for (int j = lr; j < N; j += 16) {
            const T v = hp[j];
            if (j < S.n_up) su += v; else sd += v;
        }
        su = row16_sum(su);
        sd = row16_sum(sd);
        if (lr == 0) {
            const int comp = item % 5;
            const T sg = (comp >= 1 && comp <= 3) ? T(-1) : T(1);
            out[item] = sg * su;
            if (nch > 1) out[K2 * 5 + item] = sg * sd;
        }
    }
}

// MEAN[w][sp][n][slot] = sum over the groups g of spin sp of MEANP[w][g][n][slot]   (groups in index order)
template <typename T>
__global__ void __launch_bounds__(256) k_group_fold(SysDev<T> S, const T* __restrict__ MEANP, size_t mp_ws, int Nout, T* __restrict__ MEAN,
                                                    size_t mean_ws) {
    const int w = blockIdx.y;
    const size_t per = (size_t)Nout * S.P;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < per * S.nch; idx += (size_t)gridDim.x * blockDim.x) {
        const int sp = (int)(idx / per);
        const size_t r = idx - (size_t)sp * per;
        T v = 0;
        for (int g = 0; g < S.n_groups; ++g)
            if (S.grp_sp[g] == sp) v += MEANP[(size_t)w * mp_ws + (size_t)g * per + r];
        MEAN[(size_t)w * mean_ws + idx] = v;
    }
}

// uniform base + 32-bit BYTE offset per lane: the form the backend turns into `global_load v, v_off, s[base]` (an element
// offset would be widened to a 64-bit lane address first)
template <typename V, typename T> __device__ __forceinline__ const V& at_b(const T* base, unsigned byte_off) {
    return *reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <typename V, typename T> __device__ __forceinline__ V& at_b(T* base, unsigned byte_off) {
    return *reinterpret_cast<V*>(reinterpret_cast<char*>(base) + byte_off);
}

// sum over the four 16-lane rows of a wave (lanes l, l + 16, l + 32, l + 48), result in all of them: two gfx950 row swaps
// (v_permlane32_swap: upper half of the first operand <-> lower half of the second; v_permlane16_swap: odd rows <-> even rows)
template <typename T> __device__ __forceinline__ T rows4_sum(T x);
template <> __device__ __forceinline__ double rows4_sum<double>(double x) {
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    auto l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const double y = __hiloint2double((int)h32[0], (int)l32[0]) + __hiloint2double((int)h32[1], (int)l32[1]);
    const unsigned ylo = (unsigned)__double2loint(y), yhi = (unsigned)__double2hiint(y);
    auto l16 = __builtin_amdgcn_permlane16_swap(ylo, ylo, false, false);
    auto h16 = __builtin_amdgcn_permlane16_swap(yhi, yhi, false, false);
    return __hiloint2double((int)h16[0], (int)l16[0]) + __hiloint2double((int)h16[1], (int)l16[1]);
}
template <> __device__ __forceinline__ float rows4_sum<float>(float x) {
    const unsigned u = __float_as_uint(x);
    auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float y = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const unsigned v = __float_as_uint(y);
    auto b = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// The products are formed TRANSPOSED relative to k_jet_gemm:  C^T[slot][n] = sum_k X[k][slot] W[k][n]  (the jet rows are the
// MFMA A operand, the weights the B operand).  An accumulator tile then has the 16 FEATURES n0 + 16a + lr in its lanes and
// the 16 slots of the tile in (lane >> 4, register): everything the tanh chain rule needs per feature -- y' and the sum of
// squares over the slots -- is lane-local or a sum over the four lane rows, instead of sixteen 16-lane reductions per tile.
// The A-operand lane lr loads the slot at tile position pos16(lr), chosen so that accumulator (lane row lq, register r)
// holds tile position own16(lq, r): float32 -- four consecutive slots per lane (one 16-byte access, the four lane rows of a
// feature fill a 64-byte sector); float64 -- two pairs, positions 2lq, 2lq+1 and 8+2lq, 9+2lq, so that EACH of the two
// 16-byte accesses of a lane completes a 64-byte sector of the feature's row together with the other three lane rows
// (pieces of 16 bytes at a stride of 32 would leave every sector half-written per instruction).
template <typename T> __device__ __forceinline__ int own16(int lq, int r) { return sizeof(T) == 8 ? 2 * lq + (r & 1) + 8 * (r >> 1) : 4 * lq + r; }
template <typename T> __device__ __forceinline__ int pos16(int lr) {
    // A-operand row mu = lr lands in accumulator (lq', r') with mu = acc_row(lq' << 4, r')
    return sizeof(T) == 8 ? own16<T>(lr & 3, lr >> 2) : lr;
}
// the lane's four owned values of a 16-slot tile <-> memory: two halves of two elements each
template <typename T> struct Half2 { typedef T type __attribute__((ext_vector_type(2))); };
template <typename T> __device__ __forceinline__ unsigned own_off0(int lq) { return (unsigned)own16<T>(lq, 0) * (unsigned)sizeof(T); }
template <typename T> __device__ __forceinline__ unsigned own_off1(int lq) { return (unsigned)own16<T>(lq, 2) * (unsigned)sizeof(T); }
template <typename T, typename B> __device__ __forceinline__ typename Acc4<T>::type load_own(const B* base, unsigned off, int lq) {
    typedef typename Half2<T>::type h2;
    const h2 lo = at_b<h2>(base, off + own_off0<T>(lq)), hi = at_b<h2>(base, off + own_off1<T>(lq));
    return typename Acc4<T>::type{lo[0], lo[1], hi[0], hi[1]};
}
template <typename T, typename B> __device__ __forceinline__ void store_own(B* base, unsigned off, int lq, typename Acc4<T>::type v, bool nt) {
    typedef typename Half2<T>::type h2;
    const h2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
    if (nt) {
        __builtin_nontemporal_store(lo, &at_b<h2>(base, off + own_off0<T>(lq)));
        __builtin_nontemporal_store(hi, &at_b<h2>(base, off + own_off1<T>(lq)));
    } else {
        at_b<h2>(base, off + own_off0<T>(lq)) = lo;
        at_b<h2>(base, off + own_off1<T>(lq)) = hi;
    }
}

// PIPE = 4 / 2: operand ring of that many k-steps with straight-line phase transitions (needs Kh and K2 multiples of 4 PIPE:
// the hidden layers); PIPE = 0: a plain load / multiply loop (layer 0: K = 4A + nch * 4).
// grid (nt * n_groups * column blocks, walkers): workgroup = (slot tile t0 + ., electron group, 256 features), four waves.
// GATHER: the rows k >= Kh come from the pair stream (above); otherwise all Kh + nch K2 rows are read from G (k_m2_expand ran).
// ST = electrons (column tiles) per pass: 4, or 3 (leaves registers for a four-deep ring and the spin sums).
template <typename T, bool RES, int PIPE, bool GATHER, int ST = 4, int WPS = 2>
__global__ void __launch_bounds__(256, WPS) k_layer_unit(SysDev<T> S, LayerArgs<T> A) {
    typedef typename Acc4<T>::type acc_t;
    constexpr int NB = 4;
    constexpr bool MSUM_REG = ST < 4;                      // spin sums in registers when the accumulators leave room
    // XCD-aware placement as in k_jet_gemm: all workgroups of one walker get linear ids of one residue class mod 8
    int gx = blockIdx.x, w = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const unsigned b = blockIdx.y * gridDim.x + blockIdx.x, q = b >> 3;
        w = (q / gridDim.x) * 8 + (b & 7);
        gx = q % gridDim.x;
    }
    const int gzf = gridDim.x / (S.n_groups * A.nt), zb = gx % gzf, g = (gx / gzf) % S.n_groups, t = A.t0 + gx / (gzf * S.n_groups);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4, n0 = (zb * (blockDim.x >> 6) + wave) * 16 * NB;
    const int Nout = A.Nout;
    if (n0 >= Nout) return;                               // (no barriers below: the LDS state is wave-private)
    long long clk_c0 = 0, clk_r0 = 0;
    if (A.clk && wave == 0) { clk_c0 = clock64(); clk_r0 = wall_clock64(); }
    // (timing experiment, DS_LG_DBG & 32: wave 0 of workgroup 0 writes shader-clock stamps at its phase boundaries to clk[2 + i])
    unsigned long long* tl = (A.clk && (A.dbg & 32) && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0 && lane == 0) ? A.clk + 2 : nullptr;
    int n_tl = 0;
    auto stamp = [&]() {
        if (tl && n_tl < 1000) { __builtin_amdgcn_s_waitcnt(0); tl[n_tl++] = (unsigned long long)clock64(); }
    };
    if (DS_EXP(A.dbg & 24)) {
        // (experiment: start skew of the waves in odd hardware slots (8) or by a hash of the workgroup id (16), A.dbg >> 20 sleeps
        //  of 8128 cycles -- would the two waves of a SIMD overlap better out of lockstep?)
        const unsigned slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11));        // HW_ID.wave_id
        const unsigned b = blockIdx.y * gridDim.x + blockIdx.x, h = (b * 2654435761u) >> 28;
        const unsigned n = (A.dbg & 16) ? h * (unsigned)(A.dbg >> 20) : ((slot & 1u) ? (unsigned)(A.dbg >> 20) : 0u);
        for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
    }
    extern __shared__ __attribute__((aligned(16))) char lg_smem[];
    T* msum = reinterpret_cast<T*>(lg_smem) + (size_t)wave * (2 * 16 * 64);         // [4a + r][lane]: spin sums of the slot tile; behind it the shared term
    const int e0 = S.grp_e0[g], ng = S.grp_n[g], gsp = S.grp_sp[g];
    const int n_up = S.n_up;
    const T inv_ns = T(1) / T(gsp == 0 ? n_up : S.n_dn);
    const unsigned P = (unsigned)S.P;
    const int D = S.D, N = S.N, NP = S.NP, nch = S.nch, K2 = A.K2;
    const int nks1 = DS_EXP(A.dbg & 4) ? 8 : (GATHER ? A.Kh : A.Kh + nch * K2) / 4, nks2 = K2 / 4;      // (experiment: 8 instead of 80 k-steps)
    const T* Gw = A.Gin + (size_t)w * A.g_ws;              // (uniform bases; the lane's place is a 32-bit byte offset)
    T* Go = A.Gout + (size_t)w * A.g_ws;
    const T* Sw = A.Sb + (size_t)w * Nout * P;
    const T* H2w = A.H2 + (size_t)w * A.h2_ws;
    const T* M2w = A.M2V + (size_t)w * A.m2v_ws;
    T* MPw = A.MEANP + (size_t)w * A.mp_ws + (size_t)g * Nout * P;
    T* Yw = A.Y + (size_t)w * A.lay_ws;
    T* ZLw = A.ZLD + (size_t)w * A.lay_ws;
    T* SSw = A.SSP + ((size_t)w * (P / 16) + t) * A.lay_ws;
    const unsigned gts = (unsigned)A.g_ts;
    const int nquad = (ng + ST - 1) / ST;
    const T rs2 = T(0.70710678118654752440);
    const T inv_n0 = T(1) / T(n_up), inv_n1 = T(1) / T(S.n_dn > 0 ? S.n_dn : 1);
    const unsigned PB = P * (unsigned)sizeof(T);                          // row pitch in bytes
    // epilogue: this lane owns features n0 + 16a + lr (a = 0..3) and the slots 16t + own16(lq, r) (r = 0..3)
    const unsigned rowb = (unsigned)(n0 + lr) * PB + (unsigned)(16 * t) * (unsigned)sizeof(T);
    const unsigned featb = (unsigned)(n0 + lr) * (unsigned)sizeof(T);     // (byte offset of feature n0 + lr in the per-electron arrays)
    const int dl = 16 * t + pos16<T>(lr);                  // slot this lane LOADS in every column tile (A-operand row lr)
    acc_t msr[NB];
#pragma unroll
    for (int a = 0; a < NB; ++a) msr[a] = acc_t{0, 0, 0, 0};
    if (!MSUM_REG) {
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) msum[qq * 64 + lane] = T(0);
    }

    // shared term of this slot tile: loaded ONCE into wave-private LDS; every pass starts its accumulators from there (a
    // global load at a pass start would sit behind the previous pass's stores in the in-order memory counter)
    T* sS = msum + 16 * 64;
#pragma unroll
    for (int a = 0; a < NB; ++a) {
        const acc_t v = load_own<T>(Sw, rowb + (unsigned)(16 * a) * PB, lq);
#pragma unroll
        for (int r = 0; r < 4; ++r) sS[(4 * a + r) * 64 + lane] = v[r];
    }
    // operand ring (PIPE k-steps), software-pipelined ACROSS passes: the first PIPE k-steps of the next pass are requested at
    // the start of the current pass's epilogue, before any of its stores
    constexpr int NS = PIPE > 0 ? PIPE : 1;
    T av[NS][NB], bv[NS][ST];
    const T* Wl;
    const T* bp[ST];
    unsigned binc[ST];                                 // element step per k-step of the pair sub-phases
    auto pass_pointers = [&](int q) {
        // (lane ids made opaque: the pointer pieces derived from them are recomputed here -- a few VALU instructions --
        //  instead of being hoisted, spilled, and reloaded in front of the k-loop, which costs a full vmcnt(0) drain inside it)
        int lqp = lq, lrp = lr;
        asm volatile("" : "+v"(lqp), "+v"(lrp));
        Wl = A.W + (size_t)lqp * Nout + n0 + lrp;
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            const int i = e0 + (ST * q + s < ng ? ST * q + s : ng - 1);
            bp[s] = Gw + ((size_t)i * gts + (size_t)lqp * P + dl);
        }
    };
    // (the scheduling barriers pin "MFMAs of a set, then its reload": moved above them, the reloads of a straight-line block
    //  need a second copy of the whole ring)
    auto load_w = [&](int u) {
#pragma unroll
        for (int a = 0; a < NB; ++a) av[u][a] = Wl[16 * a];
        Wl += (size_t)4 * Nout;
    };
    auto load_g = [&](int u) {
        __builtin_amdgcn_sched_barrier(0);
        load_w(u);
#pragma unroll
        for (int s = 0; s < ST; ++s) { bv[u][s] = *bp[s]; bp[s] += (size_t)4 * P; }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_m = [&](int u) {
        __builtin_amdgcn_sched_barrier(0);
        load_w(u);
#pragma unroll
        for (int s = 0; s < ST; ++s) { bv[u][s] = *bp[s]; bp[s] += (size_t)binc[s]; }
        __builtin_amdgcn_sched_barrier(0);
    };
    pass_pointers(0);
    if (PIPE > 0 && !GATHER) {
#pragma unroll
        for (int u = 0; u < NS; ++u) load_g(u);
    }

    for (int q = 0; q < nquad; ++q) {
        const int ecount = ng - ST * q < ST ? ng - ST * q : ST;
        int iel[ST];                                   // electron of column tile s (dead tiles repeat the last one)
#pragma unroll
        for (int s = 0; s < ST; ++s) iel[s] = e0 + (ST * q + s < ng ? ST * q + s : ng - 1);
        // z = W x + (S + b): the accumulators of all four electrons start at the walker's shared term
        stamp();                                       // 0: pass start
        acc_t acc[ST][NB];
#pragma unroll
        for (int a = 0; a < NB; ++a) {
            acc_t v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = sS[(4 * a + r) * 64 + lane];
#pragma unroll
            for (int s = 0; s < ST; ++s) acc[s][a] = v;
        }
        // Pair sub-phase p (partner spin p): jet row (p, k2 = 4 ks + lq) of electron i is, per slot,
        //   slots 0 / 1:              + sum_j value / Laplacian            (M2V, signed sums)
        //   own gradient slots:       - sum_j d/dr                          (M2V)
        //   slot (j, c), j != i:      H2[k2][1+c][i*N+j] if spin(j) == p, else 0   (lanes of the other spin read a zero word)
        //   padding:                  0
        // all times 1 / n_p, which multiplies the WEIGHT operand of the sub-phase instead (one factor per row).
        auto to_pairs = [&](int p) {
            int lq2 = lq, lr2 = lr;
            asm volatile("" : "+v"(lq2), "+v"(lr2));
            const int d2 = 16 * t + pos16<T>(lr2);
            const bool g2 = d2 >= 2 && d2 < D;
            const int j2 = g2 ? (d2 - 2) / 3 : -1, c2 = g2 ? (d2 - 2) - 3 * j2 : 0;
            const bool jp = (j2 >= n_up ? 1 : 0) == p;
#pragma unroll
            for (int s = 0; s < ST; ++s) {
                const int i = iel[s];
                const bool own = g2 && j2 == i, pair = g2 && j2 != i, mean = d2 < 2 || own;
                const int comp = d2 == 0 ? 0 : (d2 == 1 ? 4 : 1 + c2);
                const T* pm = M2w + (((size_t)i * nch + p) * K2 + lq2) * 5 + comp;
                const T* ph = H2w + ((size_t)lq2 * 5 + 1 + c2) * NP + (size_t)i * N + (pair ? j2 : 0);
                bp[s] = mean ? pm : ((pair && jp) ? ph : A.zero);
                binc[s] = mean ? 20u : ((pair && jp) ? 20u * (unsigned)NP : 0u);
            }
        };
        auto mm = [&](int u) {
#pragma unroll
            for (int s = 0; s < ST; ++s)
#pragma unroll
                for (int a = 0; a < NB; ++a) acc[s][a] = mfma16(bv[u][s], av[u][a], acc[s][a]);
        };
        if (PIPE > 0 && !GATHER) {
            // all Kh + nch K2 rows from G; the ring was primed by the previous pass (or in front of the loop)
            stamp();                                   // 1
            for (int ks = 0; ks + 2 * NS <= nks1; ks += NS) {
#pragma unroll
                for (int u = 0; u < NS; ++u) { mm(u); load_g(u); }
            }
#pragma unroll
            for (int u = 0; u < NS; ++u) mm(u);
        } else {
            if (q > 0) pass_pointers(q);
            auto one = [&](bool pairs, T f) {
                load_w(0);
#pragma unroll
                for (int s = 0; s < ST; ++s) { bv[0][s] = *bp[s]; bp[s] += pairs ? (size_t)binc[s] : (size_t)4 * P; }
                if (pairs) {
#pragma unroll
                    for (int a = 0; a < NB; ++a) av[0][a] *= f;
                }
                mm(0);
            };
            for (int ks = 0; ks < nks1; ++ks) one(false, T(1));
            if (GATHER) {
                to_pairs(0);
                for (int ks = 0; ks < nks2; ++ks) one(true, inv_n0);
                if (nch == 2) {
                    to_pairs(1);
                    for (int ks = 0; ks < nks2; ++ks) one(true, inv_n1);
                }
            }
        }
        stamp();                                       // 2: products done
        // ---- epilogue of the pass, one electron (column tile) after the other.  All loads of an electron are requested before
        //      its first store (loads and stores retire through one counter: a load behind a store waits for the store's ack).
        acc_t hv[2][NB];
        T yv[2][NB];
        auto fetch = [&](int s) {
            const unsigned gb = (unsigned)iel[s] * gts * (unsigned)sizeof(T) + rowb;
            const unsigned yb = (unsigned)iel[s] * (unsigned)Nout * (unsigned)sizeof(T) + featb;
#pragma unroll
            for (int a = 0; a < NB; ++a) {
                if (RES) hv[s & 1][a] = load_own<T>(Gw, gb + (unsigned)(16 * a) * PB, lq);
                if (t != 0) yv[s & 1][a] = at_b<T>(Yw, yb + (unsigned)(16 * a) * (unsigned)sizeof(T));
            }
        };
        if (PIPE > 0 && !GATHER && q + 1 < nquad) {
            pass_pointers(q + 1);
#pragma unroll
            for (int u = 0; u < NS; ++u) load_g(u);
        }
        fetch(0);
#pragma unroll
        for (int s = 0; s < ST; ++s) {
            if (s >= ecount) continue;
            if (DS_EXP(A.dbg & 1)) {                       // (experiment: keep the accumulators alive, nothing else)
                T v = 0;
#pragma unroll
                for (int a = 0; a < NB; ++a) v += acc[s][a][0] + acc[s][a][1] + acc[s][a][2] + acc[s][a][3];
                if (v == T(12345.678)) at_b<T>(Go, rowb) = v;
                continue;
            }
            if (s + 1 < ST) fetch(s + 1);
            const unsigned gb = (unsigned)iel[s] * gts * (unsigned)sizeof(T) + rowb;      // (electron, feature n0 + lr, first slot of the tile)
            const unsigned yb = (unsigned)iel[s] * (unsigned)Nout * (unsigned)sizeof(T) + featb;
            T y[NB], d1[NB];
#pragma unroll
            for (int a = 0; a < NB; ++a) {
                if (t == 0) {
                    // value slot = (lane row 0, register 0): tanh in lane row 0, handed to the other rows by a row sum with zeros
                    const T yy = lq == 0 ? ds_tanh(acc[s][a][0]) : T(0);
                    y[a] = rows4_sum(yy);
                } else
                    y[a] = yv[s & 1][a];
                d1[a] = 1 - y[a] * y[a];
            }
#pragma unroll
            for (int a = 0; a < NB; ++a) {
                acc_t o;
                T sq = T(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const T z = acc[s][a][r];
                    const int d = 16 * t + own16<T>(lq, r);
                    sq += (d >= 2 && d < D) ? z * z : T(0);
                    o[r] = d1[a] * z;
                }
                if (t == 0 && lq == 0) o[0] = y[a];
                if (RES) o = (hv[s & 1][a] + o) * rs2;
                const T ss = DS_EXP(A.dbg & 128) ? sq : rows4_sum(sq);
                if (MSUM_REG) msr[a] += o;
                else if (!DS_EXP(A.dbg & 64)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) msum[(4 * a + r) * 64 + lane] += o[r];      // spin sum of the slot tile (wave-private LDS)
                }
                const unsigned off = gb + (unsigned)(16 * a) * PB;
                if DS_EXP(A.dbg & 256) {
                    if (o[0] == T(12345.678)) at_b<T>(Go, off) = o[1] + o[2] + o[3];
                } else if (t == 0) {
                    // (the Laplacian slot -- lane row 0, register 1 -- is written by k_layer_fin)
                    at_b<T>(Go, off + own_off0<T>(lq)) = o[0];
                    if (lq != 0) at_b<T>(Go, off + own_off0<T>(lq) + (unsigned)sizeof(T)) = o[1];
                    typedef typename Half2<T>::type h2;
                    at_b<h2>(Go, off + own_off1<T>(lq)) = h2{o[2], o[3]};
                } else
                    store_own<T>(Go, off, lq, o, true);
                if (lq == 0 && !DS_EXP(A.dbg & 512)) {
                    const unsigned fo = yb + (unsigned)(16 * a) * (unsigned)sizeof(T);
                    at_b<T>(SSw, fo) = ss;
                    if (t == 0) {
                        at_b<T>(Yw, fo) = y[a];
                        at_b<T>(ZLw, fo) = d1[a] * acc[s][a][1];
                    }
                }
            }
            stamp();                                   // 3..6: electron s done (stores retired)
        }
    }
    // partial spin mean of this slot tile over the group's electrons
#pragma unroll
    for (int a = 0; a < NB; ++a) {
        const unsigned off = rowb + (unsigned)(16 * a) * PB;
        acc_t m;
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = (MSUM_REG ? msr[a][r] : msum[(4 * a + r) * 64 + lane]) * inv_ns;
        if (t == 0) {
            at_b<T>(MPw, off + own_off0<T>(lq)) = m[0];
            if (lq != 0) at_b<T>(MPw, off + own_off0<T>(lq) + (unsigned)sizeof(T)) = m[1];
            typedef typename Half2<T>::type h2;
            at_b<h2>(MPw, off + own_off1<T>(lq)) = h2{m[2], m[3]};
        } else
            store_own<T>(MPw, off, lq, m, false);
    }
    if (A.clk && wave == 0 && lane == 0) {
        atomicAdd(A.clk, (unsigned long long)(clock64() - clk_c0));
        atomicAdd(A.clk + 1, (unsigned long long)(wall_clock64() - clk_r0));
    }
}

// Laplacian slot of the layer output:  o_L = y' z_L + y'' sum_d z_d^2  (sum over ALL gradient slots: the slot-tile partials of
// k_layer_unit, added in tile order), residual, and its partial spin mean.  grid (n_groups, walkers), block = Nout threads.
template <typename T, bool RES>
__global__ void __launch_bounds__(1024) k_layer_fin(SysDev<T> S, LayerArgs<T> A) {
    const int g = blockIdx.x, w = blockIdx.y, n = threadIdx.x;
    if (n >= A.Nout) return;
    const int e0 = S.grp_e0[g], ng = S.grp_n[g], P = S.P, ntile = P / 16;
    const T inv_ns = T(1) / T(S.grp_sp[g] == 0 ? S.n_up : S.n_dn);
    const T rs2 = T(0.70710678118654752440);
    const T* Yw = A.Y + (size_t)w * A.lay_ws;
    const T* ZLw = A.ZLD + (size_t)w * A.lay_ws;
    const T* SSw = A.SSP + (size_t)w * ntile * A.lay_ws;
    const T* Gw = A.Gin + (size_t)w * A.g_ws;
    T* Go = A.Gout + (size_t)w * A.g_ws;
    T sumL = 0;
    for (int e = 0; e < ng; ++e) {
        const int i = e0 + e;
        const size_t fo = (size_t)i * A.Nout + n;
        const T y = Yw[fo], d1 = 1 - y * y, d2 = -2 * y * d1;
        T ss = 0;
        for (int t = 0; t < ntile; ++t) ss += SSw[(size_t)t * A.lay_ws + fo];
        T oL = ZLw[fo] + d2 * ss;
        const size_t go = (size_t)i * A.g_ts + (size_t)n * P + 1;
        if (RES) oL = (Gw[go] + oL) * rs2;
        Go[go] = oL;
        sumL += oL;
    }
    A.MEANP[(size_t)w * A.mp_ws + ((size_t)g * A.Nout + n) * P + 1] = sumL * inv_ns;
}

}  // namespace ds
