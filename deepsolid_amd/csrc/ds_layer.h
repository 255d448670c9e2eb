// ds_layer.h -- one-electron-stream layer of the forward-Laplacian chain, "electron group" formulation (round 3).
//
//   k_layer_group   z_i = W_loc^T [h_i ; mean_j h2_ji] + S  ->  tanh chain rule on the jets  ->  residual  ->  G_out
//                   for a GROUP of up to LG_GE electrons of one spin per workgroup (network.py:305-332, 521-528).
//   k_m2_means      the partner means of the pair stream as 5-jets (value, d/dr, Laplacian) per (electron, spin, feature).
//   k_group_fold    partial spin means of several groups -> one mean per spin.
//
// Why groups.  A workgroup of k_jet_gemm owns ONE electron, so the spin means of a layer's output (the next layer's shared
// term) needed a second pass over all of G (k_shared_term: 4 GB per 1024-walker launch at 24 electrons), and the pair-mean
// rows of the layer input had to be expanded to dense jet rows in HBM first (k_m2_expand: 1.4 GB per launch).  Here the
// four waves of a workgroup (64 output features each) walk the group's electrons four at a time: a "pass" is one 16-slot
// tile of four electrons (4 x 4 accumulator tiles).  With the slot tile as the OUTER loop
//   * the sum over the group's electrons of an output tile is a lane-local sum over accumulator tiles, carried in 16
//     registers across the passes of a slot tile and written once: MEANP[walker][group][n][slot] (already divided by the
//     spin's electron count) -- the input of the next layer's shared term, no second pass over G;
//   * the rows k >= Kh of the B operand (pair means) are generated in the operand load from the pair stream's 5-jets:
//     slot (j, c) of row (spin s, k2) of electron i is  H2[k2][1+c][i*N+j] / n_s  (j in s, j != i),  -mean_c (j == i),
//     the mean value / Laplacian for slots 0 / 1, zero otherwise (network.py:323-328; d/dx_j = +d/dr, d/dx_i = -d/dr);
//   * the per-electron quantities that tie the slot tiles of an electron together -- y = tanh(z_0) and the Laplacian
//     accumulator  y' z_L + y'' sum_d z_d^2 -- live in wave-private LDS (one value per lane: lane <-> feature), so no
//     barrier is needed anywhere; the Laplacian slot is written by a short final step.
#pragma once
#include "ds_gemm.h"

namespace ds {

constexpr int LG_GE = 12;   // electrons per group (three passes per slot tile)
constexpr int LG_LDS_PER_WAVE = 2 * LG_GE * 64;                // elements of wave-private LDS (y and the Laplacian accumulator per electron and feature)
template <typename T> inline size_t layer_group_lds_bytes(unsigned threads) { return (size_t)(threads / 64) * LG_LDS_PER_WAVE * sizeof(T); }

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
    // optional in-kernel clock probe (ds_profile_*): wave 0 of every workgroup adds its shader-clock cycles (s_memtime) and
    // its constant-rate 100 MHz ticks (s_memrealtime) between entry and exit: clk[0] += cycles, clk[1] += ticks
    const T* zero;         // one element holding 0 (the B operand of lanes that contribute nothing in a pair sub-phase)
    unsigned long long* clk;
    int dbg;               // timing experiments only (DS_LG_DBG): 1 = skip the epilogue arithmetic, 2 = skip the pair-mean rows, 4 = skip the G rows
};

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
// ST = electrons (column tiles) per pass, WPS = waves per SIMD the register budget is cut for.
template <typename T, bool RES, int PIPE, int ST, int WPS>
__global__ void __launch_bounds__(256, WPS) k_layer_group(SysDev<T> S, LayerArgs<T> A) {
    typedef typename Acc4<T>::type acc_t;
    constexpr int NB = 4;
    // XCD-aware placement as in k_jet_gemm: all workgroups of one walker get linear ids of one residue class mod 8
    int gx = blockIdx.x, w = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const unsigned b = blockIdx.y * gridDim.x + blockIdx.x, q = b >> 3;
        w = (q / gridDim.x) * 8 + (b & 7);
        gx = q % gridDim.x;
    }
    const int gzf = gridDim.x / S.n_groups, zb = gx % gzf, g = gx / gzf;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4, n0 = (zb * (blockDim.x >> 6) + wave) * 16 * NB;
    const int Nout = A.Nout;
    if (n0 >= Nout) return;                               // (no barriers below: the LDS state is wave-private)
    long long clk_c0 = 0, clk_r0 = 0;
    if (A.clk && wave == 0) { clk_c0 = clock64(); clk_r0 = wall_clock64(); }
    // Phase skew.  The two waves that share a SIMD run identical work and would stay in lockstep -- both in the MFMA loop (sharing
    // the matrix pipe), then both in the latency-bound epilogue with the pipe idle.  The wave in the odd hardware slot starts
    // half a main loop late, so that one wave's epilogue runs under the other's MFMAs.  Timing only: no result depends on it.
    // (timing experiment, DS_LG_DBG & 32: wave 0 of workgroup 0 writes shader-clock stamps at its phase boundaries to clk[2 + i])
    unsigned long long* tl = (A.clk && (A.dbg & 32) && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0 && lane == 0) ? A.clk + 2 : nullptr;
    int n_tl = 0;
    auto stamp = [&]() {
        if (tl && n_tl < 1000) { __builtin_amdgcn_s_waitcnt(0); tl[n_tl++] = (unsigned long long)clock64(); }
    };
    if (A.dbg & 8) {
        const unsigned slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11));        // HW_ID.wave_id: wave slot of the SIMD
        if (A.dbg & 16) {            // experiment: pseudo-random skew per workgroup
            const unsigned b = blockIdx.y * gridDim.x + blockIdx.x, h = (b * 2654435761u) >> 28;
            for (unsigned i = 0; i < h * (unsigned)(A.dbg >> 8); ++i) __builtin_amdgcn_s_sleep(127);
        } else if (slot & 1u) {
            for (int i = 0; i < (A.dbg >> 8); ++i) __builtin_amdgcn_s_sleep(127);
        }
    }
    extern __shared__ __attribute__((aligned(16))) char lg_smem[];
    T* ylds = reinterpret_cast<T*>(lg_smem) + (size_t)wave * LG_LDS_PER_WAVE;       // [electron of the group][feature]: y = tanh(z_0)
    T* accL = ylds + LG_GE * 64;                                                    // [electron of the group][feature]: Laplacian accumulator
    const int e0 = S.grp_e0[g], ng = S.grp_n[g], gsp = S.grp_sp[g];
    const int n_up = S.n_up;
    const T inv_ns = T(1) / T(gsp == 0 ? n_up : S.n_dn);
    const unsigned P = (unsigned)S.P;
    const int D = S.D, N = S.N, NP = S.NP, nch = S.nch, K2 = A.K2;
    const int nks1 = (A.dbg & 4) ? 8 : A.Kh / 4, nks2 = (A.dbg & 2) ? 2 : K2 / 4;
    const T* Gw = A.Gin + (size_t)w * A.g_ws;              // (uniform bases; the lane's place is a 32-bit byte offset)
    T* Go = A.Gout + (size_t)w * A.g_ws;
    const T* Sw = A.Sb + (size_t)w * Nout * P;
    const T* H2w = A.H2 + (size_t)w * A.h2_ws;
    const T* M2w = A.M2V + (size_t)w * A.m2v_ws;
    T* MPw = A.MEANP + (size_t)w * A.mp_ws + (size_t)g * Nout * P;
    const unsigned gts = (unsigned)A.g_ts;
    const int ntile = (int)P / 16, nquad = (ng + ST - 1) / ST;
    const T rs2 = T(0.70710678118654752440);
    const T inv_n0 = T(1) / T(n_up), inv_n1 = T(1) / T(S.n_dn > 0 ? S.n_dn : 1);
    const unsigned PB = P * (unsigned)sizeof(T);                          // row pitch in bytes
    // epilogue: this lane owns features n0 + 16a + lr (a = 0..3) and, in slot tile t, the slots 16t + own16(lq, r) (r = 0..3)
    const unsigned rowb = (unsigned)(n0 + lr) * PB;

    // the shared term of the NEXT pass is requested at the start of every epilogue (into registers the operand ring has just
    // released), so a pass starts with its accumulators ready instead of waiting for sixteen loads
    acc_t Scur[NB];
#pragma unroll
    for (int a = 0; a < NB; ++a) Scur[a] = load_own<T>(Sw, rowb + (unsigned)(16 * a) * PB, lq);
    for (int t = 0; t < ntile; ++t) {
        const int dl = 16 * t + pos16<T>(lr);              // slot this lane LOADS in every column tile (A-operand row lr)
        const unsigned tb = (unsigned)(16 * t) * (unsigned)sizeof(T);
        acc_t msum[NB];                                    // spin sums of the slot tile over the group's electrons
#pragma unroll
        for (int a = 0; a < NB; ++a) msum[a] = acc_t{0, 0, 0, 0};
        for (int q = 0; q < nquad; ++q) {
            const int ecount = ng - ST * q < ST ? ng - ST * q : ST;
            int iel[ST];                                   // electron of column tile s (dead tiles repeat the last one)
#pragma unroll
            for (int s = 0; s < ST; ++s) iel[s] = e0 + (ST * q + s < ng ? ST * q + s : ng - 1);
            // z = W x + (S + b): the accumulators of all four electrons start at the walker's shared term
            stamp();                                       // 0: pass start
            acc_t acc[ST][NB];
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int s = 0; s < ST; ++s) acc[s][a] = Scur[a];
            // ---- A operand (jet rows): rows k < Kh from G (phase 1), rows k >= Kh generated from the pair stream (one
            //      sub-phase per partner spin p: row = (p, k2 = 4 ks + lq)).  Lane kinds in the pair sub-phases: mean lanes
            //      (slots 0 / 1 and the electron's own three gradient slots, negated), pair-stream lanes (the other gradient
            //      slots, non-zero only when slot electron jd has spin p), zero lanes (padding).
            // (lane ids made opaque per pass: the pointer pieces derived from them are then recomputed here -- a few VALU
            //  instructions -- instead of being hoisted out of the pass loop, spilled, and reloaded in front of the k-loop,
            //  which costs a full vmcnt(0) drain inside it)
            int lqp = lq, lrp = lr;
            asm volatile("" : "+v"(lqp), "+v"(lrp));
            const T* Wl = A.W + (size_t)lqp * Nout + n0 + lrp;
            const T* bp[ST];
            unsigned binc[ST];                             // element step per k-step of the pair sub-phases
#pragma unroll
            for (int s = 0; s < ST; ++s) bp[s] = Gw + ((size_t)iel[s] * gts + (size_t)lqp * P + dl);
            // Pair sub-phase p (partner spin p): B-operand row (p, k2 = 4 ks + lq) of electron i is, per slot,
            //   slots 0 / 1:              + sum_j value / Laplacian            (M2V, signed sums)
            //   own gradient slots:       - sum_j d/dr                          (M2V)
            //   slot (j, c), j != i:      H2[k2][1+c][i*N+j] if spin(j) == p, else 0   (lanes of the other spin read a zero word)
            //   padding:                  0
            // all times 1 / n_p, which multiplies the WEIGHT operand of the sub-phase instead (one factor per row).
            // (everything is computed here from opaque copies of the lane ids: hoisted in front of the G loop these pieces are
            //  spilled and their reload drains the operand ring)
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
            if (PIPE > 0) {
                // operand ring of NS = PIPE k-steps: a set is re-requested right after its MFMAs are issued, i.e. NS - 1 k-steps
                // (16 MFMAs each) before it is needed again.  With two sets a wave that runs alone on its SIMD (its partner in
                // the epilogue) waits for memory every k-step; four sets cover the latency, so one wave's epilogue really
                // runs under the other's MFMAs.  All phase changes are straight-line code: every load is unconditional.
                constexpr int NS = PIPE > 0 ? PIPE : 1;
                T av[NS][NB], bv[NS][ST];
                auto load_w = [&](int u) {
#pragma unroll
                    for (int a = 0; a < NB; ++a) av[u][a] = Wl[16 * a];
                    Wl += (size_t)4 * Nout;
                };
                auto load_g = [&](int u) {
                    load_w(u);
#pragma unroll
                    for (int s = 0; s < ST; ++s) { bv[u][s] = *bp[s]; bp[s] += (size_t)4 * P; }
                };
                auto load_m = [&](int u) {
                    load_w(u);
#pragma unroll
                    for (int s = 0; s < ST; ++s) { bv[u][s] = *bp[s]; bp[s] += (size_t)binc[s]; }
                };
                auto mm = [&](int u) {
#pragma unroll
                    for (int s = 0; s < ST; ++s)
#pragma unroll
                        for (int a = 0; a < NB; ++a) acc[s][a] = mfma16(bv[u][s], av[u][a], acc[s][a]);
                };
                // (w_scaled: the sets in flight carry weights of pair sub-phase p: times 1 / n_p)
                auto mm_w = [&](int u, T f) {
#pragma unroll
                    for (int a = 0; a < NB; ++a) av[u][a] *= f;
                    mm(u);
                };
#pragma unroll
                for (int u = 0; u < NS; ++u) load_g(u);
                stamp();                                   // 1: accumulators set, first operands landed (the stamp waits)
                for (int ks = 0; ks + 2 * NS <= nks1; ks += NS) {
#pragma unroll
                    for (int u = 0; u < NS; ++u) { mm(u); load_g(u); }
                }
                stamp();                                   // 2: G rows done
                to_pairs(0);
#pragma unroll
                for (int u = 0; u < NS; ++u) { mm(u); load_m(u); }           // last NS k-steps of G; first NS of the pair rows
                for (int ks = 0; ks + 2 * NS <= nks2; ks += NS) {
#pragma unroll
                    for (int u = 0; u < NS; ++u) { mm_w(u, inv_n0); load_m(u); }
                }
                if (nch == 2) {
                    to_pairs(1);
#pragma unroll
                    for (int u = 0; u < NS; ++u) { mm_w(u, inv_n0); load_m(u); }      // last NS of spin 0; first NS of spin 1
                    for (int ks = 0; ks + 2 * NS <= nks2; ks += NS) {
#pragma unroll
                        for (int u = 0; u < NS; ++u) { mm_w(u, inv_n1); load_m(u); }
                    }
#pragma unroll
                    for (int u = 0; u < NS; ++u) mm_w(u, inv_n1);
                } else {
#pragma unroll
                    for (int u = 0; u < NS; ++u) mm_w(u, inv_n0);
                }
                stamp();                                   // 3: pair rows done
            } else {
                T av[NB], bv[ST];
                auto one = [&](bool pairs, T f) {
#pragma unroll
                    for (int a = 0; a < NB; ++a) av[a] = Wl[16 * a];
                    Wl += (size_t)4 * Nout;
#pragma unroll
                    for (int s = 0; s < ST; ++s) { bv[s] = *bp[s]; bp[s] += pairs ? (size_t)binc[s] : (size_t)4 * P; }
                    if (pairs) {
#pragma unroll
                        for (int a = 0; a < NB; ++a) av[a] *= f;
                    }
#pragma unroll
                    for (int s = 0; s < ST; ++s)
#pragma unroll
                        for (int a = 0; a < NB; ++a) acc[s][a] = mfma16(bv[s], av[a], acc[s][a]);
                };
                for (int ks = 0; ks < nks1; ++ks) one(false, T(1));
                to_pairs(0);
                for (int ks = 0; ks < nks2; ++ks) one(true, inv_n0);
                if (nch == 2) {
                    to_pairs(1);
                    for (int ks = 0; ks < nks2; ++ks) one(true, inv_n1);
                }
            }
            // ---- epilogue of the pass, one electron (column tile) after the other: tanh chain rule, residual, store, spin sums
            // residual rows requested two electrons ahead of their use (the registers of a finished electron's accumulators
            // take the next request): the memory latency is paid once per pass, not once per electron
            acc_t hvall[ST][NB];
            auto fetch_res = [&](int s) {
                const unsigned gb = (unsigned)iel[s] * gts * (unsigned)sizeof(T) + rowb + tb;
#pragma unroll
                for (int a = 0; a < NB; ++a) hvall[s][a] = load_own<T>(Gw, gb + (unsigned)(16 * a) * PB, lq);
            };
            if (RES) { fetch_res(0); if (ST > 1) fetch_res(1); }
#pragma unroll
            for (int s = 0; s < ST; ++s) {
                if (s >= ecount) continue;
                if (A.dbg & 1) {                           // (timing experiment: keep the accumulators alive, nothing else)
                    T v = 0;
#pragma unroll
                    for (int a = 0; a < NB; ++a) v += acc[s][a][0] + acc[s][a][1] + acc[s][a][2] + acc[s][a][3];
                    if (v == T(12345.678)) at_b<T>(Go, rowb) = v;
                    continue;
                }
                const int el = ST * q + s;
                T* yl = ylds + el * 64;
                T* al = accL + el * 64;
                const unsigned gb = (unsigned)iel[s] * gts * (unsigned)sizeof(T) + rowb + tb;      // byte offset of (electron, feature n0 + lr, slot d0)
                if (t == 0) {
                    // value slot = (lane row 0, register 0): spread the 64 pre-activations over the lanes, ONE tanh, keep y per feature
                    if (lq == 0) {
#pragma unroll
                        for (int a = 0; a < NB; ++a) yl[16 * a + lr] = acc[s][a][0];
                    }
                    const T y = ds_tanh(yl[lane]);
                    yl[lane] = y;
                }
                T y[NB], d1[NB], ss[NB];
#pragma unroll
                for (int a = 0; a < NB; ++a) {
                    y[a] = yl[16 * a + lr];
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
                    if (RES) o = (hvall[s][a] + o) * rs2;
                    ss[a] = rows4_sum(sq);
                    msum[a] += o;
                    const unsigned off = gb + (unsigned)(16 * a) * PB;
                    if (t == 0) {
                        // (the Laplacian slot -- lane row 0, register 1 -- is written by the final step)
                        at_b<T>(Go, off + own_off0<T>(lq)) = o[0];
                        if (lq != 0) at_b<T>(Go, off + own_off0<T>(lq) + (unsigned)sizeof(T)) = o[1];
                        typedef typename Half2<T>::type h2;
                        at_b<h2>(Go, off + own_off1<T>(lq)) = h2{o[2], o[3]};
                    } else
                        store_own<T>(Go, off, lq, o, true);
                }
                // Laplacian accumulator of the electron's features 16a + lr, kept by lane row 0:  y' z_L + y'' sum_d z_d^2
                if (lq == 0) {
#pragma unroll
                    for (int a = 0; a < NB; ++a) {
                        const T d2 = -2 * y[a] * d1[a];
                        const T base = t == 0 ? d1[a] * acc[s][a][1] : al[16 * a + lr];
                        al[16 * a + lr] = base + d2 * ss[a];
                    }
                }
                if (RES && s + 2 < ST) fetch_res(s + 2);
                if (s == (ecount > 1 ? 1 : 0)) {
                    // shared term of the next pass (same slot tile, or the next one after the last quad; the very last request
                    // repeats the current tile and is dropped), into registers of finished accumulators
                    const int tn = q + 1 < nquad ? t : (t + 1 < ntile ? t + 1 : t);
                    unsigned so = rowb + (unsigned)(16 * tn) * (unsigned)sizeof(T);
                    asm volatile("" : "+v"(so));
#pragma unroll
                    for (int a = 0; a < NB; ++a) Scur[a] = load_own<T>(Sw, so + (unsigned)(16 * a) * PB, lq);
                }
                stamp();                                   // 4..7: electron s done
            }
        }
        // partial spin mean of this slot tile over the group's electrons
#pragma unroll
        for (int a = 0; a < NB; ++a) {
            const unsigned off = rowb + tb + (unsigned)(16 * a) * PB;
            const acc_t m = msum[a] * inv_ns;
            if (t == 0) {
                at_b<T>(MPw, off + own_off0<T>(lq)) = m[0];
                if (lq != 0) at_b<T>(MPw, off + own_off0<T>(lq) + (unsigned)sizeof(T)) = m[1];
                typedef typename Half2<T>::type h2;
                at_b<h2>(MPw, off + own_off1<T>(lq)) = h2{m[2], m[3]};
            } else
                store_own<T>(MPw, off, lq, m, false);
        }
    }
    // ---- Laplacian slot: lane <-> feature n0 + lane
    {
        const unsigned nP = (unsigned)(n0 + lane) * PB + (unsigned)sizeof(T);
        T sumL = 0;
        for (int e = 0; e < ng; ++e) {
            const unsigned off = (unsigned)(e0 + e) * gts * (unsigned)sizeof(T) + nP;
            T oL = accL[e * 64 + lane];
            if (RES) oL = (at_b<T>(Gw, off) + oL) * rs2;
            at_b<T>(Go, off) = oL;
            sumL += oL;
        }
        at_b<T>(MPw, nP) = sumL * inv_ns;
    }
    if (A.clk && wave == 0 && lane == 0) {
        atomicAdd(A.clk, (unsigned long long)(clock64() - clk_c0));
        atomicAdd(A.clk + 1, (unsigned long long)(wall_clock64() - clk_r0));
    }
}

}  // namespace ds
