// ds_wide.h -- the per-electron GEMM kernels for WIDE slot ranges (more than 10 jet-slot tiles: N > 52 electrons).
//
// A wave of k_jet_gemm holds all ST slot tiles of its 16 NB features in accumulators; beyond 10 tiles that leaves one wave per SIMD
// (152+ accumulator registers, four waves per workgroup, one workgroup per CU): nothing covers a wave's epilogue and a lone wave
// reaches about 3/4 of the MFMA issue rate.  Here a wave keeps the 24-electron wave tile -- 64 features x STC <= 5 slot tiles, two
// waves per SIMD -- and walks the slot range in CHUNKS of STC tiles: per chunk the full K loop (the weights are re-streamed from the
// L2, the jet rows are read once) and the epilogue.  What couples the slot tiles of a row goes from chunk to chunk in three
// numbers per row, kept one row group per lane ("lane-q form": the value of row group q = 4 a + r sits in lane lr = q of every
// 16-lane row, as the tanh of k_jet_gemm's epilogue already does):
//     layer epilogues:  y = tanh z_0 and z_L from chunk 0 (which holds slots 0 and 1), the running sum of squares; the Laplacian
//                       slot of the output is written after the last chunk;
//     orbital head:     phi_0, phi_L from chunk 0, the running sum 2 sum_c phi_own,c q_c; the Laplacian slot is written last.
// The last chunk may be narrower than STC: its surplus columns read on into the next row (every load of the operand ring stays
// unconditional, with immediate offsets) and are neither summed nor stored.
#pragma once
#include "ds_gemm.h"

namespace ds {

template <typename T> struct WideCarry { T yall, zLall, ssall, r1all; };
// operand ring depth: three k-steps in float64 (the chunk loop's carried state leaves no room for the fourth), four in float32
template <typename T> constexpr int wide_ring_sets() { return sizeof(T) == 8 ? 3 : 4; }

// K loop of one chunk: acc += W[:, n0 ..]^T X[:, tiles], operand ring of k_jet_gemm, residual stash on the way (NA blocks).
// RING: the four-set operand ring (K a multiple of 16: hidden layers, orbital head) or the plain loop (layer 0: K = 12, 16, ...);
// a compile-time choice -- two loop shapes in one kernel cost 700 bytes of spills.
template <typename T, int NB, int STC, int NA, bool RING>
__device__ __forceinline__ void wide_kloop(typename Acc4<T>::type (&acc)[NB][STC], const T* Wl, size_t wstep, const T* Xl, size_t xstep,
                                           int co, int nks, T* stash, int n0, int lane) {
    constexpr int NSET = wide_ring_sets<T>();
    T av[NSET][NB], bv[NSET][STC];
    auto load_set = [&](int u) {
#pragma unroll
        for (int a = 0; a < NB; ++a) av[u][a] = Wl[16 * a];
#pragma unroll
        for (int s = 0; s < STC; ++s) bv[u][s] = Xl[(co + 16 * s)];
        Wl += wstep;
        Xl += xstep;
    };
    auto step = [&](int u, int k) {
        if (NA > 0) {
            const int j = k - (n0 >> 2);
            if (j >= 0 && j < 4 * NA) {
#pragma unroll
                for (int s = 0; s < STC; ++s) stash[(j * STC + s) * 64 + lane] = bv[u][s];
            }
        }
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int s = 0; s < STC; ++s) acc[a][s] = mfma16(av[u][a], bv[u][s], acc[a][s]);
    };
    if constexpr (RING) {
        // (every load of the steady state unconditional; the last rounds reload conditionally -- nks need not divide by NSET)
#pragma unroll
        for (int u = 0; u < NSET; ++u) load_set(u);
        int ks = 0;
        for (; ks + 2 * NSET <= nks; ks += NSET) {
#pragma unroll
            for (int u = 0; u < NSET; ++u) { step(u, ks + u); load_set(u); }
        }
#pragma unroll
        for (int u = 0; u < NSET; ++u) {
            if (ks + u < nks) step(u, ks + u);
            if (ks + u + NSET < nks) load_set(u);
        }
        ks += NSET;
#pragma unroll
        for (int u = 0; u < NSET; ++u)
            if (ks + u < nks) step(u, ks + u);
    } else {
        for (int ks = 0; ks < nks; ++ks) { load_set(0); step(0, ks); }
    }
}

// Layer epilogue of one chunk (EPI 1 / 2 / 9 as in k_jet_gemm; rf as in layer_epilogue, with the chunk as its argument).
//   Gi / Go: residual rows / output tile WITHOUT the lane offset; (co + 16 * s) = 16 * (slot tile of accumulator column s)
template <typename T, int NB, int STC, int EPI, int NA, typename RF = NoResidFn>
__device__ __forceinline__ void wide_layer_epilogue(typename Acc4<T>::type (&acc)[NB][STC], const T* __restrict__ Gi, T* __restrict__ Go,
                                                    const T* stash, int n0, int lane, int P, int co, int cw, bool first,
                                                    WideCarry<T>& cy, RF&& rf = RF()) {
    constexpr bool RESID = EPI == 2;
    constexpr bool CUSTOM = !std::is_same<typename std::decay<RF>::type, NoResidFn>::value;
    const int lr = lane & 15;
    const T rs2 = T(0.70710678118654752440);
    constexpr int NQ = NB * 4, NQL = NA * 4, NQG = (RESID && !CUSTOM) ? NQ - NQL : 0,
                  DMAX = 40 / (STC * (int)sizeof(T) / 4) > 1 ? 40 / (STC * (int)sizeof(T) / 4) : 1, DEPTH = NQG < DMAX ? NQG : DMAX;
    T hq[DEPTH > 0 ? DEPTH : 1][STC];
    auto fetch = [&](int q, int slot) {
        const int n = n0 + 16 * (q >> 2) + acc_row<T>(lane, q & 3);
#pragma unroll
        for (int s = 0; s < STC; ++s) hq[slot][s] = Gi[(size_t)n * P + (co + 16 * s) + lr];
    };
#pragma unroll
    for (int g = 0; g < DEPTH; ++g) fetch(NQL + g, g);
    if (NA > 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (first) {
        T zsel = 0, zl = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const T v = row16_bcast<0>(acc[q >> 2][0][q & 3]), vl = row16_bcast<1>(acc[q >> 2][0][q & 3]);
            zsel = lr == q ? v : zsel;
            zl = lr == q ? vl : zl;
        }
        cy.yall = ds_tanh(zsel);
        cy.zLall = zl;
        cy.ssall = 0;
        cy.r1all = 0;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int a = q >> 2, r = q & 3;
        const int n = n0 + 16 * a + acc_row<T>(lane, r);
        T z[STC], hv[STC];
#pragma unroll
        for (int s = 0; s < STC; ++s) z[s] = acc[a][s][r];
        if (RESID && !CUSTOM) {
            if (q < NQL) {
                const int rr = 16 * a + acc_row<T>(lane, r);
#pragma unroll
                for (int s = 0; s < STC; ++s) hv[s] = stash[((rr >> 2) * STC + s) * 64 + ((rr & 3) << 4) + lr];
            } else {
                const int slot = (q - NQL) % (DEPTH > 0 ? DEPTH : 1);
#pragma unroll
                for (int s = 0; s < STC; ++s) hv[s] = hq[slot][s];
                if (q + DEPTH < NQ) fetch(q + DEPTH, slot);
            }
        }
        T ss = 0;
#pragma unroll
        for (int s = 0; s < STC; ++s) {
            const bool use = s < cw && !(first && s == 0 && lr < 2);
            ss += use ? z[s] * z[s] : T(0);
        }
        ss = row16_sum(ss);
        cy.ssall += lr == q ? ss : T(0);
        const T y = row16_bcast_dyn<NQ>(cy.yall, q), d1 = 1 - y * y;
#pragma unroll
        for (int s = 0; s < STC; ++s) {
            T o = d1 * z[s];
            const bool slot0 = first && s == 0 && lr == 0, slot1 = first && s == 0 && lr == 1;
            o = slot0 ? y : o;
            if (EPI == 9) { if (slot0) Go[2 * (size_t)n] = y; continue; }
            if (CUSTOM && RESID) { acc[a][s][r] = o; continue; }
            if (EPI == 2) {
                if (first && s == 0) { const T h1 = row16_bcast<1>(hv[0]); cy.r1all = lr == q ? h1 : cy.r1all; }
                o = (hv[s] + o) * rs2;
            }
            if (s < cw && !slot1) __builtin_nontemporal_store(o, &Go[(size_t)n * P + (co + 16 * s) + lr]);
        }
        if constexpr (CUSTOM && RESID) { if (r == 3) rf(a); }
    }
}
// ... and the Laplacian slot of the output rows after the last chunk (EPI 9: oL into YO).  yl: (y, oL) of the layer-0 output rows
// when the residual is recomputed (k_layer1_lr), else null (the residual's slot 1 was kept in cy.r1all).
template <typename T, int NB, int EPI>
__device__ __forceinline__ void wide_layer_finish(const WideCarry<T>& cy, T* __restrict__ Go, int n0, int lane, int P, const T* yl) {
    const int lr = lane & 15;
    if (lr >= NB * 4) return;
    const int n = n0 + 16 * (lr >> 2) + acc_row<T>(lane, lr & 3);
    const T y = cy.yall, d1 = 1 - y * y, d2 = -2 * y * d1;
    const T oL = d1 * cy.zLall + d2 * cy.ssall;
    if (EPI == 9) { Go[2 * (size_t)n + 1] = oL; return; }
    const T res = yl ? yl[2 * n + 1] : cy.r1all;
    Go[(size_t)n * P + 1] = EPI == 2 ? (res + oL) * T(0.70710678118654752440) : oL;
}

// k_jet_gemm for wide slot ranges: EPI 1 / 2 / 9 (layer epilogues) and 5 (orbital head).  Arguments as k_jet_gemm; NB = 4.
template <typename T, int STC, int EPI>
__global__ void __launch_bounds__(256, 2)
k_jet_gemm_wide(const T* __restrict__ X, size_t x_walker_stride, size_t x_tile_stride, const T* __restrict__ W, int K, int n_tiles,
                T* __restrict__ Z, size_t z_walker_stride, size_t z_tile_stride, int Nout, int P, const T* __restrict__ Sb, OrbEpi<T> oe) {
    typedef typename Acc4<T>::type acc_t;
    constexpr int NB = 4;
    int tile = blockIdx.x, w = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const unsigned b = blockIdx.y * gridDim.x + blockIdx.x, q = b >> 3;
        w = (q / gridDim.x) * 8 + (b & 7);
        tile = q % gridDim.x;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int zb = 0;
    if (gridDim.x > (unsigned)n_tiles) {
        const int gzf = gridDim.x / n_tiles;
        zb = tile % gzf;
        tile /= gzf;
    }
    const int lr = lane & 15, lq = lane >> 4, n0 = (zb * (blockDim.x >> 6) + wave) * 16 * NB;
    if (n0 >= Nout) return;
    const int ntile = P / 16, nchunk = (ntile + STC - 1) / STC, nks = K / 4;
    const T* Xt = X + (size_t)w * x_walker_stride + (size_t)tile * x_tile_stride;
    constexpr bool RESID = EPI == 2;
    constexpr int NA = RESID ? stash_blocks<T, NB, STC>() : 0;
    extern __shared__ __attribute__((aligned(16))) char wide_smem[];
    T* stash = reinterpret_cast<T*>(wide_smem) + (size_t)wave * (NA * 4 * STC * 64);
    WideCarry<T> cy{0, 0, 0, 0};
    // orbital head: phi_0, phi_L and the running own-slot sum of the wave's 2 NB orbital pairs, one pair per lane of a row
    Cx<T> f0all(0, 0), fLall(0, 0), lapall(0, 0);
    for (int c = 0; c < nchunk; ++c) {
        const int t0 = c * STC, cw = ntile - t0 < STC ? ntile - t0 : STC;
        const int co = 16 * t0;                      // first slot of the chunk (columns beyond the last tile read on into the next row: never stored)
        // (an opaque copy of the row stride: the row addresses of S, of the residual and of the output would otherwise be hoisted out of
        //  the chunk loop -- ~100 registers of loop-invariant pointers next to the accumulators)
        int Pl = P;
        asm volatile("" : "+s"(Pl));
        acc_t acc[NB][STC];
        if (EPI != 5 || Sb) {
            const T* Sp0 = Sb + (size_t)w * Nout * P + lr;
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + 16 * a + acc_row<T>(lane, r);
#pragma unroll
                    for (int s = 0; s < STC; ++s) acc[a][s][r] = Sp0[(size_t)n * Pl + (co + 16 * s)];
                }
        } else {
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int s = 0; s < STC; ++s) acc[a][s] = acc_t{0, 0, 0, 0};
        }
        wide_kloop<T, NB, STC, NA, (EPI == 2 || EPI == 5)>(acc, W + n0 + (size_t)lq * Nout + lr, (size_t)4 * Nout, Xt + (size_t)lq * Pl + lr, (size_t)4 * Pl, co, nks, stash,
                                   n0, lane);
        if (EPI != 5) {
            T* Go = Z + (size_t)w * z_walker_stride + (size_t)tile * z_tile_stride;
            wide_layer_epilogue<T, NB, STC, EPI, NA>(acc, Xt, Go, stash, n0, lane, Pl, co, cw, c == 0, cy);
            if (NA > 0) {      // the next chunk's k-loop overwrites the stash
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            // orbital head (k_jet_gemm EPI 5) on the chunk's tiles
            const int i = oe.i0 + tile, so = 2 + 3 * i, base = lane & 48;
            T* Mw = oe.MOUT + (size_t)w * oe.mout_stride + oe.mout_off;
            const size_t tstride = (size_t)oe.n * oe.n * 2 * 16;
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int ab = 0; ab < 2; ++ab) {
                    const int cb = 2 * a + ab;
                    const int p = 8 * (n0 / 16 + a) + lq + 4 * ab;
                    const bool valid = p < oe.nparam;
                    const T* q = oe.Q + ((size_t)(w * oe.N + i) * oe.nparam_max + (valid ? p : 0)) * 10;
                    const Cx<T> qv(q[0], q[1]), qg0(q[2], q[3]), qg1(q[4], q[5]), qg2(q[6], q[7]);
                    Cx<T> phi[STC];
#pragma unroll
                    for (int s = 0; s < STC; ++s) phi[s] = Cx<T>(acc[a][s][2 * ab], acc[a][s][2 * ab + 1]);
                    if (c == 0) {
                        if (oe.bias && valid && lr == 0) { phi[0].re += oe.bias[p]; phi[0].im += oe.bias[oe.nparam + p]; }
                        const Cx<T> f0(row16_bcast<0>(phi[0].re), row16_bcast<0>(phi[0].im)), fL(row16_bcast<1>(phi[0].re), row16_bcast<1>(phi[0].im));
                        f0all.re = lr == cb ? f0.re : f0all.re; f0all.im = lr == cb ? f0.im : f0all.im;
                        fLall.re = lr == cb ? fL.re : fLall.re; fLall.im = lr == cb ? fL.im : fLall.im;
                    }
                    const Cx<T> f0(row16_bcast_dyn<8>(f0all.re, cb), row16_bcast_dyn<8>(f0all.im, cb));
                    // the electron's own three coordinate slots that fall into this chunk
                    Cx<T> own(0, 0);
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        const int sl = so + cc, st = (sl >> 4) - t0;            // accumulator column of that slot (workgroup-uniform)
                        if (st >= 0 && st < cw) {
                            T re = phi[0].re, im = phi[0].im;
#pragma unroll
                            for (int s = 1; s < STC; ++s)
                                if (s == st) { re = phi[s].re; im = phi[s].im; }
                            const Cx<T> fo(__shfl(re, base | (sl & 15)), __shfl(im, base | (sl & 15)));
                            own = own + T(2) * (fo * (cc == 0 ? qg0 : (cc == 1 ? qg1 : qg2)));
                        }
                    }
                    lapall.re += lr == cb ? own.re : T(0);
                    lapall.im += lr == cb ? own.im : T(0);
                    if (valid) {
                        const int kdet = p / oe.norb, m = p % oe.norb;
                        T* mo = Mw + (size_t)kdet * oe.n * oe.n * 2 * Pl + (((size_t)(oe.row0 + tile) * oe.n + m) * 2) * 16 + lr;
                        const Cx<T> t0c = f0 * qg0, t1c = f0 * qg1, t2c = f0 * qg2;
#pragma unroll
                        for (int s = 0; s < STC; ++s) {
                            const int tt = t0 + s;                                   // slot tile of this column
                            T vr = phi[s].re * qv.re - phi[s].im * qv.im, vi = phi[s].re * qv.im + phi[s].im * qv.re;
                            const int dsl = 16 * tt + lr - so;
                            vr += dsl == 0 ? t0c.re : (dsl == 1 ? t1c.re : (dsl == 2 ? t2c.re : T(0)));
                            vi += dsl == 0 ? t0c.im : (dsl == 1 ? t1c.im : (dsl == 2 ? t2c.im : T(0)));
                            if (s < cw && !(tt == 0 && lr == 1)) {
                                __builtin_nontemporal_store(vr, &mo[(size_t)tt * tstride]);
                                __builtin_nontemporal_store(vi, &mo[(size_t)tt * tstride + 16]);
                            }
                        }
                    }
                }
        }
    }
    if (EPI != 5) {
        T* Go = Z + (size_t)w * z_walker_stride + (size_t)tile * z_tile_stride;
        wide_layer_finish<T, NB, EPI>(cy, Go, n0, lane, P, (const T*)nullptr);
    } else if (lr < 2 * NB) {
        // Laplacian slot of the wave's orbital pairs: lane lr = 2 a + ab of every row holds phi_0, phi_L and the own-slot sum
        const int a = lr >> 1, ab = lr & 1, i = oe.i0 + tile;
        const int p = 8 * (n0 / 16 + a) + lq + 4 * ab;
        if (p < oe.nparam) {
            const T* q = oe.Q + ((size_t)(w * oe.N + i) * oe.nparam_max + p) * 10;
            const Cx<T> qv(q[0], q[1]), ql(q[8], q[9]);
            const Cx<T> lap = fLall * qv + f0all * ql + lapall;
            const int kdet = p / oe.norb, m = p % oe.norb;
            T* mo = oe.MOUT + (size_t)w * oe.mout_stride + oe.mout_off + (size_t)kdet * oe.n * oe.n * 2 * P +
                    (((size_t)(oe.row0 + tile) * oe.n + m) * 2) * 16 + 1;
            mo[0] = lap.re;
            mo[16] = lap.im;
        }
    }
}

// k_layer1_lr for wide slot ranges: phase 1 (the per-electron weights C) once, then per chunk phase 2 and the epilogue.
template <typename T, int STC, int NC, bool RES>
__global__ void __launch_bounds__(256, 2) k_layer1_lr_wide(LrArgs<T> A) {
    typedef typename Acc4<T>::type acc_t;
    constexpr int NB = 4, NCP = lr_ncp<NB, NC>();
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
    const int P = A.P, Nout = A.Nout, Kh = A.Kh, K0 = A.K0loc + A.K0sh, ntile = P / 16, nchunk = (ntile + STC - 1) / STC;
    extern __shared__ __attribute__((aligned(16))) char lrw_smem[];
    T* yl = reinterpret_cast<T*>(lrw_smem);
    T* Cl = yl + 2 * Kh + (size_t)wave * (16 * NB * NCP);
    const T* G1t = A.G1 + (size_t)w * A.g_ws + (size_t)tile * A.g_ts;
    {
        typedef T vec2 __attribute__((ext_vector_type(2)));
        const T* yo = A.YO + (size_t)w * A.yo_ws + (size_t)tile * 2 * Kh;
        for (int n = threadIdx.x; n < Kh; n += blockDim.x)
            *reinterpret_cast<vec2*>(yl + 2 * n) = *reinterpret_cast<const vec2*>(yo + 2 * n);
    }
    __syncthreads();
    if (n0 >= Nout) return;
    {
        // phase 1: C[m][c] = sum_n W1[n][m] B1[n][c]  (as in k_layer1_lr)
        acc_t c1[NB][NC];
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int s = 0; s < NC; ++s) c1[a][s] = acc_t{0, 0, 0, 0};
        T av[4][NB], wv[4][NC];
        const T* Wl = A.W1 + n0 + (size_t)lq * Nout + lr;
        const T* Tl = A.W0T + lq * (16 * NC) + lr;
        auto load_set = [&](int u) {
#pragma unroll
            for (int a = 0; a < NB; ++a) av[u][a] = Wl[16 * a];
#pragma unroll
            for (int s = 0; s < NC; ++s) wv[u][s] = Tl[16 * s];
            Wl += (size_t)4 * Nout;
            Tl += 4 * 16 * NC;
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
        };
        const int nks = Kh / 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) load_set(u);
        int ks = 0;
        for (; ks + 4 < nks; ks += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { step(u, ks + u); load_set(u); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) step(u, ks + u);
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int s = 0; s < NC; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) Cl[(16 * a + acc_row<T>(lane, r)) * NCP + 16 * s + lr] = c1[a][s][r];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const T* Ca = Cl + lr * NCP + lq;
    T* Got = A.Gout + (size_t)w * A.go_ws + (size_t)tile * A.go_ts;
    WideCarry<T> cy{0, 0, 0, 0};
    for (int c = 0; c < nchunk; ++c) {
        const int t0 = c * STC, cw = ntile - t0 < STC ? ntile - t0 : STC;
        const bool first = c == 0;
        const int co = 16 * t0;                      // first slot of the chunk (columns beyond the last tile read on into the next row: never stored)
        // (an opaque copy of the row stride: the row addresses of S, of the residual and of the output would otherwise be hoisted out of
        //  the chunk loop -- ~100 registers of loop-invariant pointers next to the accumulators)
        int Pl = P;
        asm volatile("" : "+s"(Pl));
        acc_t acc[NB][STC];
        {
            const T* Sp0 = A.S1 + (size_t)w * Nout * P + lr;
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + 16 * a + acc_row<T>(lane, r);
#pragma unroll
                    for (int s = 0; s < STC; ++s) acc[a][s][r] = Sp0[(size_t)n * Pl + (co + 16 * s)];
                }
        }
        {
            // phase 2 on the chunk's tiles: pair-mean rows (dense), then the K0 low-rank rows (A = C from LDS), one operand ring
            constexpr int NSET = wide_ring_sets<T>();
            T av[NSET][NB], bv[NSET][STC];
            const int nm2 = A.Km2 / 4, nl = A.K0loc / 4, nk = nm2 + K0 / 4;
            const T* Wm = A.W1 + (size_t)(Kh + lq) * Nout + n0 + lr;
            const T* Xm = G1t + (size_t)(Kh + lq) * Pl + lr;
            const T* Xl = A.XL + (size_t)w * A.xl_ws + (size_t)tile * A.xl_ts + (size_t)lq * Pl + lr;
            const T* Ml = A.M0 + (size_t)w * A.m0_ws + (size_t)lq * Pl + lr;
            int kl = 0;
            auto load_set = [&](int u) {
                const int km = kl < nm2 ? kl : nm2 - 1;
                const T* wp = Wm + (size_t)(4 * km) * Nout;
                const T* xp = kl < nm2 ? Xm + (size_t)(4 * kl) * Pl : (kl - nm2 < nl ? Xl + (size_t)(4 * (kl - nm2)) * Pl : Ml + (size_t)(4 * (kl - nm2 - nl)) * Pl);
#pragma unroll
                for (int a = 0; a < NB; ++a) av[u][a] = wp[16 * a];
#pragma unroll
                for (int s = 0; s < STC; ++s) bv[u][s] = xp[(co + 16 * s)];
                ++kl;
            };
            auto step = [&](int u, int k) {
                const bool low = k >= nm2;
                const int cc = low ? 4 * (k - nm2) : 0;
                T b0 = bv[u][0];
                b0 = (low && first && lr < 2) ? T(0) : b0;          // (slots 0 / 1 live in the first chunk's tile 0)
#pragma unroll
                for (int a = 0; a < NB; ++a) {
                    const T cl = Ca[16 * a * NCP + cc];
                    const T aa = low ? cl : av[u][a];
#pragma unroll
                    for (int s = 0; s < STC; ++s) acc[a][s] = mfma16(aa, s == 0 ? b0 : bv[u][s], acc[a][s]);
                }
            };
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
            if (first) {
                const T one = (lq == 0 && lr == 0) || (lq == 1 && lr == 1) ? T(1) : T(0);
#pragma unroll
                for (int a = 0; a < NB; ++a) acc[a][0] = mfma16(Ca[16 * a * NCP + K0], one, acc[a][0]);
            }
        }
        if constexpr (RES) {
            const T* Xl = A.XL + (size_t)w * A.xl_ws + (size_t)tile * A.xl_ts + (size_t)lq * Pl + lr;
            const T* S0p = A.S0 + (size_t)w * Kh * Pl + lr;
            auto rf = [&](int a) {
                const T rs2 = T(0.70710678118654752440);
                acc_t racc[STC];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int n = n0 + 16 * a + acc_row<T>(lane, rr);
#pragma unroll
                    for (int s = 0; s < STC; ++s) racc[s][rr] = S0p[(size_t)n * Pl + (co + 16 * s)];
                }
                for (int ks = 0; ks < A.K0loc / 4; ++ks) {
                    const T av = A.W0[(size_t)(4 * ks + lq) * Kh + n0 + 16 * a + lr];
#pragma unroll
                    for (int s = 0; s < STC; ++s) racc[s] = mfma16(av, Xl[(size_t)(4 * ks) * Pl + (co + 16 * s)], racc[s]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + 16 * a + acc_row<T>(lane, r);
                    typedef T vec2 __attribute__((ext_vector_type(2)));
                    const vec2 yo = *reinterpret_cast<const vec2*>(yl + 2 * n);
                    const T d1 = 1 - yo[0] * yo[0];
#pragma unroll
                    for (int s = 0; s < STC; ++s) {
                        T hv = d1 * racc[s][r];
                        const bool t00 = first && s == 0;
                        hv = (t00 && lr == 0) ? yo[0] : hv;
                        if (s < cw && !(t00 && lr == 1))      // (the Laplacian slot is written after the last chunk)
                            __builtin_nontemporal_store((hv + acc[a][s][r]) * rs2, &Got[(size_t)n * Pl + (co + 16 * s) + lr]);
                    }
                }
            };
            wide_layer_epilogue<T, NB, STC, 2, 0>(acc, (const T*)nullptr, Got, (const T*)nullptr, n0, lane, Pl, co, cw, first, cy, rf);
        } else
            wide_layer_epilogue<T, NB, STC, 1, 0>(acc, (const T*)nullptr, Got, (const T*)nullptr, n0, lane, Pl, co, cw, first, cy);
    }
    wide_layer_finish<T, NB, (RES ? 2 : 1)>(cy, Got, n0, lane, P, RES ? (const T*)yl : (const T*)nullptr);
}

template <typename T, int STC> inline size_t wide_stash_bytes(unsigned threads) {
    return (size_t)(threads / 64) * stash_blocks<T, 4, STC>() * 4 * STC * 64 * sizeof(T);
}

}  // namespace ds
