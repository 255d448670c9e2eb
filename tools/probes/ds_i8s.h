// ds_i8s.h -- the int8-split dense hidden layer of csrc/ds_i8.h with TWO workgroups per CU (round 6).  NOT part of the library:
// parity-exact against the 8-wave kernel but slower (20.8 ms against 16.6 per 4096 bcc-Li walkers; EXPERIMENTS.md, *Round 6*); kept
// as a runnable probe (tools/probes/i8s_probe.hip, tools/i8s_probe.py) with its ablation switches (-DI8S_NO_BURST / _EPI / _STAGE / _CUT).
//
// The 8-wave kernel of ds_i8.h keeps the float64 accumulators of a whole electron tile (256 features x 80 slots: a third of the CU's
// register file) in ONE workgroup that is alone on its CU: nothing runs under its epilogue, its chunk barriers or its slicing, and
// its matrix pipe (38 % busy) and vector ALU (40 %) are busy at different times (profiles/r05_pmc_sq_counters.txt).  Here the work
// item is (electron tile, slot group): group A = slot tiles 0..2 (48 slots, among them the value and the Laplacian), group B = slot
// tiles 3, 4 (32 slots).  Slicing is per (chunk, slot) column, so the groups share nothing but the weights; a four-wave workgroup
// (wave = 64 features = 4 passes of 16, x 3 or 2 slot tiles: 96 / 64 accumulator registers) takes the two items of a tile back to
// back, and two such workgroups -- on different tiles, out of phase -- share a CU: one's epilogue, barrier waits and slicing run
// under the other's MFMA bursts.
//
//   LDS per workgroup (71 KB): digit planes of two chunks [2][plane][k quarter][slot][16 B] + their column scales, ONE raw chunk
//     (64 rows x 48 slots, global -> LDS), three sets of column maxima, a 4-number stash per feature that hands (y', y'', the
//     Laplacian so far, its residual) of a tile from item A to item B
//   raw rows are wave-private: wave w brings rows 16 w .. 16 w + 15 of a chunk (= k quarter w of its planes) and cuts exactly those,
//     one lane per slot: 16 rows -> one 16-byte piece per plane, ds_write_b128 (no bank conflicts; the 8-wave kernel wrote dwords at
//     a 16-byte lane stride: 40 % of its LDS cycles were conflicts).  Only the column maxima cross waves (LDS atomics, ordered by
//     the chunk barrier), so there is ONE barrier per chunk, and it is a bare s_barrier behind lgkmcnt(0): outstanding global -> LDS
//     loads (whose only reader is the issuing wave) are not drained
//   pipeline of chunk g: raw rows requested at the top of period g - 2 (behind that period's first weight loads, so that the weight wait
//     in front of the second pass does not wait for them), column maxima at the end of period g - 2, cut into planes at the top of
//     period g - 1 (behind the barrier that completes the maxima; the raw image is then free for chunk g + 1), multiplied in period g
//   the tanh chain rule couples the groups in one place: the Laplacian slot (group A) needs sum_d z_d^2 over ALL derivative slots.
//     Item A leaves its part of the sum, and item B -- same wave, same features, right behind it -- finishes and stores that one
//     column (LDS stash, wave-private: no barrier)
//
// Arithmetic of the products (digits6 / burst / recombine) and the weight planes (k_i8_prep_w) are those of ds_i8.h: the two
// kernels agree to the last bit in z; the epilogue adds the two halves of sum_d z_d^2 in another order (1 ulp in the Laplacian slot).
#pragma once
#include "../../deepsolid_amd/csrc/ds_i8.h"

namespace ds {
namespace i8 {

constexpr int NSA = 48, NSB = 32;                       // slots of item A (slot tiles 0..2) / item B (slot tiles 3, 4)
constexpr int S_PL16 = NPL * 4 * NSA;                   // 16-byte pieces of one plane buffer (sized for item A): 1152 = 18432 bytes
constexpr int S_PLANES = 0;                             // byte offsets of the LDS carve
constexpr int S_SCALES = S_PLANES + 2 * S_PL16 * 16;    // 2 x 48 doubles
constexpr int S_RAW = S_SCALES + 2 * NSA * 8;           // 64 x 48 doubles
constexpr int S_MX = S_RAW + 64 * NSA * 8;              // 3 x 64 uint32
constexpr int S_STASH = S_MX + 3 * 64 * 4;              // 256 x 4 doubles
constexpr int S_TOTAL = S_STASH + NOUT * 4 * 8;
inline size_t lds_bytes_split() { return S_TOTAL; }

#define DS_I8S_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define DS_I8S_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// workgroup barrier that orders LDS traffic only (no vmcnt drain: global -> LDS loads stay in flight across it)
#define DS_I8S_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// raw rows 16 w .. 16 w + 15 of chunk c, slots [slot0, slot0 + NS) of a tile -> the wave's part of the raw image: [16][NS] at
// 16 w NSA for BOTH item types -- with parts packed by the item's own NS, wave 1's part of a B chunk overlaps wave 0's part of the A
// chunk in front of it, and wave 1 requests its B rows while wave 0 may still be cutting (an intermittent wrong digit at item seams)
template <int NS>
__device__ __forceinline__ void stage_rows(const double* __restrict__ Xt, int c, int wave, int lane, double* __restrict__ raw) {
    constexpr int HP = NS / 2, NI = 16 * HP / 64;       // 16-byte pieces per row; instructions per wave (6 / 4)
    const double* src = Xt + (size_t)(64 * c + 16 * wave) * P;
    uint4* dst = reinterpret_cast<uint4*>(raw + 16 * wave * NSA);      // (the wave's part is the same for both item types: see SplitCtx::raw)
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int q = 64 * u + lane, row = q / HP, cp = q - row * HP;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + row * P + 2 * cp),
                                         (__attribute__((address_space(3))) void*)(dst + 64 * u), 16, 0, 0);
    }
}

// the largest high word of the wave's 16 rows of slot `lane` -> the chunk's column maxima
template <int NS>
__device__ __forceinline__ void max_rows(const double* __restrict__ raw, uint32_t* __restrict__ MX, int wave, int lane) {
    if (lane < NS) {
        const uint32_t* rh = reinterpret_cast<const uint32_t*>(raw + 16 * wave * NSA + lane) + 1;
        uint32_t m = 0;
#pragma unroll
        for (int b = 0; b < 16; ++b) m = max(m, rh[2 * b * NS] & 0x7fffffffu);
        atomicMax(&MX[lane], m);
    }
}

// the wave's 16 rows of slot `lane` -> the six 16-byte pieces (plane p, k quarter = wave, slot = lane) of the chunk's planes; wave 0
// also leaves the column scales 2^(e - 15).  (The rows are read from the raw image a second time: carried in registers from the maxima
// pass to here they are 32 registers that the allocator parks in scratch memory across the bursts -- and every scratch reload is a
// vmcnt wait, i.e. a wait for the raw rows in flight.)
template <int NS>
__device__ __forceinline__ void cut_rows(const double* __restrict__ raw, const uint32_t* __restrict__ MX, uint4* __restrict__ PL, double* __restrict__ SC,
                                         int wave, int lane) {
    if (lane < NS) {
        const uint32_t mh = MX[lane];
        const int e = col_exp(mh);
        if (wave == 0) SC[lane] = (mh >> 20) == 0x7ff ? __longlong_as_double(0x7ff8000000000000ll) : __hiloint2double((1023 + e - 15) << 20, 0);
        const double sc = __hiloint2double((1023 + FB - e) << 20, 0);
        const double* rv = raw + 16 * wave * NSA + lane;
        uint32_t w[NPL][4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            uint32_t lo[4], hi[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) digits6(rv[(4 * q4 + b) * NS], sc, lo[b], hi[b]);
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                // plane p = byte 5 - p of the 48-bit values: bytes 0..3 of lo, 0..1 of hi
                const uint32_t* src = p >= 2 ? lo : hi;
                const int byte = p >= 2 ? 5 - p : 1 - p;
                const uint32_t sel01 = 0x0c0c0000u | ((4 + byte) << 8) | byte;
                const uint32_t x01 = __builtin_amdgcn_perm(src[1], src[0], sel01);
                const uint32_t x23 = __builtin_amdgcn_perm(src[3], src[2], sel01);
                w[p][q4] = x01 | (x23 << 16);
            }
        }
#pragma unroll
        for (int p = 0; p < NPL; ++p) PL[(p * 4 + wave) * NS + lane] = uint4{w[p][0], w[p][1], w[p][2], w[p][3]};
    }
}

// 21 MFMAs on one 16 x 16 output tile (ds_i8.h::burst with the plane stride of this layout)
template <int NS>
__device__ __forceinline__ void burst_s(const v4i (&a)[NPL], const uint4* bq, v4i (&acc)[NPL]) {
    v4i bf[2];
    bf[0] = ld_frag(bq);
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        if (j + 1 < NPL) bf[(j + 1) & 1] = ld_frag(bq + (j + 1) * 4 * NS);
#pragma unroll
        for (int i = 0; i < NPL - j; ++i)
            acc[i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], bf[j & 1], j == 0 ? v4i{0, 0, 0, 0} : acc[i + j], 0, 0, 0);
    }
}

// Pipeline state of a workgroup's chunk stream (chunk g = 5 x item + c; item = 2 x tile index + (0: A, 1: B))
struct SplitCtx {
    const double* X; size_t tile_stride; const uint4* WP; const double* SW; const double* Sb; double* Gout;
    int N, n_chunks, wave, lane;
    uint4* PLb; double* SCb; double* raw; uint32_t* MXb; double* stash;
};

// One item: five chunks of bursts + the layer epilogue (EPI 2: h_out = (h_in + tanh-chain(z)) / sqrt 2) for its slot tiles.
//   NT = slot tiles of the item (3: A, 2: B); tile_of(it) -> electron tile of the workgroup's it-th tile.
template <int NT, typename TileOf>
__device__ __forceinline__ void run_item(const SplitCtx& C, int item, TileOf&& tile_of) {
    constexpr int NS = 16 * NT, T0 = NT == 3 ? 0 : 3;          // slots of this item, its first slot tile
    constexpr int nf16 = NOUT / 16, NCH = 5;
    typedef typename Acc4<double>::type acc_t;
    // (the lane id goes through an empty asm per item: per-lane addresses hoisted out of the item loop as invariants do not fit the
    //  register file next to the accumulators and come back from scratch memory)
    int lane = C.lane;
    asm volatile("" : "+v"(lane));
    const int wave = C.wave, lq = lane >> 4, lr = lane & 15;
    const int tile = tile_of(item >> 1);
    auto a_ptr = [&](int c, int p, int pass) { return C.WP + ((((size_t)c * NPL + p) * nf16 + 4 * wave + pass) * 4 + lq) * 16 + lr; };
    // item type / source of chunk gg of the stream (for the stages that run ahead of the products)
    auto chunk_tile = [&](int gg) { return tile_of((gg / NCH) >> 1); };
    auto chunk_is_a = [&](int gg) { return (((gg / NCH) & 1) == 0); };
    acc_t zacc[4][NT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < NT; ++s) zacc[q][s] = acc_t{0, 0, 0, 0};
    v4i aw[2][NPL], accs[2][NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) aw[0][p] = ld_frag(a_ptr(0, p, 0));
#pragma unroll
    for (int gI = 0; gI < NPL; ++gI) accs[1][gI] = v4i{0, 0, 0, 0};
    double sx_prev = 0;
    auto rec = [&](const v4i (&acc)[NPL], double sx, acc_t& z) {
        double zz[4] = {z[0], z[1], z[2], z[3]};
        recombine(acc, sx, zz);
        // (pinned here: left to itself the compiler sinks the whole recombination behind the chunk's last burst and keeps all twelve
        //  bursts' group sums alive -- 890 spilled registers)
        asm volatile("" : "+v"(zz[0]), "+v"(zz[1]), "+v"(zz[2]), "+v"(zz[3]));
        z = acc_t{zz[0], zz[1], zz[2], zz[3]};
    };
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        const int g = NCH * item + c;
        DS_I8S_BARRIER();            // planes of chunk g and the maxima of chunk g + 1 are complete; every wave has left chunk g - 1
        // pass 1's weight digits first: their wait (in front of pass 1) then does not wait for the raw rows requested next
#pragma unroll
        for (int p = 0; p < NPL; ++p) aw[1][p] = ld_frag(a_ptr(c, p, 1));
#ifdef I8S_NO_CUT
        if (g + 1 < 0) {
#else
        if (g + 1 < C.n_chunks) {    // chunk g + 1 -> digit planes (its rows sit in the raw image, its maxima are complete)
#endif
            uint4* PLn = C.PLb + ((g + 1) & 1) * S_PL16;
            double* SCn = C.SCb + ((g + 1) & 1) * NSA;
            const uint32_t* MXn = C.MXb + ((g + 1) % 3) * 64;
            if (chunk_is_a(g + 1)) cut_rows<NSA>(C.raw, MXn, PLn, SCn, wave, lane);
            else cut_rows<NSB>(C.raw, MXn, PLn, SCn, wave, lane);
            DS_I8S_WAIT_LGKM0();     // the raw image is free for the next request
        }
#ifdef I8S_NO_STAGE
        if (g + 2 < 0) {
#else
        if (g + 2 < C.n_chunks) {    // raw rows of chunk g + 2: they land under this period's bursts
#endif
            const double* Xt = C.X + (size_t)chunk_tile(g + 2) * C.tile_stride;
            if (chunk_is_a(g + 2)) stage_rows<NSA>(Xt, (g + 2) % NCH, wave, lane, C.raw);
            else stage_rows<NSB>(Xt + NSA, (g + 2) % NCH, wave, lane, C.raw);
        }
        C.MXb[(g % 3) * 64 + lane] = 0;      // (maxima of chunk g: used up; collects for chunk g + 3 from the end of the next period on)
        const uint4* PL = C.PLb + (g & 1) * S_PL16;
        const uint4* bp = PL + lq * NS + lr;
        const double* sxp = C.SCb + (g & 1) * NSA + lr;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            if (pass >= 1) {
                // the next pass's weight digits (pass 0 of the next chunk behind pass 3; branch-free -- a branch here splits the chunk's
                // bursts into two blocks -- so the item's last chunk re-reads chunk 0)
                const int cn = pass < 3 ? c : (c + 1 < NCH ? c + 1 : 0);
#pragma unroll
                for (int p = 0; p < NPL; ++p) aw[(pass + 1) & 1][p] = ld_frag(a_ptr(cn, p, (pass + 1) & 3));
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#ifdef I8S_NO_BURST
                if (C.n_chunks > 0) continue;
#endif
                const int b = pass * NT + t;
                const double sx = sxp[16 * t];
                burst_s<NS>(aw[pass & 1], bp + 16 * t, accs[b & 1]);
                // the previous burst's tile: (pass, t - 1), (pass - 1, NT - 1) for t = 0, (3, NT - 1) of the previous chunk for b = 0
                rec(accs[(b & 1) ^ 1], sx_prev, zacc[t == 0 ? (pass + 3) & 3 : pass][t == 0 ? NT - 1 : t - 1]);
                sx_prev = sx;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (c + 1 < NCH) {
            // column maxima of chunk g + 2 (its rows were requested at the top of this period) into the third set
            if (g + 2 < C.n_chunks) {
                DS_I8S_WAIT_VM0();
                uint32_t* MXl = C.MXb + ((g + 2) % 3) * 64;
                if (chunk_is_a(g + 2)) max_rows<NSA>(C.raw, MXl, wave, lane);
                else max_rows<NSB>(C.raw, MXl, wave, lane);
            }
        }
    }
    // end of the item: flush the last burst
    rec(accs[1], sx_prev, zacc[3][NT - 1]);
    // ---- epilogue: z = W x (weight column scales) + S + b; tanh chain rule on the jets; residual; store
#ifdef I8S_NO_EPI
    if (zacc[0][0][0] == 1.2345 || zacc[1][1][1] == 1.2345 || zacc[2][0][2] == 1.2345 || zacc[3][1][3] == 1.2345)
#endif
    {
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));       // (per-lane row offsets recomputed here, not hoisted out of the item loop)
        const int lr_e = lane_e & 15, lq_e = lane_e >> 4;
        const int n0 = 64 * wave;
        const double* Sp = C.Sb + (size_t)(tile / C.N) * NOUT * P + 16 * T0 + lr_e;
        const double* Gi = C.X + (size_t)tile * C.tile_stride + 16 * T0 + lr_e;
        double* Go = C.Gout + (size_t)tile * C.tile_stride + 16 * T0 + lr_e;
        const double rs2 = 0.70710678118654752440;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_sched_barrier(0);      // (one 16-row block at a time: its 2 x 4 x NT loads go out together, the next block's stay behind)
            double sv[4][NT], hv[4][NT], sw[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 16 * q + lq_e + 4 * r;
                sw[r] = C.SW[n];
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    sv[r][s] = Sp[(size_t)n * P + 16 * s];
                    hv[r][s] = Gi[(size_t)n * P + 16 * s];
                }
            }
            double z[4][NT];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int s = 0; s < NT; ++s) z[r][s] = fma(zacc[q][s][r], sw[r], sv[r][s]);
            if (NT == 3) {
                // item A: value slot 0, Laplacian slot 1.  tanh of the four value slots in ONE evaluation (lane lr = r takes row r)
                double zsel = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double vv = row16_bcast<0>(z[r][0]);
                    zsel = lr_e == r ? vv : zsel;
                }
                const double yall = ds_tanh(zsel);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + 16 * q + lq_e + 4 * r;
                    double ss = 0;
#pragma unroll
                    for (int s = 0; s < NT; ++s)
                        if (s > 0 || lr_e >= 2) ss += z[r][s] * z[r][s];
                    ss = row16_sum(ss);
                    const double zL = row16_bcast<1>(z[r][0]);
                    const double y = row16_bcast_dyn<4>(yall, r), d1 = 1 - y * y, d2 = -2 * y * d1;
                    const double lpart = d1 * zL + d2 * ss;
                    if (lr_e == 1) {
                        // handed to item B of this tile (same wave): y', y'', the Laplacian without B's derivative slots, its residual
                        double* st = C.stash + 4 * n;
                        st[0] = d1; st[1] = d2; st[2] = lpart; st[3] = hv[r][0];
                    }
#pragma unroll
                    for (int s = 0; s < NT; ++s) {
                        double o = d1 * z[r][s];
                        if (s == 0 && lr_e == 0) o = y;
                        if (!(s == 0 && lr_e == 1)) __builtin_nontemporal_store((hv[r][s] + o) * rs2, &Go[(size_t)n * P + 16 * s]);
                    }
                }
            } else {
                // item B: derivative slots only; finishes the Laplacian column
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + 16 * q + lq_e + 4 * r;
                    const double* st = C.stash + 4 * n;
                    const double d1 = st[0], d2 = st[1];
                    double ss = 0;
#pragma unroll
                    for (int s = 0; s < NT; ++s) ss += z[r][s] * z[r][s];
                    ss = row16_sum(ss);
#pragma unroll
                    for (int s = 0; s < NT; ++s) __builtin_nontemporal_store((hv[r][s] + d1 * z[r][s]) * rs2, &Go[(size_t)n * P + 16 * s]);
                    if (lr_e == 1) {
                        const double lap = st[2] + d2 * ss;
                        __builtin_nontemporal_store((st[3] + lap) * rs2, C.Gout + (size_t)tile * C.tile_stride + (size_t)n * P + 1);
                    }
                }
            }
        }
    }
    // the last chunk's look-ahead stages, behind the epilogue (their registers are free again)
    {
        const int g = NCH * item + NCH - 1;
        if (g + 2 < C.n_chunks) {
            DS_I8S_WAIT_VM0();
            uint32_t* MXl = C.MXb + ((g + 2) % 3) * 64;
            if (chunk_is_a(g + 2)) max_rows<NSA>(C.raw, MXl, wave, lane);
            else max_rows<NSB>(C.raw, MXl, wave, lane);
        }
    }
}

// The layer (EPI 2: residual layer of the forward-Laplacian chain).  Arguments as k_layer_i8; grid = 2 x CUs (a multiple of 8 for the
// XCD-aware tile order; fewer when there are fewer tiles), block = 256, LDS = lds_bytes_split().
__global__ void __launch_bounds__(256, 2) k_layer_i8_split(const double* __restrict__ X, size_t tile_stride, const uint4* __restrict__ WP,
                                                            const double* __restrict__ SW, const double* __restrict__ Sb, int N,
                                                            double* __restrict__ Gout, int ntiles) {
    extern __shared__ uint4 i8s_smem[];
    char* sm = reinterpret_cast<char*>(i8s_smem);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // tiles of this workgroup: as in k_layer_i8 (workgroup ids b, b + 8, ... share an XCD: the walkers w = x (mod 8) of XCD x are dealt,
    // electron tile by electron tile, to its workgroups, so that a walker's shared term sits in that L2 once)
    const int n_walkers = ntiles / N;
    const int nx = ((gridDim.x & 7) == 0 && n_walkers >= 16) ? 8 : 1, per = (int)gridDim.x / nx;
    const int xcd = (int)blockIdx.x % nx, jx = (int)blockIdx.x / nx;
    const int stream_tiles = ((n_walkers - xcd + nx - 1) / nx) * N;
    const int n_my = stream_tiles > jx ? (stream_tiles - jx + per - 1) / per : 0;
    if (n_my <= 0) return;
    auto tile_of = [&](int it) {
        const int t = it * per + jx;
        return ((t / N) * nx + xcd) * N + t % N;
    };
    SplitCtx C;
    C.X = X; C.tile_stride = tile_stride; C.WP = WP; C.SW = SW; C.Sb = Sb; C.Gout = Gout; C.N = N;
    C.n_chunks = n_my * 2 * 5; C.wave = wave; C.lane = lane;
    C.PLb = reinterpret_cast<uint4*>(sm + S_PLANES);
    C.SCb = reinterpret_cast<double*>(sm + S_SCALES);
    C.raw = reinterpret_cast<double*>(sm + S_RAW);
    C.MXb = reinterpret_cast<uint32_t*>(sm + S_MX);
    C.stash = reinterpret_cast<double*>(sm + S_STASH);
    if (tid < 3 * 64) C.MXb[tid] = 0;
    // prologue: chunk 0 -> planes, chunk 1 -> raw image + maxima (the first item is an A item)
    {
        const double* Xt = X + (size_t)tile_of(0) * tile_stride;
        __syncthreads();                                   // maxima zeroed
        stage_rows<NSA>(Xt, 0, wave, lane, C.raw);
        DS_I8S_WAIT_VM0();
        max_rows<NSA>(C.raw, C.MXb, wave, lane);
        DS_I8S_BARRIER();                                  // maxima of chunk 0 complete
        cut_rows<NSA>(C.raw, C.MXb, C.PLb, C.SCb, wave, lane);
        DS_I8S_WAIT_LGKM0();
        stage_rows<NSA>(Xt, 1, wave, lane, C.raw);
        DS_I8S_WAIT_VM0();
        max_rows<NSA>(C.raw, C.MXb + 64, wave, lane);
        // (period 0 cuts chunk 1, requests chunk 2 and zeroes maxima set 0 again, behind its barrier)
    }
#pragma unroll 1
    for (int it = 0; it < n_my; ++it) {
        run_item<3>(C, 2 * it, tile_of);
        run_item<2>(C, 2 * it + 1, tile_of);
    }
}

}  // namespace i8
}  // namespace ds
