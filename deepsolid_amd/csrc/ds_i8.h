// ds_i8.h -- the dense hidden one-electron layer (network.py:517-533) with its per-electron contraction on the int8 matrix pipe.
//
//   Z[tile][n][slot] = sum_k W[k][n] X[tile][k][slot]      K = 64 NCH per-electron rows, 256 features, 80 jet slots (N = 24 ... 26)
//
// as a 47-bit truncating fixed-point split (an "Ozaki scheme" with the low plane products dropped: NOT error-free, see `error` below)
// on v_mfma_i32_16x16x64_i8 (16 cycles for 32768 operations; v_mfma_f64_16x16x4_f64 takes 64
// cycles for 2048), float64 in, float64 out, the layer epilogue of ds_gemm.h (shared term, tanh chain rule on the jets, residual)
// behind it:
//
//   operands   fixed point with FB = 47 fractional bits under ONE power-of-two scale per (tile, 64-row chunk, slot) column of X and
//              per output feature of W, cut into six balanced radix-256 digits: int8 planes, plane 0 most significant
//   products   digit planes (i, j) with i + j <= 5: 21 MFMA passes per chunk, int32 accumulation exact (|sum| < 2^23 per group)
//   recombine  per chunk and 16 x 16 output tile: the six group sums merged pairwise in int32 (acc_g 256 + acc_g+1 < 2^31), three
//              int -> float64 conversions and four FMAs per element, accumulated in float64 under the chunk's column scale
//   error      |dZ| <= ~1e-12 max|Z| per column (2^-46 of the column maxima of both operands, the dropped planes 2^-48);
//              E_kin moves by 4e-11 Ha on the benchmark cell (tools/i8split_accuracy.py; five planes: 3e-9, seven: 2e-13)
//
// Kernel shape (tools/probes/i8split_probe.hip is the standalone development version, profiles/r05_i8split_probe*.json its numbers):
// one persistent workgroup of 8 waves per CU walks electron tiles; wave w owns output features 32 w .. + 31 (two passes of 16) and all
// five slot tiles.  A chunk of the float64 input tile lands in LDS as it is (global -> LDS, 40 KB), is cut into digit planes for
// the NEXT chunk's products while the current chunk's products run (column exponents by LDS atomics on the rows each wave brought
// in; 320 (k quarter, slot) items convert 16 rows each), the weight digits stream through registers.  A wave's products are one
// stream of bursts -- 21 MFMAs on the six group accumulators of one output tile -- with the previous burst's recombination on the
// vector ALU in their shadow.  One chunk image serves all 256 features: the global -> LDS path sustains ~10 bytes per clock and CU.
#pragma once
#include "ds_gemm.h"

namespace ds {
namespace i8 {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int NPL = 6;                                  // digit planes
constexpr int FB = 47;                                  // fractional bits
constexpr int P = 80, ST = 5;                           // jet slots of the instance (5 tiles of 16)
constexpr int CH16 = NPL * 4 * P;                       // 16-byte pieces of a chunk's planes (1920 = 30720 bytes)
constexpr int CHI = (CH16 + P / 2 + 63) / 64 * 64;      // ... of a chunk image in LDS: planes + the 80 column scales (1984)
constexpr int NOUT = 256;
// cache policy of the raw-tile stream (global -> LDS); 2 = non-temporal measured no different (same fabric bytes, 17.7 against 17.6 ms)
#ifndef DS_I8_RAW_AUX
#define DS_I8_RAW_AUX 0
#endif
inline size_t lds_bytes() { return 2 * ((size_t)CHI * 16 + 64 * P * 8) + 2 * P * 4; }
inline size_t wp_bytes(int K) { return (size_t)K * NOUT * NPL; }

// x 2^(FB - e) rounded to an integer of <= 47 bits -> its six balanced digits as the bytes of (lo, hi): byte b of the 48-bit value
// is digit plane 5 - b (two's complement int8).  Adding 128 at every digit position turns the balanced digits into the unsigned
// base-256 digits of the sum; xor 0x80 maps each back to its int8 bit pattern.
__device__ __forceinline__ void digits6(double x, double scale, uint32_t& lo, uint32_t& hi) {
    const double t = fma(x, scale, 6755399441055744.0);                 // 1.5 * 2^52: the integer sits in the low mantissa bits
    const uint64_t bits = (uint64_t)__double_as_longlong(t);
    const uint64_t d = (bits - 0x4338000000000000ull + 0x0000808080808080ull) ^ 0x0000808080808080ull;
    lo = (uint32_t)d;
    hi = (uint32_t)(d >> 32);
}

// exponent e of a column from the largest high word of |x|: |x| < 2^(field - 1022) = 2^(e - 1), so |x| 2^-e < 0.5
__device__ __forceinline__ int col_exp(uint32_t hi_max) {
    const int field = (int)(hi_max >> 20);
    return max(field, 122) - 1021;
}

// ---- weights W[K][256] (row stride ldw) -> WP[chunk][plane][n / 16][k quarter][fragment row][16 bytes], SW[n] = 2^(f_n - 7).
// Fragment row rho = 4 lq + r hands its result to accumulator register r of lane group lq; feature n % 16 = lq + 4 r sits there,
// so that the output tile has the float64 MFMA's layout (acc_row<double>) and the float64 epilogues apply.
// One workgroup of 256 threads per 16 features: thread = (feature n % 16, k lane), k = k lane + 16 i.
__global__ void __launch_bounds__(256) k_i8_prep_w(const double* __restrict__ W, int K, int ldw, uint8_t* __restrict__ WP, double* __restrict__ SW) {
    __shared__ uint32_t mx[16];
    const int fr = threadIdx.x & 15, kl = threadIdx.x >> 4, n = 16 * blockIdx.x + fr;
    if (threadIdx.x < 16) mx[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mh = 0;
    for (int k = kl; k < K; k += 16) mh = max(mh, (uint32_t)(__double2hiint(W[(size_t)k * ldw + n]) & 0x7fffffff));
    atomicMax(&mx[fr], mh);
    __syncthreads();
    mh = mx[fr];
    const int f = col_exp(mh);
    if (kl == 0) SW[n] = (mh >> 20) == 0x7ff ? __longlong_as_double(0x7ff8000000000000ll) : __hiloint2double((1023 + f - 7) << 20, 0);
    const double sc = __hiloint2double((1023 + FB - f) << 20, 0);
    const int rho = 4 * (fr & 3) + (fr >> 2);
    for (int k = kl; k < K; k += 16) {
        uint32_t lo, hi;
        digits6(W[(size_t)k * ldw + n], sc, lo, hi);
        const int c = k >> 6, kq = (k >> 4) & 3, b = k & 15;
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            const int byte = 5 - p;
            const uint8_t v = byte < 4 ? (uint8_t)(lo >> (8 * byte)) : (uint8_t)(hi >> (8 * (byte - 4)));
            WP[(((((size_t)c * NPL + p) * (NOUT / 16) + blockIdx.x) * 4 + kq) * 16 + rho) * 16 + b] = v;
        }
    }
}

__device__ __forceinline__ v4i ld_frag(const uint4* p) {
    const uint4 t = *p;
    return v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
}

// 21 MFMAs on one 16 x 16 output tile; the B planes come from LDS one plane ahead of their products (bq = the tile's plane 0)
__device__ __forceinline__ void burst(const v4i (&a)[NPL], const uint4* bq, v4i (&acc)[NPL]) {
    v4i bf[2];
    bf[0] = ld_frag(bq);
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        if (j + 1 < NPL) bf[(j + 1) & 1] = ld_frag(bq + (j + 1) * 4 * P);
#pragma unroll
        for (int i = 0; i < NPL - j; ++i)
            acc[i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], bf[j & 1], j == 0 ? v4i{0, 0, 0, 0} : acc[i + j], 0, 0, 0);
    }
}

// z += sx (acc0 + 2^-8 acc1 + ... + 2^-40 acc5) 2^8, sx = 2^(e - 15) of the chunk's column
__device__ __forceinline__ void recombine(const v4i (&acc)[NPL], double sx, double (&z)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m1 = (acc[0][r] << 8) + acc[1][r];
        const int m2 = (acc[2][r] << 8) + acc[3][r];
        const int m3 = (acc[4][r] << 8) + acc[5][r];
        double u = (double)m1;
        u = fma((double)m2, 0x1p-16, u);
        u = fma((double)m3, 0x1p-32, u);
        z[r] = fma(u, sx, z[r]);
    }
}

// Column exponents of a chunk, phase A: every wave looks at the 8 raw rows IT brought in (rows 8 w .. 8 w + 7) and folds the high
// words of |x| into the chunk's 80 column maxima with LDS atomics.  Only the largest EXPONENT matters; a non-finite entry wins the
// maximum and turns the column's scale -- hence every output of the column -- into NaN.
__device__ __forceinline__ void chunk_maxima(const double* __restrict__ R, uint32_t* __restrict__ MXH, int wave, int lane) {
    const uint32_t* Rh = reinterpret_cast<const uint32_t*>(R) + 1;
    uint32_t m0 = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) m0 = max(m0, Rh[((8 * wave + k) * P + lane) * 2] & 0x7fffffffu);
    const int s1 = 64 + (lane & 15), k1 = 8 * wave + 2 * (lane >> 4);
    const uint32_t m1 = max(Rh[(k1 * P + s1) * 2] & 0x7fffffffu, Rh[((k1 + 1) * P + s1) * 2] & 0x7fffffffu);
    atomicMax(&MXH[lane], m0);
    atomicMax(&MXH[s1], m1);
}

// phase B: item = (k quarter, slot) -> the six 16-byte pieces of its 16 rows, and (k quarter 0) the column's scale 2^(e - 15)
__device__ __forceinline__ void slice_chunk(const double* __restrict__ R, const uint32_t* __restrict__ MXH, uint4* __restrict__ PL, int item) {
    const int kq = item / P, slot = item % P;
    const uint32_t mh = MXH[slot];
    const int e = col_exp(mh);
    if (kq == 0)
        reinterpret_cast<double*>(PL + CH16)[slot] = (mh >> 20) == 0x7ff ? __longlong_as_double(0x7ff8000000000000ll) : __hiloint2double((1023 + e - 15) << 20, 0);
    const double sc = __hiloint2double((1023 + FB - e) << 20, 0);
    uint32_t* PLw = reinterpret_cast<uint32_t*>(PL);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        double v[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) v[b] = R[(16 * kq + 8 * h + b) * P + slot];
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
            uint32_t lo[4], hi[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) digits6(v[4 * q2 + b], sc, lo[b], hi[b]);
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                // plane p = byte 5 - p of the 48-bit values: bytes 0..3 of lo, 0..1 of hi
                const uint32_t* src = p >= 2 ? lo : hi;
                const int byte = p >= 2 ? 5 - p : 1 - p;
                const uint32_t sel01 = 0x0c0c0000u | ((4 + byte) << 8) | byte;      // v_perm: (src1 byte, src0 byte) -> low half
                const uint32_t x01 = __builtin_amdgcn_perm(src[1], src[0], sel01);
                const uint32_t x23 = __builtin_amdgcn_perm(src[3], src[2], sel01);
                PLw[((p * 4 + kq) * P + slot) * 4 + 2 * h + q2] = x01 | (x23 << 16);      // (a dword at a time: no staging of the pieces)
            }
        }
    }
}

// The layer.  X: input tiles [tile][ldx rows][P] (the first 64 NCH rows are contracted, rows 0 .. 255 are the residual);
// Sb: shared term + bias [walker][256][P] (walker = tile / N); Gout: output tiles, same strides.  EPI = 2: residual layer
// (h_out = (h_in + tanh-chain(z)) / sqrt 2), EPI = 1: no residual, EPI = 4: the value chain's residual layer (h_out = (h_in + tanh z) / sqrt 2
// in every column; tiles = (80-walker group, electron), Sb per group).  grid = CUs (a multiple of 8 for the XCD-aware tile order; fewer when there are fewer tiles), block = 512, LDS = lds_bytes(); ntiles = walkers x N.
template <int NCH, int EPI>
__global__ void __launch_bounds__(512, 1) k_layer_i8(const double* __restrict__ X, size_t tile_stride, const uint4* __restrict__ WP,
                                                     const double* __restrict__ SW, const double* __restrict__ Sb, int N,
                                                     double* __restrict__ Gout, int ntiles) {
    extern __shared__ uint4 i8_smem[];
    uint4* const PLb = i8_smem;                                               // 2 x CHI pieces: planes + column scales
    double* const Rb = reinterpret_cast<double*>(i8_smem + 2 * CHI);          // 2 x (64 x 80) raw rows
    uint32_t* const MXb = reinterpret_cast<uint32_t*>(Rb + 2 * 64 * P);       // 2 x 80 column maxima (high words)
    constexpr int nf16 = NOUT / 16;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lq = lane >> 4, lr = lane & 15;
    // This workgroup's tiles.  Workgroup ids b, b + 8, ... share an XCD (its L2): XCD x = b % 8 takes the walkers w = x (mod 8) and
    // deals their electron tiles, in order, to its gridDim.x / 8 workgroups -- at any time the CUs of an XCD work on the ~1.3 walkers
    // whose shared term S (164 KB each) then sits in that L2 once.  (Tiles of a walker back to back on ONE workgroup, 40 us apart,
    // re-fetched S for every tile: 16 GB per launch on the fabric counters.)  Pure speed: any placement gives the same result.
    const int n_walkers = ntiles / N;
    const int nx = ((gridDim.x & 7) == 0 && n_walkers >= 16) ? 8 : 1, per = (int)gridDim.x / nx;      // (few walkers: plain tile order)
    const int xcd = (int)blockIdx.x % nx, jx = (int)blockIdx.x / nx;
    const int stream_tiles = ((n_walkers - xcd + nx - 1) / nx) * N;              // tiles of this XCD's walkers
    const int n_my = stream_tiles > jx ? (stream_tiles - jx + per - 1) / per : 0;
    auto tile_of = [&](int it) {
        const int t = it * per + jx;
        return ((t / N) * nx + xcd) * N + t % N;
    };
    auto a_ptr = [&](int c, int p, int pass) { return WP + ((((size_t)c * NPL + p) * nf16 + 2 * wave + pass) * 4 + lq) * 16 + lr; };
    auto stage = [&](int g) {                 // raw rows of chunk g of the stream -> R[g & 1]: wave w brings rows 8 w .. 8 w + 7 (5 x 1 KB)
        const int tile = tile_of(g / NCH), c = g % NCH;
        const uint4* src = reinterpret_cast<const uint4*>(X + (size_t)tile * tile_stride + (size_t)c * 64 * P) + lane;
        uint4* dst = reinterpret_cast<uint4*>(Rb + (g & 1) * 64 * P);
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int pc = 5 * wave + u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * pc),
                                             (__attribute__((address_space(3))) void*)(dst + 64 * pc), 16, 0, DS_I8_RAW_AUX);
        }
    };
    if (n_my <= 0) return;
    const int n_chunks = n_my * NCH;
    if (tid < 2 * P) MXb[tid] = 0;
    stage(0);
    if (n_chunks > 1) stage(1);
    v4i aw[2][NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) aw[0][p] = ld_frag(a_ptr(0, p, 0));
    v4i accs[2][NPL];
#pragma unroll
    for (int g = 0; g < NPL; ++g) accs[1][g] = v4i{0, 0, 0, 0};
    double sx_prev = 0;
    typedef typename Acc4<double>::type acc_t;
    acc_t zacc[2][ST];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int s = 0; s < ST; ++s) zacc[q][s] = acc_t{0, 0, 0, 0};
    auto rec = [&](const v4i (&acc)[NPL], double sx, acc_t& z) {
        double zz[4] = {z[0], z[1], z[2], z[3]};
        recombine(acc, sx, zz);
        z = acc_t{zz[0], zz[1], zz[2], zz[3]};
    };
    __syncthreads();                 // raw rows of chunks 0 and 1 have landed, the maxima are zero
    chunk_maxima(Rb, MXb, wave, lane);
    __syncthreads();
    if (tid < 4 * P) slice_chunk(Rb, MXb, PLb, tid);
    int g = 0;
#pragma unroll 1
    for (int it = 0; it < n_my; ++it) {
#pragma unroll 1
        for (int c = 0; c < NCH; ++c, ++g) {
            __syncthreads();         // planes of chunk g complete, raw rows of chunk g + 1 landed; planes (g + 1) & 1 and raw g & 1 are free
            if (g + 1 < n_chunks) {
                // chunk g + 1 -> digit planes: column maxima (every wave on the 8 rows it brought in), barrier, then 320 (k quarter, slot) items
                if (tid < P) MXb[(g & 1) * P + tid] = 0;    // (used up; collects for chunk g + 2 after the next chunk barrier)
                chunk_maxima(Rb + ((g + 1) & 1) * 64 * P, MXb + ((g + 1) & 1) * P, wave, lane);
                __syncthreads();
            }
            // (behind the second barrier: a barrier drains the wave's outstanding global -> LDS loads)
            if (g + 2 < n_chunks) stage(g + 2);
            if (g + 1 < n_chunks && tid < 4 * P) slice_chunk(Rb + ((g + 1) & 1) * 64 * P, MXb + ((g + 1) & 1) * P, PLb + ((g + 1) & 1) * CHI, tid);
            const uint4* PL = PLb + (g & 1) * CHI;
            const uint4* bp = PL + lq * P + lr;
            const double* sxp = reinterpret_cast<const double*>(PL + CH16) + lr;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                {
                    // the other pass's weight digits arrive while this pass computes: pass 1 of this chunk, then pass 0 of the next
                    const int cn = pass == 0 ? c : (c + 1 < NCH ? c + 1 : 0);
#pragma unroll
                    for (int p = 0; p < NPL; ++p) aw[pass ^ 1][p] = ld_frag(a_ptr(cn, p, pass ^ 1));
                }
#pragma unroll
                for (int t = 0; t < ST; ++t) {
                    const int b = pass * ST + t;
                    const double sx = sxp[16 * t];
                    burst(aw[pass], bp + 16 * t, accs[b & 1]);
                    // the previous burst's tile: (pass, t - 1), (0, 4) for b = 5, (1, 4) of the previous chunk for b = 0
                    rec(accs[(b & 1) ^ 1], sx_prev, zacc[b == 0 ? 1 : (t == 0 ? 0 : pass)][t == 0 ? ST - 1 : t - 1]);
                    sx_prev = sx;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // end of tile: flush the last burst; z = W x (the weight columns' scales) + S + b; the layer epilogue of ds_gemm.h
        rec(accs[1], sx_prev, zacc[1][ST - 1]);
        sx_prev = 0;
        const int tile = tile_of(it);
        const int n0 = 32 * wave;
        // (the lane id goes through an empty asm so that the epilogue's ~100 per-lane row offsets are recomputed per tile: hoisted out
        //  of the tile loop as invariants they do not fit the register file next to the main loop and come back from scratch memory
        //  one dependent load at a time)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int lr_e = lane_e & 15;
        const double* Sp = Sb + (size_t)(tile / N) * NOUT * P + lr_e;
        const double* Gi = X + (size_t)tile * tile_stride + lr_e;
        double* Go = Gout + (size_t)tile * tile_stride + lr_e;
        // The workgroup is alone on its CU and its waves reach this point together: nothing hides a load's latency here, and with
        // global -> LDS loads in flight every wait is a wait for everything, so a deeper software pipeline would buy nothing.
        // One half (16 rows) at a time: its 20 shared-term and 20 residual loads per lane go out together, one round trip per half
        // (the residual + store stage of layer_epilogue is handed in).  (Both halves' loads in one or two volleys with a
        // hand-written epilogue: 284-468 bytes of scratch per lane, 20.8 ms instead of 17.9.)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            acc_t zh[1][ST];
            double sv[4][ST], hv[4][ST], sw[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 16 * q + acc_row<double>(lane_e, r);
                sw[r] = SW[n];
#pragma unroll
                for (int s = 0; s < ST; ++s) {
                    sv[r][s] = Sp[(size_t)n * P + 16 * s];
                    if (EPI == 2) hv[r][s] = Gi[(size_t)n * P + 16 * s];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int s = 0; s < ST; ++s) zh[0][s][r] = fma(zacc[q][s][r], sw[r], sv[r][s]);
            if (EPI == 2) {
                // (residual + store stage of layer_epilogue: the residual rows were requested with the shared term above)
                auto rf = [&](int) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = n0 + 16 * q + acc_row<double>(lane_e, r);
#pragma unroll
                        for (int s = 0; s < ST; ++s)
                            __builtin_nontemporal_store((hv[r][s] + zh[0][s][r]) * 0.70710678118654752440, &Go[(size_t)n * P + 16 * s]);
                    }
                };
                layer_epilogue<double, 1, ST, 2, 0>(zh, (const double*)nullptr, Go, (const double*)nullptr, n0 + 16 * q, lane_e, P, rf);
            } else if (EPI == 4)     // value chain (the slot axis carries 80 walkers): tanh of every column + residual, rows fetched by the epilogue
                layer_epilogue<double, 1, ST, 4, 0>(zh, Gi, Go, (const double*)nullptr, n0 + 16 * q, lane_e, P);
            else
                layer_epilogue<double, 1, ST, 1, 0>(zh, (const double*)nullptr, Go, (const double*)nullptr, n0 + 16 * q, lane_e, P);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int s = 0; s < ST; ++s) zacc[q][s] = acc_t{0, 0, 0, 0};
    }
}

}  // namespace i8
}  // namespace ds
