// i8split_probe.hip -- round-5 gated probe: the dense hidden-layer contraction of the forward-Laplacian chain
//     Z[tile][n][slot] = sum_k W[k][n] X[tile][k][slot]        (K = 320, 256 features, 80 jet slots, 24 x 4096 electron tiles)
// as an error-free split on the int8 matrix pipe (v_mfma_i32_16x16x64_i8) instead of v_mfma_f64_16x16x4_f64.
//
//   operands   fixed point with 47 fractional bits under one power-of-two scale per (tile, 64-row chunk, slot) column of X and per
//              output feature of W, cut into six balanced radix-256 digits (int8 planes, plane 0 most significant)
//   products   digit planes (i, j) with i + j <= 5 (0-based): 21 int8 MFMA passes per 64-row chunk, exact in int32
//   recombine  per chunk: groups g = i + j merged pairwise in int32 (acc_g * 256 + acc_g+1 < 2^31), three int -> f64 conversions and
//              four FMAs per output element, accumulated in float64 under the chunk's column scale
//
// Kernel shape (k_i8_gemm): workgroup = 4 waves = one electron tile x 64 output features, wave = 16 features x all 80 slots, two
// workgroups per CU (<= 256 registers, 60 KB of LDS each).  The tile's digit planes stream through LDS one 64-row chunk at a time
// (double-buffered: 2 x 30 KB), the wave's weight digits (6 x 1 KB per chunk) through registers, one chunk ahead.
//
// Built as a shared library for tools/i8probe.py (ctypes); not part of the product library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

typedef int v4i __attribute__((ext_vector_type(4)));

namespace {

constexpr int NPL = 6;            // digit planes
constexpr int FB = 47;            // fractional bits
constexpr int P = 80, ST = 5;     // jet slots (5 tiles of 16)
constexpr int CH16 = NPL * 4 * P; // 16-byte pieces of a chunk's planes (1920 = 30720 bytes)
constexpr int CHI = (CH16 + P / 2 + 63) / 64 * 64;   // ... of a chunk image: planes + 80 column scales, in whole 1 KB pieces (1984)

__device__ __forceinline__ int exp_for(double m) {
    // e with |x| 2^-e <= 0.5 for every |x| <= m
    if (!(m > 0)) return 0;
    int ex;
    frexp(m, &ex);             // m = fr 2^ex, fr in [0.5, 1)
    return ex + 1;
}

// x 2^(FB - e) rounded to an integer of <= 47 bits -> the six balanced digits as bytes of (lo, hi): byte b of the 48-bit value is
// digit plane 5 - b (two's complement int8)
__device__ __forceinline__ void digits6(double x, double scale, uint32_t& lo, uint32_t& hi) {
    const double t = fma(x, scale, 6755399441055744.0);                 // 1.5 * 2^52: the integer sits in the low mantissa bits
    const uint64_t bits = (uint64_t)__double_as_longlong(t);
    const uint64_t xb = bits - 0x4338000000000000ull + 0x0000808080808080ull;      // + 128 at every digit: unsigned base-256 digits
    const uint64_t d = xb ^ 0x0000808080808080ull;                                    // digit - 128 as int8 bit patterns
    lo = (uint32_t)d;
    hi = (uint32_t)(d >> 32);
}

// ---- weights: W[K][Nout] -> WP[chunk][plane][n / 16][k quarter][n % 16][16 bytes], SW[n] = 2^(f_n - 7)
__global__ void k_prep_w(const double* __restrict__ W, int K, int Nout, uint8_t* __restrict__ WP, double* __restrict__ SW) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Nout) return;
    double m = 0;
    for (int k = 0; k < K; ++k) m = fmax(m, fabs(W[(size_t)k * Nout + n]));
    const int f = exp_for(m);
    SW[n] = ldexp(1.0, f - 7);
    const double sc = ldexp(1.0, FB - f);
    const int nch = K / 64;
    for (int k = 0; k < K; ++k) {
        uint32_t lo, hi;
        digits6(W[(size_t)k * Nout + n], sc, lo, hi);
        const int c = k >> 6, kq = (k >> 4) & 3, b = k & 15;
        for (int p = 0; p < NPL; ++p) {
            const int byte = 5 - p;
            const uint8_t v = byte < 4 ? (uint8_t)(lo >> (8 * byte)) : (uint8_t)(hi >> (8 * (byte - 4)));
            // fragment row rho = 4 lq + r hands its result to accumulator register r of lane group lq; feature n % 16 = lq + 4 r sits
            // there, so that the output tile has the float64 MFMA's layout (ds_device.h acc_row<double>) and its epilogues apply
            const int f = n & 15, rho = 4 * (f & 3) + (f >> 2);
            WP[(((((size_t)c * NPL + p) * (Nout / 16) + (n >> 4)) * 4 + kq) * 16 + rho) * 16 + b] = v;
        }
    }
    (void)nch;
}

// ---- jets: X[tile][ldk rows][P] -> chunk images XP[tile][chunk]{[plane][k quarter][slot][16 bytes] | scales[slot] = 2^(e - 15) | pad}
// (CHI 16-byte pieces per chunk: planes, then the 80 column scales as doubles, padded to whole 1 KB wave pieces)
// one workgroup of 320 threads per (tile, chunk): thread = (k quarter, slot)
__global__ void __launch_bounds__(320) k_slice(const double* __restrict__ X, size_t tile_stride, int nch, uint4* __restrict__ XP) {
    __shared__ double mx[4][P];
    const int c = blockIdx.x, tile = blockIdx.y;
    const int kq = threadIdx.x / P, slot = threadIdx.x % P;
    const double* xp = X + (size_t)tile * tile_stride + (size_t)(64 * c + 16 * kq) * P + slot;
    uint4* img = XP + ((size_t)tile * nch + c) * CHI;
    double v[16], m = 0;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        v[b] = xp[(size_t)b * P];
        m = fmax(m, fabs(v[b]));
    }
    mx[kq][slot] = m;
    __syncthreads();
    m = fmax(fmax(mx[0][slot], mx[1][slot]), fmax(mx[2][slot], mx[3][slot]));
    const int e = exp_for(m);
    if (kq == 0) reinterpret_cast<double*>(img + CH16)[slot] = ldexp(1.0, e - 15);
    const double sc = ldexp(1.0, FB - e);
    uint32_t lo[16], hi[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) digits6(v[b], sc, lo[b], hi[b]);
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        const int byte = 5 - p;
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t r = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t src = byte < 4 ? (lo[4 * q + b] >> (8 * byte)) : (hi[4 * q + b] >> (8 * (byte - 4)));
                r |= (src & 0xffu) << (8 * b);
            }
            w[q] = r;
        }
        img[(p * 4 + kq) * P + slot] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---- the contraction (second form).  One persistent workgroup of 8 waves per CU walks electron tiles t = blockIdx.x, + gridDim.x, ...;
// wave w owns output features 16 w .. + 15 and 128 + 16 w .. + 15 (two passes over every chunk, so that one chunk image in LDS
// serves all 256 features of the tile: the global -> LDS path sustains ~10 bytes per clock and CU, a quarter of what 64 features per
// image would ask for).  The work of a wave is one stream of bursts (chunk, pass, slot tile): 21 MFMAs on the six group accumulators
// of ONE 16 x 16 output tile, while the previous burst's accumulators are recombined into float64 on the vector ALU.
// MODE (timing experiments, wrong results): 1 no stores, 2 no global -> LDS traffic after the first chunk, 8 no recombination,
// 16 no MFMA
__device__ __forceinline__ v4i ld_frag(const uint4* p) {
    const uint4 t = *p;
    return v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
}

// 21 MFMAs on one 16 x 16 output tile; the B planes come from LDS one plane ahead of their products (bq = the tile's plane 0)
template <int MODE>
__device__ __forceinline__ void burst(const v4i (&a)[NPL], const uint4* bq, v4i (&acc)[NPL]) {
    v4i bf[2];
    bf[0] = ld_frag(bq);
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        if (j + 1 < NPL) bf[(j + 1) & 1] = ld_frag(bq + (j + 1) * 4 * P);
#pragma unroll
        for (int i = 0; i < NPL - j; ++i) {
            if (MODE & 16) { acc[i + j] = j == 0 ? a[i] + bf[j & 1] : acc[i + j] + bf[j & 1]; continue; }
            acc[i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], bf[j & 1], j == 0 ? v4i{0, 0, 0, 0} : acc[i + j], 0, 0, 0);
        }
    }
}

template <int MODE>
__device__ __forceinline__ void recombine(const v4i (&acc)[NPL], double sx, double (&z)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (MODE & 8) { z[r] += (double)(acc[0][r] ^ acc[1][r] ^ acc[2][r] ^ acc[3][r] ^ acc[4][r] ^ acc[5][r]) * sx; continue; }
        const int m1 = (acc[0][r] << 8) + acc[1][r];
        const int m2 = (acc[2][r] << 8) + acc[3][r];
        const int m3 = (acc[4][r] << 8) + acc[5][r];
        double u = (double)m1;
        u = fma((double)m2, 0x1p-16, u);
        u = fma((double)m3, 0x1p-32, u);
        z[r] = fma(u, sx, z[r]);
    }
}

template <int NCH, int MODE>
__global__ void __launch_bounds__(512, 1) k_i8_gemm(const uint4* __restrict__ XP, const uint4* __restrict__ WP, const double* __restrict__ SW,
                                                    double* __restrict__ Z, int ntiles, int Nout) {
    extern __shared__ uint4 smem[];
    const int nf16 = Nout / 16;                  // (= 16: two passes of 8 waves)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lq = lane >> 4, lr = lane & 15;
    const int n_my = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto a_ptr = [&](int c, int p, int pass) { return WP + ((((size_t)c * NPL + p) * nf16 + wave + 8 * pass) * 4 + lq) * 16 + lr; };
    // chunk g of this workgroup's stream -> LDS buffer g & 1 (1 KB per wave instruction, destination = wave-uniform base + lane x 16)
    auto stage = [&](int g) {
        const int tile = (int)blockIdx.x + (g / NCH) * (int)gridDim.x, c = g % NCH;
        const uint4* src = XP + ((size_t)tile * NCH + c) * CHI + lane;
        uint4* dst = smem + (g & 1) * CHI;
#pragma unroll
        for (int u = 0; u < (CHI / 64 + 7) / 8; ++u) {
            const int pc = wave + 8 * u;
            if (pc < CHI / 64)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * pc),
                                                 (__attribute__((address_space(3))) void*)(dst + 64 * pc), 16, 0, 0);
        }
    };
    if (n_my <= 0) return;
    stage(0);
    v4i aw[2][NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) aw[0][p] = ld_frag(a_ptr(0, p, 0));
    v4i accs[2][NPL];
#pragma unroll
    for (int g = 0; g < NPL; ++g) accs[1][g] = v4i{0, 0, 0, 0};
    double sx_prev = 0;
    double zacc[2][ST][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int s = 0; s < ST; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) zacc[q][s][r] = 0;
    const int n_chunks = n_my * NCH;
    int g = 0;
#pragma unroll 1
    for (int it = 0; it < n_my; ++it) {
#pragma unroll 1
        for (int c = 0; c < NCH; ++c, ++g) {
            __syncthreads();         // chunk g has landed (vmcnt drained in front of the barrier), buffer (g + 1) & 1 is free
            if (g + 1 < n_chunks && !(MODE & 2)) stage(g + 1);
            const uint4* buf = smem + ((MODE & 2) ? 0 : (g & 1)) * CHI;
            const uint4* bp = buf + lq * P + lr;
            const double* sxp = reinterpret_cast<const double*>(buf + CH16) + lr;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                // the other pass's weight digits arrive while this pass computes: pass 1 of this chunk, then pass 0 of the next
                {
                    const int cn = pass == 0 ? c : (c + 1 < NCH ? c + 1 : 0);
#pragma unroll
                    for (int p = 0; p < NPL; ++p) aw[pass ^ 1][p] = ld_frag(a_ptr(cn, p, pass ^ 1));
                }
#pragma unroll
                for (int t = 0; t < ST; ++t) {
                    const int b = pass * ST + t;
                    const double sx = sxp[16 * t];
                    burst<MODE>(aw[pass], bp + 16 * t, accs[b & 1]);
                    // the previous burst's tile: (pass, t - 1), (0, 4) for b = 5, (1, 4) of the previous chunk for b = 0
                    recombine<MODE>(accs[(b & 1) ^ 1], sx_prev, zacc[b == 0 ? 1 : (t == 0 ? 0 : pass)][t == 0 ? ST - 1 : t - 1]);
                    sx_prev = sx;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // end of tile: flush the last burst, scale by the weight columns' factors, store, start over
        recombine<MODE>(accs[1], sx_prev, zacc[1][ST - 1]);
        sx_prev = 0;
        const int tile = (int)blockIdx.x + it * (int)gridDim.x;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int n = 16 * (wave + 8 * q) + lq;             // rows n + 4 r, slots 16 s + lr
            double* zp = Z + ((size_t)tile * Nout + n) * P + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double sw = SW[n + 4 * r];
#pragma unroll
                for (int s = 0; s < ST; ++s) {
                    const double o = zacc[q][s][r] * sw;
                    if (!(MODE & 1) || o == 1.2345e-300) __builtin_nontemporal_store(o, &zp[(size_t)(4 * r) * P + 16 * s]);
                    zacc[q][s][r] = 0;
                }
            }
        }
    }
}

// ---- third form: the layer kernel's own input.  The float64 tile is the input (no pre-sliced planes in memory): every 64-row
// chunk lands in LDS as it is (global -> LDS, 40 KB, requested one chunk before it is cut), is cut into digit planes for the NEXT
// chunk's bursts while the current chunk's bursts run, the weight digits stream through registers.  Wave w owns output features
// 32 w .. + 31 (two passes of 16), rows placed like the float64 MFMA's accumulator (lane group + 4 x register).
//
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
    const int field = (int)(MXH[slot] >> 20);
    const int e = max(field, 122) - 1021;             // |x| < 2^(field - 1022) = 2^(e - 1): |x| 2^-e < 0.5
    if (kq == 0)
        reinterpret_cast<double*>(PL + CH16)[slot] = field == 0x7ff ? __longlong_as_double(0x7ff8000000000000ll) : __hiloint2double((1023 + e - 15) << 20, 0);
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

template <int NCH, int MODE>
__global__ void __launch_bounds__(512, 1) k_i8_layer(const double* __restrict__ X, size_t tile_stride, const uint4* __restrict__ WP,
                                                     const double* __restrict__ SW, double* __restrict__ Z, int ntiles, int Nout,
                                                     unsigned long long* clk) {
    // MODE & 32: wave 0 and wave 5 of every workgroup add their shader cycles per phase to clk[8 * (wave != 0) + ...]:
    // 0 total, 1 100 MHz ticks, 2 waiting at the chunk's barrier, 3 slicing (both phases and their barrier), 4 bursts, 5 tile epilogue
    long long tk0 = 0, tr0 = 0, t_bar = 0, t_slice = 0, t_burst = 0, t_epi = 0, ts = 0;
    auto tick = [&](long long& acc) { if (MODE & 32) { const long long n = clock64(); acc += n - ts; ts = n; } };
    if (MODE & 32) { tk0 = clock64(); tr0 = wall_clock64(); ts = tk0; }
    extern __shared__ uint4 smem[];
    uint4* const PLb = smem;                                                  // 2 x CHI pieces: planes + column scales
    double* const Rb = reinterpret_cast<double*>(smem + 2 * CHI);             // 2 x (64 x 80) raw rows
    uint32_t* const MXb = reinterpret_cast<uint32_t*>(Rb + 2 * 64 * P);       // 2 x 80 column maxima (high words)
    const int nf16 = Nout / 16;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lq = lane >> 4, lr = lane & 15;
    const int n_my = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto a_ptr = [&](int c, int p, int pass) { return WP + ((((size_t)c * NPL + p) * nf16 + 2 * wave + pass) * 4 + lq) * 16 + lr; };
    auto stage = [&](int g) {                 // raw rows of chunk g of the stream -> R[g & 1]: wave w brings rows 8 w .. 8 w + 7 (5 x 1 KB)
        const int tile = (int)blockIdx.x + (g / NCH) * (int)gridDim.x, c = g % NCH;
        const uint4* src = reinterpret_cast<const uint4*>(X + (size_t)tile * tile_stride + (size_t)c * 64 * P) + lane;
        uint4* dst = reinterpret_cast<uint4*>(Rb + (g & 1) * 64 * P);
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int pc = 5 * wave + u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * pc),
                                             (__attribute__((address_space(3))) void*)(dst + 64 * pc), 16, 0, 0);
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
    double zacc[2][ST][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int s = 0; s < ST; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) zacc[q][s][r] = 0;
    __syncthreads();                 // raw rows of chunks 0 and 1 have landed, the maxima are zero
    chunk_maxima(Rb, MXb, wave, lane);
    __syncthreads();
    if (tid < 4 * P) slice_chunk(Rb, MXb, PLb, tid);
    int g = 0;
#pragma unroll 1
    for (int it = 0; it < n_my; ++it) {
#pragma unroll 1
        for (int c = 0; c < NCH; ++c, ++g) {
            tick(t_burst);
            __syncthreads();         // planes of chunk g complete, raw rows of chunk g + 1 landed; planes (g + 1) & 1 and raw g & 1 are free
            tick(t_bar);
            if (g + 1 < n_chunks && !(MODE & 4)) {
                // chunk g + 1 -> digit planes: column maxima (every wave on the 8 rows it brought in), barrier, then 320 (k quarter, slot) items
                if (tid < P) MXb[(g & 1) * P + tid] = 0;    // (used up; collects for chunk g + 2 after the next chunk barrier)
                chunk_maxima(Rb + ((g + 1) & 1) * 64 * P, MXb + ((g + 1) & 1) * P, wave, lane);
                __syncthreads();
            }
            // (behind the second barrier: a barrier drains the wave's outstanding global -> LDS loads)
            if (g + 2 < n_chunks && !(MODE & 2)) stage(g + 2);
            if (g + 1 < n_chunks && !(MODE & 4)) {
                if (tid < 4 * P) slice_chunk(Rb + ((g + 1) & 1) * 64 * P, MXb + ((g + 1) & 1) * P, PLb + ((g + 1) & 1) * CHI, tid);
                tick(t_slice);
            }
            const uint4* PL = PLb + ((MODE & 4) ? 0 : (g & 1)) * CHI;
            const uint4* bp = PL + lq * P + lr;
            const double* sxp = reinterpret_cast<const double*>(PL + CH16) + lr;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                {
                    const int cn = pass == 0 ? c : (c + 1 < NCH ? c + 1 : 0);
#pragma unroll
                    for (int p = 0; p < NPL; ++p) aw[pass ^ 1][p] = ld_frag(a_ptr(cn, p, pass ^ 1));
                }
#pragma unroll
                for (int t = 0; t < ST; ++t) {
                    const int b = pass * ST + t;
                    const double sx = sxp[16 * t];
                    burst<MODE>(aw[pass], bp + 16 * t, accs[b & 1]);
                    recombine<MODE>(accs[(b & 1) ^ 1], sx_prev, zacc[b == 0 ? 1 : (t == 0 ? 0 : pass)][t == 0 ? ST - 1 : t - 1]);
                    sx_prev = sx;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        tick(t_burst);
        recombine<MODE>(accs[1], sx_prev, zacc[1][ST - 1]);
        sx_prev = 0;
        const int tile = (int)blockIdx.x + it * (int)gridDim.x;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int n = 16 * (2 * wave + q) + lq;             // rows n + 4 r, slots 16 s + lr
            double* zp = Z + ((size_t)tile * Nout + n) * P + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double sw = SW[n + 4 * r];
#pragma unroll
                for (int s = 0; s < ST; ++s) {
                    const double o = zacc[q][s][r] * sw;
                    if (!(MODE & 1) || o == 1.2345e-300) __builtin_nontemporal_store(o, &zp[(size_t)(4 * r) * P + 16 * s]);
                    zacc[q][s][r] = 0;
                }
            }
        }
        tick(t_epi);
    }
    if ((MODE & 32) && lane == 0 && (wave == 0 || wave == 5)) {
        unsigned long long* o = clk + (wave == 0 ? 0 : 8);
        atomicAdd(&o[0], (unsigned long long)(clock64() - tk0));
        atomicAdd(&o[1], (unsigned long long)(wall_clock64() - tr0));
        atomicAdd(&o[2], (unsigned long long)t_bar);
        atomicAdd(&o[3], (unsigned long long)t_slice);
        atomicAdd(&o[4], (unsigned long long)t_burst);
        atomicAdd(&o[5], (unsigned long long)t_epi);
    }
}

// ---- float64 reference of a few tiles (plain loop): Zref[tile][n][slot]
__global__ void k_ref(const double* __restrict__ X, size_t tile_stride, const double* __restrict__ W, int K, int Nout, double* __restrict__ Z,
                      double* __restrict__ ZA) {
    const int tile = blockIdx.y, n = blockIdx.x, slot = threadIdx.x;
    if (slot >= P) return;
    double s = 0, sa = 0;
    for (int k = 0; k < K; ++k) {
        const double w = W[(size_t)k * Nout + n], x = X[(size_t)tile * tile_stride + (size_t)k * P + slot];
        s = fma(w, x, s);
        sa = fma(fabs(w), fabs(x), sa);
    }
    Z[((size_t)tile * Nout + n) * P + slot] = s;
    ZA[((size_t)tile * Nout + n) * P + slot] = sa;
}

// ---- rate of v_mfma_i32_16x16x64_i8 by itself: every wave issues `iters` rounds of NACC independent MFMAs (no memory traffic);
// out[0] += shader cycles, out[1] += 100 MHz ticks of wave 0 of every workgroup
template <int NACC>
__global__ void __launch_bounds__(512, 1) k_i8_rate(int iters, unsigned long long* out, int* sink) {
    v4i acc[NACC];
    const v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = v4i{0, 0, 0, 0};
    const long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
    }
    const long long c1 = clock64(), r1 = wall_clock64();
    int t = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][3];
    if (t == 0x7fffffff) sink[0] = t;
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], (unsigned long long)(c1 - c0));
        atomicAdd(&out[1], (unsigned long long)(r1 - r0));
    }
}

// ---- the 21-product burst by itself (operands in registers, no memory): ORDER 0 = groups ascending within a round (the kernel's
// source order), 1 = descending (the longest chain first), 2 = two independent tiles interleaved
template <int ORDER>
__global__ void __launch_bounds__(512, 1) k_i8_burst_rate(int iters, unsigned long long* out, int* sink) {
    v4i a[NPL], b[NPL], acc[2][NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        a[i] = v4i{(int)threadIdx.x, i, 2, 3};
        b[i] = v4i{4, 5, (int)blockIdx.x, i};
        acc[0][i] = acc[1][i] = v4i{0, 0, 0, 0};
    }
    int t = 0;
    const long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            if (ORDER == 0) {
#pragma unroll
                for (int i = 0; i < NPL - j; ++i)
                    acc[0][i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], j == 0 ? v4i{0, 0, 0, 0} : acc[0][i + j], 0, 0, 0);
            } else if (ORDER == 1) {
#pragma unroll
                for (int i = NPL - j - 1; i >= 0; --i)
                    acc[0][i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], j == 0 ? v4i{0, 0, 0, 0} : acc[0][i + j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = NPL - j - 1; i >= 0; --i) {
                    acc[0][i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], j == 0 ? v4i{0, 0, 0, 0} : acc[0][i + j], 0, 0, 0);
                    acc[1][i + j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b[i], a[j], j == 0 ? v4i{0, 0, 0, 0} : acc[1][i + j], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NPL; ++i) t ^= acc[0][i][0] ^ (ORDER == 2 ? acc[1][i][1] : 0);
        a[0][1] = t & 1;
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long c1 = clock64(), r1 = wall_clock64();
    if (t == 0x7fffffff) sink[0] = t;
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], (unsigned long long)(c1 - c0));
        atomicAdd(&out[1], (unsigned long long)(r1 - r0));
    }
}

}  // namespace

extern "C" {

// bytes of the chunk images of n tiles with K rows
int64_t i8p_xp_bytes(int64_t ntiles, int K) { return ntiles * (K / 64) * (int64_t)CHI * 16; }
int64_t i8p_wp_bytes(int K, int Nout) { return (int64_t)K * Nout * NPL; }

int i8p_prep_w(const double* W, int K, int Nout, void* WP, double* SW, void* stream) {
    if (K % 64 || Nout % 64) return 1;
    hipLaunchKernelGGL(k_prep_w, dim3((Nout + 63) / 64), dim3(64), 0, (hipStream_t)stream, W, K, Nout, (uint8_t*)WP, SW);
    return hipGetLastError() != hipSuccess;
}

int i8p_slice(const double* X, int64_t ntiles, int64_t tile_stride, int K, void* XP, void* stream) {
    if (K % 64) return 1;
    hipLaunchKernelGGL(k_slice, dim3(K / 64, (unsigned)ntiles), dim3(320), 0, (hipStream_t)stream, X, (size_t)tile_stride, K / 64, (uint4*)XP);
    return hipGetLastError() != hipSuccess;
}

int i8p_gemm(const void* XP, const void* WP, const double* SW, double* Z, int64_t ntiles, int K, int Nout, int mode, void* stream) {
    if (Nout != 256 || K != 320) return 1;
    const dim3 grid((unsigned)std::min<int64_t>(ntiles, 256)), block(512);
    const size_t sh = 2 * (size_t)CHI * 16;
    hipStream_t st = (hipStream_t)stream;
#define I8P_GO(M) case M: hipLaunchKernelGGL((k_i8_gemm<5, M>), grid, block, sh, st, (const uint4*)XP, (const uint4*)WP, SW, Z, (int)ntiles, Nout); break
    switch (mode) {
        I8P_GO(0); I8P_GO(1); I8P_GO(2); I8P_GO(3); I8P_GO(8); I8P_GO(11); I8P_GO(16); I8P_GO(19); I8P_GO(27);
        default: return 1;
    }
#undef I8P_GO
    return hipGetLastError() != hipSuccess;
}

int i8p_rate(int blocks, int threads, int iters, int nacc, unsigned long long* out, int* sink, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (nacc == 4) hipLaunchKernelGGL(k_i8_rate<4>, dim3(blocks), dim3(threads), 0, st, iters, out, sink);
    else if (nacc == 8) hipLaunchKernelGGL(k_i8_rate<8>, dim3(blocks), dim3(threads), 0, st, iters, out, sink);
    else if (nacc == 16) hipLaunchKernelGGL(k_i8_rate<16>, dim3(blocks), dim3(threads), 0, st, iters, out, sink);
    else return 1;
    return hipGetLastError() != hipSuccess;
}

int i8p_layer(const double* X, int64_t tile_stride, const void* WP, const double* SW, double* Z, int64_t ntiles, int K, int Nout, int mode,
              unsigned long long* clk, void* stream) {
    if (Nout != 256 || K != 320) return 1;
    const dim3 grid((unsigned)std::min<int64_t>(ntiles, 256)), block(512);
    const size_t sh = 2 * ((size_t)CHI * 16 + 64 * P * 8) + 2 * P * 4;
    hipStream_t st = (hipStream_t)stream;
#define I8P_GO(M) case M: hipLaunchKernelGGL((k_i8_layer<5, M>), grid, block, sh, st, X, (size_t)tile_stride, (const uint4*)WP, SW, Z, (int)ntiles, Nout, clk); break
    switch (mode) {
        I8P_GO(0); I8P_GO(1); I8P_GO(2); I8P_GO(3); I8P_GO(4); I8P_GO(7); I8P_GO(8); I8P_GO(15); I8P_GO(16); I8P_GO(32); I8P_GO(39); I8P_GO(47);
        default: return 1;
    }
#undef I8P_GO
    return hipGetLastError() != hipSuccess;
}

int i8p_burst_rate(int blocks, int threads, int iters, int order, unsigned long long* out, int* sink, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (order == 0) hipLaunchKernelGGL(k_i8_burst_rate<0>, dim3(blocks), dim3(threads), 0, st, iters, out, sink);
    else if (order == 1) hipLaunchKernelGGL(k_i8_burst_rate<1>, dim3(blocks), dim3(threads), 0, st, iters, out, sink);
    else if (order == 2) hipLaunchKernelGGL(k_i8_burst_rate<2>, dim3(blocks), dim3(threads), 0, st, iters, out, sink);
    else return 1;
    return hipGetLastError() != hipSuccess;
}

int i8p_ref(const double* X, int64_t ntiles, int64_t tile_stride, const double* W, int K, int Nout, double* Z, double* ZA, void* stream) {
    hipLaunchKernelGGL(k_ref, dim3(Nout, (unsigned)ntiles), dim3(128), 0, (hipStream_t)stream, X, (size_t)tile_stride, W, K, Nout, Z, ZA);
    return hipGetLastError() != hipSuccess;
}

}  // extern "C"
