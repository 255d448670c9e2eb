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

typedef int v4i __attribute__((ext_vector_type(4)));

namespace {

constexpr int NPL = 6;            // digit planes
constexpr int FB = 47;            // fractional bits
constexpr int P = 80, ST = 5;     // jet slots (5 tiles of 16)
constexpr int CH16 = NPL * 4 * P; // 16-byte pieces per chunk (1920 = 30720 bytes)

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
            WP[(((((size_t)c * NPL + p) * (Nout / 16) + (n >> 4)) * 4 + kq) * 16 + (n & 15)) * 16 + b] = v;
        }
    }
    (void)nch;
}

// ---- jets: X[tile][ldk rows][P] -> XP[tile][chunk][plane][k quarter][slot][16 bytes], XS[tile][chunk][slot] = 2^(e - 15)
// one workgroup of 320 threads per (tile, chunk): thread = (k quarter, slot)
__global__ void __launch_bounds__(320) k_slice(const double* __restrict__ X, size_t tile_stride, int nch, uint4* __restrict__ XP,
                                               double* __restrict__ XS) {
    __shared__ double mx[4][P];
    const int c = blockIdx.x, tile = blockIdx.y;
    const int kq = threadIdx.x / P, slot = threadIdx.x % P;
    const double* xp = X + (size_t)tile * tile_stride + (size_t)(64 * c + 16 * kq) * P + slot;
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
    if (kq == 0) XS[((size_t)tile * nch + c) * P + slot] = ldexp(1.0, e - 15);
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
        XP[((((size_t)tile * nch + c) * NPL + p) * 4 + kq) * P + slot] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// ---- the contraction.  grid.x = ntiles * Nout / 64 (workgroup ids b, b + 8, ... share an XCD: the Nout / 64 feature blocks of one
// electron tile are dealt to one XCD back to back, so the tile's planes are fetched from memory once)
template <int S0, int NS>
__device__ __forceinline__ void chunk_half(const v4i (&a)[NPL], const uint4* buf, int lq, int lr, const double* __restrict__ xs,
                                           double (&zacc)[ST][4]) {
    v4i acc[NPL][NS];
#pragma unroll
    for (int g = 0; g < NPL; ++g)
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[g][s] = v4i{0, 0, 0, 0};
    double sx[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) sx[s] = xs[16 * (S0 + s) + lr];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        v4i bf[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint4 t = buf[(j * 4 + lq) * P + 16 * (S0 + s) + lr];
            bf[s] = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
        }
#pragma unroll
        for (int i = 0; i < NPL - j; ++i)
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[i + j][s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], bf[s], acc[i + j][s], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m1 = (acc[0][s][r] << 8) + acc[1][s][r];
            const int m2 = (acc[2][s][r] << 8) + acc[3][s][r];
            const int m3 = (acc[4][s][r] << 8) + acc[5][s][r];
            double u = (double)m1;
            u = fma((double)m2, 0x1p-16, u);
            u = fma((double)m3, 0x1p-32, u);
            zacc[S0 + s][r] = fma(u, sx[s], zacc[S0 + s][r]);
        }
}

template <int NCH>
__global__ void __launch_bounds__(256, 2) k_i8_gemm(const uint4* __restrict__ XP, const double* __restrict__ XS, const uint4* __restrict__ WP,
                                                    const double* __restrict__ SW, double* __restrict__ Z, int ntiles, int Nout) {
    extern __shared__ uint4 smem[];
    const int nfb = Nout / 64, nf16 = Nout / 16;
    int tile, fb;
    {
        const unsigned b = blockIdx.x;
        if ((ntiles & 7) == 0) {
            const unsigned x = b & 7, q = b >> 3;
            tile = (int)(q / nfb) * 8 + (int)x;
            fb = (int)(q % nfb);
        } else {
            tile = (int)(b / nfb);
            fb = (int)(b % nfb);
        }
    }
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lq = lane >> 4, lr = lane & 15;
    const int f16 = fb * 4 + wave;
    const uint4* xp = XP + (size_t)tile * NCH * CH16;
    const double* xs = XS + (size_t)tile * NCH * P;
    auto a_ptr = [&](int c, int p) { return WP + ((((size_t)c * NPL + p) * nf16 + f16) * 4 + lq) * 16 + lr; };
    for (int i = tid; i < CH16; i += 256) smem[i] = xp[i];
    v4i a[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        const uint4 t = *a_ptr(0, p);
        a[p] = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
    }
    double zacc[ST][4];
#pragma unroll
    for (int s = 0; s < ST; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) zacc[s][r] = 0;
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        __syncthreads();
        v4i an[NPL];
        const bool more = c + 1 < NCH;
        if (more) {
            // next chunk: global -> LDS without passing registers (1 KB per wave instruction; the destination is wave-uniform
            // base + lane x 16, which is the planes' own order); it has landed before the barrier that opens chunk c + 1
            const uint4* src = xp + (size_t)(c + 1) * CH16 + lane;
            uint4* dst = smem + ((c + 1) & 1) * CH16;
            for (int pc = wave; pc < CH16 / 64; pc += 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 64 * pc),
                                                 (__attribute__((address_space(3))) void*)(dst + 64 * pc), 16, 0, 0);
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                const uint4 t = *a_ptr(c + 1, p);
                an[p] = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
            }
        }
        const uint4* buf = smem + (c & 1) * CH16;
        chunk_half<0, 3>(a, buf, lq, lr, xs + c * P, zacc);
        chunk_half<3, 2>(a, buf, lq, lr, xs + c * P, zacc);
        if (more) {
#pragma unroll
            for (int p = 0; p < NPL; ++p) a[p] = an[p];
        }
    }
    // rows n = 16 f16 + 4 lq + r, slots 16 s + lr
    double* zp = Z + ((size_t)tile * Nout + 16 * f16 + 4 * lq) * P + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double sw = SW[16 * f16 + 4 * lq + r];
#pragma unroll
        for (int s = 0; s < ST; ++s) __builtin_nontemporal_store(zacc[s][r] * sw, &zp[(size_t)r * P + 16 * s]);
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

}  // namespace

extern "C" {

// bytes of the digit planes / scales of n tiles with K rows
int64_t i8p_xp_bytes(int64_t ntiles, int K) { return ntiles * (K / 64) * (int64_t)CH16 * 16; }
int64_t i8p_xs_bytes(int64_t ntiles, int K) { return ntiles * (K / 64) * (int64_t)P * 8; }
int64_t i8p_wp_bytes(int K, int Nout) { return (int64_t)K * Nout * NPL; }

int i8p_prep_w(const double* W, int K, int Nout, void* WP, double* SW, void* stream) {
    if (K % 64 || Nout % 64) return 1;
    hipLaunchKernelGGL(k_prep_w, dim3((Nout + 63) / 64), dim3(64), 0, (hipStream_t)stream, W, K, Nout, (uint8_t*)WP, SW);
    return hipGetLastError() != hipSuccess;
}

int i8p_slice(const double* X, int64_t ntiles, int64_t tile_stride, int K, void* XP, double* XS, void* stream) {
    if (K % 64) return 1;
    hipLaunchKernelGGL(k_slice, dim3(K / 64, (unsigned)ntiles), dim3(320), 0, (hipStream_t)stream, X, (size_t)tile_stride, K / 64, (uint4*)XP, XS);
    return hipGetLastError() != hipSuccess;
}

int i8p_gemm(const void* XP, const double* XS, const void* WP, const double* SW, double* Z, int64_t ntiles, int K, int Nout, void* stream) {
    if (Nout % 64) return 1;
    const dim3 grid((unsigned)(ntiles * (Nout / 64))), block(256);
    const size_t sh = 2 * (size_t)CH16 * 16;
    hipStream_t st = (hipStream_t)stream;
    if (K == 320) hipLaunchKernelGGL(k_i8_gemm<5>, grid, block, sh, st, (const uint4*)XP, XS, (const uint4*)WP, SW, Z, (int)ntiles, Nout);
    else if (K == 256) hipLaunchKernelGGL(k_i8_gemm<4>, grid, block, sh, st, (const uint4*)XP, XS, (const uint4*)WP, SW, Z, (int)ntiles, Nout);
    else return 1;
    return hipGetLastError() != hipSuccess;
}

int i8p_ref(const double* X, int64_t ntiles, int64_t tile_stride, const double* W, int K, int Nout, double* Z, double* ZA, void* stream) {
    hipLaunchKernelGGL(k_ref, dim3(Nout, (unsigned)ntiles), dim3(128), 0, (hipStream_t)stream, X, (size_t)tile_stride, W, K, Nout, Z, ZA);
    return hipGetLastError() != hipSuccess;
}

}  // extern "C"
