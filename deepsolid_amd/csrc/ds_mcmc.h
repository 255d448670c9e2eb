// ds_mcmc.h -- Metropolis walker update with an in-kernel counter-based generator (Philox4x32-10).
//
// Reference: DeepSolid/qmc.py:153-224 (mh_update, symmetric branch) inside the fori_loop of make_mcmc_step
// (:335-362).  JAX draws the noise from threefry keys split per step (:190-192, :217-218); here every draw is a
// pure function of (seed, offset + step, element index, stream), so a step needs no noise tensors, no generator
// state on the device and no host synchronisation: the whole `steps`-move loop is enqueued on one stream.
//   stream 0/1: the two Philox blocks that give the three normal deviates of one electron (Box-Muller, float64)
//   stream 2  : the uniform deviate of one walker's accept test
#pragma once
#include "ds_value.h"

namespace ds {

struct Philox4 { unsigned v[4]; };

__host__ __device__ __forceinline__ void philox_mulhilo(unsigned a, unsigned b, unsigned* hi, unsigned* lo) {
    const unsigned long long p = (unsigned long long)a * b;
    *hi = (unsigned)(p >> 32);
    *lo = (unsigned)p;
}

// Philox4x32 with 10 rounds (Salmon et al., SC'11): counter c[4], key k[2]
__host__ __device__ __forceinline__ Philox4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                                         unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned hi0, lo0, hi1, lo1;
        philox_mulhilo(0xD2511F53u, c0, &hi0, &lo0);
        philox_mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
        const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}

// 53-bit uniforms from two 32-bit words: [0, 1) and (0, 1]
__host__ __device__ __forceinline__ double u53_co(unsigned a, unsigned b) {
    return (double)((((unsigned long long)a << 21) ^ ((unsigned long long)b >> 11)) & ((1ull << 53) - 1)) * (1.0 / 9007199254740992.0);
}
__host__ __device__ __forceinline__ double u53_oc(unsigned a, unsigned b) {
    return (double)(((((unsigned long long)a << 21) ^ ((unsigned long long)b >> 11)) & ((1ull << 53) - 1)) + 1) * (1.0 / 9007199254740992.0);
}

struct PhiloxKey { unsigned long long seed, offset; };

// the three standard-normal deviates of electron `e` at move `step` (float64 Box-Muller on two Philox blocks)
__device__ __forceinline__ void philox_normal3(const PhiloxKey k, unsigned long long step, unsigned long long e, double z[3]) {
    const unsigned long long off = k.offset + step;
    const unsigned k0 = (unsigned)k.seed, k1 = (unsigned)(k.seed >> 32);
    const unsigned c0 = (unsigned)e, c1 = (unsigned)(e >> 32), c2 = (unsigned)off;
    const unsigned c3 = (unsigned)(off >> 32) & 0x3fffffffu;          // the two top bits select the stream
    const Philox4 a = philox4x32_10(c0, c1, c2, c3, k0, k1);
    const Philox4 b = philox4x32_10(c0, c1, c2, c3 | 0x40000000u, k0, k1);
    double s, c;
    const double ra = sqrt(-2.0 * log(u53_oc(a.v[0], a.v[1])));
    sincos(2.0 * DS_PI * u53_co(a.v[2], a.v[3]), &s, &c);
    z[0] = ra * c; z[1] = ra * s;
    const double rb = sqrt(-2.0 * log(u53_oc(b.v[0], b.v[1])));
    z[2] = rb * cos(2.0 * DS_PI * u53_co(b.v[2], b.v[3]));
}

__device__ __forceinline__ double philox_uniform(const PhiloxKey k, unsigned long long step, unsigned long long w) {
    const unsigned long long off = k.offset + step;
    const Philox4 a = philox4x32_10((unsigned)w, (unsigned)(w >> 32), (unsigned)off, ((unsigned)(off >> 32) & 0x3fffffffu) | 0x80000000u,
                                    (unsigned)k.seed, (unsigned)(k.seed >> 32));
    return u53_co(a.v[0], a.v[1]);
}

// x2 = wrap(x1 + width * N(0,1))      qmc.py:192-193; `normal` != nullptr replays caller-supplied noise (test mode)
template <typename T>
__global__ void k_mcmc_propose(const T* __restrict__ a, const T* __restrict__ ainv, const T* __restrict__ x1,
                               const T* __restrict__ normal, PhiloxKey key, unsigned long long step, T width, size_t n_elec,
                               T* __restrict__ x2) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elec) return;
    T r[3], o[3], wr[3];
    if (normal) {
        for (int c = 0; c < 3; ++c) r[c] = x1[3 * e + c] + width * normal[3 * e + c];
    } else {
        double z[3];
        philox_normal3(key, step, e, z);
        for (int c = 0; c < 3; ++c) r[c] = x1[3 * e + c] + width * (T)z[c];
    }
    wrap_point(r, a, ainv, o, wr);
    for (int c = 0; c < 3; ++c) x2[3 * e + c] = o[c];
}

// lp2 = 2 log|psi(x2)|; accept iff lp2 - lp1 > log u; select x, lp; count      qmc.py:195-196, 217-222
// one workgroup per walker; n_accept is an integer-valued count, so the order of the atomic adds does not matter
template <typename T>
__global__ void k_mcmc_accept(T* __restrict__ x1, T* __restrict__ lp1, const T* __restrict__ x2, const T* __restrict__ logabs2,
                              const T* __restrict__ uniform, PhiloxKey key, unsigned long long step, int n3, long w0,
                              T* __restrict__ n_accept) {
    const long w = blockIdx.x;
    const T lp2 = 2 * logabs2[w];
    const T u = uniform ? uniform[w] : (T)philox_uniform(key, step, (unsigned long long)(w0 + w));
    const bool cond = (lp2 - lp1[w]) > ds_log(u);
    if (cond)
        for (int c = threadIdx.x; c < n3; c += blockDim.x) x1[(size_t)w * n3 + c] = x2[(size_t)w * n3 + c];
    __syncthreads();
    if (threadIdx.x == 0 && cond) {
        lp1[w] = lp2;
        atomicAdd(n_accept, T(1));
    }
}

template <typename T> __global__ void k_scale2(const T* __restrict__ in, long n, T* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = 2 * in[i];
}

// Packed batch statistics of the local energy (train.py:74-82 and the n_nonfinite slot of SURVEY 8(e)):
//   out[0..7] = [ sum Re E_L, sum Im E_L, sum |E_L|^2, n, n_nonfinite, sum Re E_kin, sum Im E_kin, sum E_ewald ]
// E_L = ke + ewald.  One workgroup, fixed summation order (bit-reproducible), float64 accumulation for both dtypes.
// The sums run over ALL walkers (a NaN propagates exactly as in the reference's jnp.mean); n_nonfinite tells the
// driver to discard the step (process.py:303-318).
template <typename T>
__global__ void __launch_bounds__(256) k_energy_stats(const T* __restrict__ ke, const T* __restrict__ ew, long B, double* __restrict__ out) {
    __shared__ double sh[8][256];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long b = threadIdx.x; b < B; b += 256) {
        const double kr = (double)ke[2 * b], ki = (double)ke[2 * b + 1], e = (double)ew[b];
        const double re = kr + e;
        acc[0] += re; acc[1] += ki; acc[2] += re * re + ki * ki; acc[3] += 1.0;
        acc[4] += (isfinite(re) && isfinite(ki)) ? 0.0 : 1.0;
        acc[5] += kr; acc[6] += ki; acc[7] += e;
    }
    for (int j = 0; j < 8; ++j) sh[j][threadIdx.x] = acc[j];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int j = 0; j < 8; ++j) sh[j][threadIdx.x] += sh[j][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 8) out[threadIdx.x] = sh[threadIdx.x][0];
}

}  // namespace ds
