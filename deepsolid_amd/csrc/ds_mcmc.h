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

// x2 = wrap(x1 + width * N(0,1))      qmc.py:192-193; `normal` != nullptr replays caller-supplied noise (test mode).
// only >= 0: one-electron move (qmc.py:266-271): electron `only` of every walker gets the step (explicit noise is then
// (B,3)), the others are only wrapped, as the reference's enforce_pbc on the whole configuration does.
template <typename T>
__global__ void k_mcmc_propose(const T* __restrict__ a, const T* __restrict__ ainv, const T* __restrict__ x1,
                               const T* __restrict__ normal, PhiloxKey key, unsigned long long step, T width, size_t n_elec,
                               T* __restrict__ x2, int N = 0, int only = -1) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elec) return;
    T r[3], o[3], wr[3];
    const bool moved = only < 0 || (int)(e % (size_t)N) == only;
    if (!moved) {
        for (int c = 0; c < 3; ++c) r[c] = x1[3 * e + c];
    } else if (normal) {
        const size_t ni = only < 0 ? e : e / (size_t)N;
        for (int c = 0; c < 3; ++c) r[c] = x1[3 * e + c] + width * normal[3 * ni + c];
    } else {
        double z[3];
        philox_normal3(key, step, e, z);
        for (int c = 0; c < 3; ++c) r[c] = x1[3 * e + c] + width * (T)z[c];
    }
    wrap_point(r, a, ainv, o, wr);
    for (int c = 0; c < 3; ++c) x2[3 * e + c] = o[c];
}

// lp2 = 2 log|psi(x2)|; accept iff lp2 - lp1 > log u; select x, lp; count      qmc.py:195-196, 217-222
// 64 walkers per workgroup; n_accept is an integer-valued count, so the order of the atomic adds does not matter
template <typename T>
__global__ void __launch_bounds__(256) k_mcmc_accept(T* __restrict__ x1, T* __restrict__ lp1, const T* __restrict__ x2,
                                                     const T* __restrict__ logabs2, const T* __restrict__ uniform, PhiloxKey key,
                                                     unsigned long long step, int n3, long w0, long B, T* __restrict__ n_accept) {
    // 64 walkers per workgroup: the first wave takes the 64 decisions and counts the accepted ones with a ballot -- ONE atomic
    // per workgroup (a workgroup per walker meant ~0.9 B atomics on the same word per move: 48 us at 4096 walkers, most of this
    // kernel) --, then the four waves copy the accepted walkers.  The count is a sum of exact small integers: order-independent.
    __shared__ unsigned long long mask_s;
    const long wb = (long)blockIdx.x * 64;
    if (threadIdx.x < 64) {
        const long w = wb + threadIdx.x;
        bool cond = false;
        if (w < B) {
            const T lp2 = 2 * logabs2[w];
            const T u = uniform ? uniform[w] : (T)philox_uniform(key, step, (unsigned long long)(w0 + w));
            cond = (lp2 - lp1[w]) > ds_log(u);
            if (cond) lp1[w] = lp2;
        }
        const unsigned long long m = __ballot(cond);
        if (threadIdx.x == 0) {
            mask_s = m;
            if (m) atomicAdd(n_accept, T(__popcll(m)));
        }
    }
    __syncthreads();
    const unsigned long long m = mask_s;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = wave; k < 64; k += 4)
        if ((m >> k) & 1ull)
            for (int c = lane; c < n3; c += 64) x1[(size_t)(wb + k) * n3 + c] = x2[(size_t)(wb + k) * n3 + c];
}

// ---------------------------------------------------------------------------------------------------------------
// The reference's other proposals (flagged "untested", base_config.py:122-126), per move, noise supplied by the caller:
//   mode 1  asymmetric all-electron move (qmc.py:197-215): step width scaled per electron by the harmonic mean of its
//           (non-periodic) distances to the nuclei `aux1` (A,3); forward / reverse proposal densities in the test
//   mode 2  drift-biased importance move (qmc.py:83-124): x2 = x1 + w N + w^2 limdrift(grad log|psi|), aux1 / aux2 =
//           gradients (B,3N) at x1 / x2.  limdrift (qmc.py:63-81) divides g by clip(|g|, cutoff, max over the batch):
//           k_max_norm3 supplies the batch maximum (it is the binding bound when every drift is below the cutoff).
template <typename T> __device__ __forceinline__ T harmonic_mean_dist(const T x[3], const T* __restrict__ atoms, int n_atoms) {
    T s = 0;
    for (int a = 0; a < n_atoms; ++a) {
        const T dx = x[0] - atoms[3 * a], dy = x[1] - atoms[3 * a + 1], dz = x[2] - atoms[3 * a + 2];
        s += T(1) / ds_sqrt(dx * dx + dy * dy + dz * dz);
    }
    return T(1) / (s / T(n_atoms));                          // qmc.py:58-60
}

// cutoff / clip(|g|, a_min = cutoff, a_max = gmax) with numpy's clip = min(max(x, a_min), a_max): when every drift of the
// batch is below the cutoff the upper bound (the batch maximum, qmc.py:78) wins and all drifts are scaled by 1 / gmax
template <typename T> __device__ __forceinline__ T limdrift_factor(const T g[3], T gmax) {
    const T tot = ds_sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    const T lo = tot > T(1) ? tot : T(1);                    // cutoff = 1
    return T(1) / (lo < gmax ? lo : gmax);
}

// out[0] = max over the n_elec electrons of |g_e| (one workgroup; the maximum does not depend on the order)
template <typename T>
__global__ void __launch_bounds__(1024) k_max_norm3(const T* __restrict__ g, size_t n_elec, T* __restrict__ out) {
    __shared__ T sh[1024];
    T m = 0;
    for (size_t e = threadIdx.x; e < n_elec; e += 1024) {
        const T t = ds_sqrt(g[3 * e] * g[3 * e] + g[3 * e + 1] * g[3 * e + 1] + g[3 * e + 2] * g[3 * e + 2]);
        m = t > m ? t : m;
    }
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int s2 = 512; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) sh[threadIdx.x] = sh[threadIdx.x + s2] > sh[threadIdx.x] ? sh[threadIdx.x + s2] : sh[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

template <typename T>
__global__ void k_mh_propose_ex(const T* __restrict__ a, const T* __restrict__ ainv, int mode, const T* __restrict__ x1,
                                const T* __restrict__ normal, T width, const T* __restrict__ aux, int n_aux, size_t n_elec,
                                T* __restrict__ x2, const T* __restrict__ gmax) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elec) return;
    T p[3], r[3], o[3], wr[3];
    for (int c = 0; c < 3; ++c) p[c] = x1[3 * e + c];
    if (mode == 1) {
        const T sc = width * harmonic_mean_dist(p, aux, n_aux);                      // qmc.py:200-203
        for (int c = 0; c < 3; ++c) r[c] = p[c] + sc * normal[3 * e + c];
    } else {
        T g[3] = {aux[3 * e], aux[3 * e + 1], aux[3 * e + 2]};
        const T f = limdrift_factor(g, gmax[0]);
        for (int c = 0; c < 3; ++c) r[c] = p[c] + width * normal[3 * e + c] + width * width * f * g[c];   // qmc.py:112-114
    }
    wrap_point(r, a, ainv, o, wr);
    for (int c = 0; c < 3; ++c) x2[3 * e + c] = o[c];
}

// one 64-lane workgroup per walker
template <typename T>
__global__ void __launch_bounds__(64) k_mh_accept_ex(int mode, T* __restrict__ x1, T* __restrict__ lp1, const T* __restrict__ x2,
                                                     const T* __restrict__ logabs2, const T* __restrict__ uniform,
                                                     const T* __restrict__ normal, T width, const T* __restrict__ aux1,
                                                     const T* __restrict__ aux2, int n_aux, int n_elec, T* __restrict__ n_accept,
                                                     const T* __restrict__ gmax) {
    const long w = blockIdx.x;
    const int n3 = 3 * n_elec;
    T acc = 0;
    for (int e = threadIdx.x; e < n_elec; e += 64) {
        const size_t o = (size_t)w * n3 + 3 * e;
        if (mode == 1) {
            T p1[3] = {x1[o], x1[o + 1], x1[o + 2]}, p2[3] = {x2[o], x2[o + 1], x2[o + 2]};
            const T s1 = width * harmonic_mean_dist(p1, aux1, n_aux), s2 = width * harmonic_mean_dist(p2, aux1, n_aux);
            T d2 = 0;
            for (int c = 0; c < 3; ++c) d2 += (p1[c] - p2[c]) * (p1[c] - p2[c]);
            // lq_2 - lq_1, _log_prob_gaussian (qmc.py:26-42): -d^2/2 sigma^2 - 3 log sigma, reverse minus forward
            acc += (T(-0.5) * d2 / (s2 * s2) - 3 * ds_log(s2)) - (T(-0.5) * d2 / (s1 * s1) - 3 * ds_log(s1));
        } else {
            T g1[3] = {aux1[o], aux1[o + 1], aux1[o + 2]}, g2[3] = {aux2[o], aux2[o + 1], aux2[o + 2]};
            const T f1 = limdrift_factor(g1, gmax[0]), f2 = limdrift_factor(g2, gmax[1]);
            for (int c = 0; c < 3; ++c) {
                const T ga = width * normal[o + c], bk = ga + width * width * (f1 * g1[c] + f2 * g2[c]);
                acc += ga * ga - bk * bk;                                            // forward - backward, qmc.py:119-122
            }
        }
    }
    acc = wave_sum(acc);
    T lp2 = 2 * logabs2[w];
    T ratio;
    if (mode == 1) ratio = lp2 + acc - lp1[w];                                      // qmc.py:212
    else { lp2 += acc / (2 * width * width); ratio = lp2 - lp1[w]; }                // qmc.py:123-126
    const bool cond = ratio > ds_log(uniform[w]);
    if (cond)
        for (int c = threadIdx.x; c < n3; c += 64) x1[(size_t)w * n3 + c] = x2[(size_t)w * n3 + c];
    __syncthreads();
    if (threadIdx.x == 0 && cond) {
        lp1[w] = lp2;
        atomicAdd(n_accept, T(1));
    }
}

// helpers of the fused importance-sampled loop (ds_mcmc_step_importance)
template <typename T> __global__ void k_real_part(const T* __restrict__ gc, size_t n, T* __restrict__ g) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i] = gc[2 * i];                       // grad log|psi| = Re of the complex gradient of log psi
}
template <typename T> __global__ void k_philox_noise(PhiloxKey key, unsigned long long step, size_t n_elec, long B, T* __restrict__ normal,
                                                     T* __restrict__ uniform) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_elec) {
        double z[3];
        philox_normal3(key, step, e, z);
        for (int c = 0; c < 3; ++c) normal[3 * e + c] = (T)z[c];
    }
    if (e < (size_t)B) uniform[e] = (T)philox_uniform(key, step, (unsigned long long)e);
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
