// Attack-stage kernels (between embed and detect) for gfx950.
//
// Reference: /root/reference/scripts/attacks.py
//   PCMBitDepthConversion.apply :44-70      -> pcm_quantize_kernel
//   Resample.apply :267-294 (scipy.signal.resample_poly = upfirdn with a Kaiser FIR)
//                                            -> upfirdn_kernel (also the 44.1k -> 16k front end,
//                                               scripts/test.py:60-63)
//   LowPassFilter / HighPassFilter :400-455 (scipy lfilter, float64, zero state)
//   RandomBandstop :324-356 (scipy filtfilt, odd padding, lfilter_zi) -> iir_kernel
//   DeleteSamples :162-178, Cropout :192-205, SampleSupression :370-385 -> segment_copy_kernel
// Extension (not in the reference, BASELINE.json north_star): additive Gaussian noise
// with a counter-based Philox generator, specified in oracle/aware_oracle.py.
#include "common.hpp"
#include "kernels.h"

namespace aware {

__global__ __launch_bounds__(256) void pcm_quantize_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            const int* __restrict__ off, const int* __restrict__ len,
                                                            const unsigned long long* __restrict__ pmax,
                                                            const int* __restrict__ pcount, int pstride, float q,
                                                            float lo, float hi) {
    __shared__ unsigned long long red[4];
    const int b = blockIdx.y;
    ClipNorm cn = clip_norm_from_partials(pmax + (size_t)b * pstride, pcount[b], red);
    const int n = len[b];
    const float* x = in + off[b];
    float* y = out + off[b];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float v = x[i] / cn.m;                 // audio / max(|audio| + 1e-8)
        v = v * q;
        v = fminf(fmaxf(v, lo), hi);           // np.clip
        v = truncf(v);                         // astype(int): toward zero
        y[i] = v / q;
    }
}

// WaveformNormalizer as a stand-alone op: x / max(|x| + 1e-8)  (utils/audio/waveform.py:18-19)
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         const int* __restrict__ off, const int* __restrict__ len,
                                                         const unsigned long long* __restrict__ pmax,
                                                         const int* __restrict__ pcount, int pstride) {
    __shared__ unsigned long long red[4];
    const int b = blockIdx.y;
    ClipNorm cn = clip_norm_from_partials(pmax + (size_t)b * pstride, pcount[b], red);
    const int n = len[b];
    const float* x = in + off[b];
    float* y = out + off[b];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = x[i] / cn.m;
}

// y[j] = sum_i x[i] * h[j*down - i*up + half_len], float32, x ascending (scipy upfirdn order)
__global__ __launch_bounds__(256) void upfirdn_kernel(const float* __restrict__ in, const int* __restrict__ in_off,
                                                       const int* __restrict__ in_len, float* __restrict__ out,
                                                       const int* __restrict__ out_off, const int* __restrict__ out_len,
                                                       const float* __restrict__ h, int nh, int up, int down,
                                                       int half_len) {
    const int b = blockIdx.y;
    const int n_in = in_len[b], n_out = out_len[b];
    const float* x = in + in_off[b];
    float* y = out + out_off[b];
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_out; j += gridDim.x * blockDim.x) {
        const long pos = (long)j * down + half_len;
        int i_hi = (int)(pos / up);
        long tap = pos - (long)i_hi * up;               // tap index for i = i_hi
        int i_lo = i_hi - (int)((nh - 1 - tap) / up);
        if (i_lo < 0) i_lo = 0;
        if (i_hi > n_in - 1) i_hi = n_in - 1;
        float acc = 0.f;
        for (int i = i_lo; i <= i_hi; ++i) acc += x[i] * h[pos - (long)i * up];
        y[j] = acc;
    }
}

// Direct-form-II-transposed IIR in float64 (scipy lfilter / filtfilt), one workgroup per clip, parallel in time.
// The recurrence is linear, so a clip is cut into kIirChunks chunks of L samples, one per thread:
//   pass A  every thread runs the recurrence over its chunk from a ZERO state (thread 0: from the true initial state)
//           and keeps only the end state v[c]; meanwhile one more wave computes the L-step state transition matrix
//           M = A^L (A = one zero-input step): A^16 by running 16 steps from the unit states, then square-and-multiply;
//   pass B  one wave chains the true chunk-start states z0[c+1] = M z0[c] + v[c] (lane k owns row k of M);
//   pass C  every thread runs the same recurrence again over its chunk from its true start state and stores y.
// Passes A and C are the sequential algorithm in plain float64.  M and pass B are in DOUBLE-DOUBLE: the transposed direct
// form of a narrow band-stop (eight poles in two tight clusters near the unit circle) is far from normal -- |M| reaches
// 1e7..1e8 while M z0 stays O(|z0|) -- so a float64 M z0 loses 7-8 digits to cancellation (measured: 3e-2 absolute error
// on a 400 Hz band, against 1.5e-8 for the sequential recurrence itself).  With M and the chaining exact to 1e-32 the
// result differs from scipy's by no more than the sequential recurrence's own rounding (5e-8 at a 300 Hz band, 1e-11 at
// 2.5 kHz; tests/test_gpu_attacks.py::test_iir_time_parallel_long_ragged).
//   mode 0: lfilter with zero state (LowPassFilter / HighPassFilter, attacks.py:400-455).
//   mode 1: filtfilt (RandomBandstop, attacks.py:324-356): odd extension by 3*ncoef samples, steady-state initial
//           conditions zi*x0, forward then backward; needs scratch[B][maxlen + 6*ncoef] doubles.
constexpr int kMaxCoef = 12;
constexpr int kIirChunks = 128;                  // worker threads (= chunks) per clip; + one wave for the matrix

struct dd { double hi, lo; };                    // unevaluated sum hi + lo, |lo| <= ulp(hi)/2
// The error-free transformations below are exact only as written: no contraction of a product into a neighbouring add
// (hipcc's default -ffp-contract=fast fused `p + e` with the multiplication that produced p and destroyed the low words).
__device__ __forceinline__ dd dd_renorm(double s, double e) {
#pragma clang fp contract(off)
    const double h = s + e;
    return dd{h, e - (h - s)};
}
__device__ __forceinline__ dd dd_add(dd a, dd b) {
#pragma clang fp contract(off)
    const double s = a.hi + b.hi, bb = s - a.hi;
    const double e = ((a.hi - (s - bb)) + (b.hi - bb)) + (a.lo + b.lo);
    return dd_renorm(s, e);
}
__device__ __forceinline__ dd dd_mul(dd a, dd b) {
#pragma clang fp contract(off)
    const double p = a.hi * b.hi;
    const double c1 = a.hi * b.lo, c2 = a.lo * b.hi;
    const double e = __builtin_fma(a.hi, b.hi, -p) + (c1 + c2);
    return dd_renorm(p, e);
}
__device__ __forceinline__ dd dd_mul_d(dd a, double b) {
#pragma clang fp contract(off)
    const double p = a.hi * b;
    const double c1 = a.lo * b;
    const double e = __builtin_fma(a.hi, b, -p) + c1;
    return dd_renorm(p, e);
}
__device__ __forceinline__ dd dd_neg(dd a) { return dd{-a.hi, -a.lo}; }
__device__ __forceinline__ double readlane_d(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)b, lane), hi = __builtin_amdgcn_readlane((unsigned)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int NC>
__global__ __launch_bounds__(kIirChunks + 64) void iir_kernel(const float* __restrict__ in, const int* __restrict__ off,
                                                              const int* __restrict__ len, void* __restrict__ outv,
                                                              int out_f64, const double* __restrict__ bc,
                                                              const double* __restrict__ ac, const double* __restrict__ zic,
                                                              int mode, double* __restrict__ scratch, int sstride) {
    constexpr int NS = NC - 1, NT = kIirChunks, CH = 16, NE = NS * NS;
    __shared__ double Vs[NT][NS];                // pass A: zero-state end states; after pass B: true start states
    __shared__ dd Mx[3][NE];                     // matrix wave: result / running power / product; Mx[0] = M at the end
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const bool matrix_wave = tid >= NT;
    double bb[NC], aa[NC], z[NS];
#pragma unroll
    for (int k = 0; k < NC; ++k) { bb[k] = bc[(size_t)b * NC + k]; aa[k] = ac[(size_t)b * NC + k]; }
    const int n = len[b];
    const float* x = in + off[b];
    double* od = reinterpret_cast<double*>(outv) + off[b];
    float* of = reinterpret_cast<float*>(outv) + off[b];
    auto step = [&](double xi) {
        const double yi = bb[0] * xi + z[0];
#pragma unroll
        for (int k = 0; k < NS - 1; ++k) z[k] = z[k + 1] + bb[k + 1] * xi - aa[k + 1] * yi;
        z[NS - 1] = bb[NS] * xi - aa[NS] * yi;
        return yi;
    };
    const int edge = mode ? 3 * NC : 0;
    const int cnt = n + 2 * edge;                              // samples the recurrence runs over
    const int L = CH * ((cnt + CH * NT - 1) / (CH * NT));     // chunk length, a multiple of CH
    const int nch = (cnt + L - 1) / L;
    double* s = scratch + (size_t)b * sstride;

    if (matrix_wave) {
        // A^16, column j on lane j: 16 zero-input steps from the unit state e_j, in double-double
        volatile dd* R = Mx[0];
        volatile dd* P = Mx[1];
        volatile dd* T = Mx[2];
        {
            const int j = lane < NS ? lane : NS - 1;
            dd zz[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) zz[k] = dd{k == j ? 1.0 : 0.0, 0.0};
            for (int i = 0; i < CH; ++i) {
                const dd yi = zz[0];
#pragma unroll
                for (int k = 0; k < NS - 1; ++k) zz[k] = dd_add(zz[k + 1], dd_neg(dd_mul_d(yi, aa[k + 1])));
                zz[NS - 1] = dd_neg(dd_mul_d(yi, aa[NS]));
            }
            if (lane < NS) {
#pragma unroll
                for (int k = 0; k < NS; ++k) { P[k * NS + j].hi = zz[k].hi; P[k * NS + j].lo = zz[k].lo; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // M = (A^16)^(L/16) by square-and-multiply (all factors are powers of A, so the order of a product is free);
        // entry (i, l) of a product on lane i*NS + l (two rounds of lanes when NS*NS > 64)
        auto matmul = [&](volatile dd* X, volatile dd* Y, volatile dd* Z) {     // Z = X Y
            for (int e = lane; e < NE; e += 64) {
                const int i = e / NS, l = e % NS;
                dd acc{0.0, 0.0};
                for (int q = 0; q < NS; ++q) {
                    const dd xv{X[i * NS + q].hi, X[i * NS + q].lo}, yv{Y[q * NS + l].hi, Y[q * NS + l].lo};
                    acc = dd_add(acc, dd_mul(xv, yv));
                }
                Z[e].hi = acc.hi; Z[e].lo = acc.lo;
            }
            __builtin_amdgcn_wave_barrier();
        };
        auto copy = [&](volatile dd* X, volatile dd* Z) {
            for (int e = lane; e < NE; e += 64) { Z[e].hi = X[e].hi; Z[e].lo = X[e].lo; }
            __builtin_amdgcn_wave_barrier();
        };
        int ex = L / CH;
        bool have = false;
        while (ex) {
            if (ex & 1) {
                if (!have) { copy(P, R); have = true; }
                else { matmul(R, P, T); copy(T, R); }
            }
            ex >>= 1;
            if (ex) { matmul(P, P, T); copy(T, P); }
        }
    }
    if (mode && !matrix_wave) {
        // odd extension (scipy filtfilt padtype='odd', padlen = 3*max(len(a), len(b)))
        const double x0 = (double)x[0], xl = (double)x[n - 1];
        for (int i = tid; i < edge; i += NT) {
            // (a clip no longer than the padding is refused by the host binding, as scipy refuses it; the clamps keep
            //  a direct C-ABI call with such a clip inside its buffer)
            s[i] = 2.0 * x0 - (double)x[min(edge - i, n - 1)];
            s[edge + n + i] = 2.0 * xl - (double)x[max(n - 2 - i, 0)];
        }
        for (int i = tid; i < n; i += NT) s[edge + i] = (double)x[i];
    }
    __syncthreads();

    // one pass of the recurrence over elements [0, cnt) of a sequence given by ld(j) / st(j, y), from the state zinit
    auto scan = [&](auto ld, auto st, const double (&zinit)[NS]) {
        const int j0 = tid * L;
        const int m = matrix_wave ? 0 : min(max(cnt - j0, 0), L);
        if (!matrix_wave) {
#pragma unroll
            for (int k = 0; k < NS; ++k) z[k] = tid == 0 ? zinit[k] : 0.0;
            int i = 0;
            for (; i + CH <= m; i += CH) {
                double xb[CH];
#pragma unroll
                for (int q = 0; q < CH; ++q) xb[q] = ld(j0 + i + q);
#pragma unroll
                for (int q = 0; q < CH; ++q) step(xb[q]);
            }
            for (; i < m; ++i) step(ld(j0 + i));
#pragma unroll
            for (int k = 0; k < NS; ++k) Vs[tid][k] = z[k];
        }
        __syncthreads();
        if (tid < 64) {
            const int k = lane < NS ? lane : NS - 1;
            dd mrow[NS];
#pragma unroll
            for (int q = 0; q < NS; ++q) mrow[q] = Mx[0][k * NS + q];
            dd cur{Vs[0][k], 0.0};                               // thread 0 started from the true state
            for (int c = 1; c < nch; ++c) {
                dd nxt{Vs[c][k], 0.0};
                if (lane < NS) Vs[c][k] = cur.hi;                // true start state of chunk c, rounded to float64
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    const dd cq{readlane_d(cur.hi, q), readlane_d(cur.lo, q)};
                    nxt = dd_add(nxt, dd_mul(mrow[q], cq));
                }
                cur = nxt;
            }
        }
        __syncthreads();
        if (!matrix_wave && m > 0) {
#pragma unroll
            for (int k = 0; k < NS; ++k) z[k] = tid == 0 ? zinit[k] : Vs[tid][k];
            int i = 0;
            for (; i + CH <= m; i += CH) {
                double xb[CH];
#pragma unroll
                for (int q = 0; q < CH; ++q) xb[q] = ld(j0 + i + q);
#pragma unroll
                for (int q = 0; q < CH; ++q) xb[q] = step(xb[q]);
#pragma unroll
                for (int q = 0; q < CH; ++q) st(j0 + i + q, xb[q]);
            }
            for (; i < m; ++i) st(j0 + i, step(ld(j0 + i)));
        }
        __syncthreads();
    };

    double zinit[NS];
    if (mode == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) zinit[k] = 0.0;
        scan([&](int j) { return (double)x[j]; },
             [&](int j, double y) { if (out_f64) od[j] = y; else of[j] = (float)y; }, zinit);
        return;
    }
    const double s0 = s[0];
#pragma unroll
    for (int k = 0; k < NS; ++k) zinit[k] = zic[(size_t)b * NS + k] * s0;
    scan([&](int j) { return s[j]; }, [&](int j, double y) { s[j] = y; }, zinit);            // forward
    const double y0 = s[cnt - 1];
#pragma unroll
    for (int k = 0; k < NS; ++k) zinit[k] = zic[(size_t)b * NS + k] * y0;
    scan([&](int j) { return s[cnt - 1 - j]; }, [&](int j, double y) { s[cnt - 1 - j] = y; }, zinit);   // backward
    for (int i = tid; i < n; i += NT + 64) {
        if (out_f64) od[i] = s[edge + i]; else of[i] = (float)s[edge + i];
    }
}

// ---- EXTENSION: Gaussian noise at a target SNR (Philox-4x32-10 + Box-Muller) -------------
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
        unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

__global__ __launch_bounds__(256) void power_kernel(const float* __restrict__ in, const int* __restrict__ off,
                                                     const int* __restrict__ len, double* __restrict__ power) {
    __shared__ double red[4];
    const int b = blockIdx.x;
    const int n = len[b];
    const float* x = in + off[b];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)x[i] * (double)x[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) power[b] = (red[0] + red[1] + red[2] + red[3]) / (double)n;
}

__global__ __launch_bounds__(256) void gaussian_noise_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              const int* __restrict__ off, const int* __restrict__ len,
                                                              const unsigned* __restrict__ seeds,
                                                              const double* __restrict__ power, float snr_db) {
    const int b = blockIdx.y;
    const int n = len[b];
    const float* x = in + off[b];
    float* y = out + off[b];
    const double sigma = sqrt(power[b] / pow(10.0, (double)snr_db / 10.0));
    const int nblk = (n + 3) / 4;
    const double two_pi = 6.283185307179586476925286766559;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nblk; q += gridDim.x * blockDim.x) {
        unsigned r[4];
        philox4x32_10((unsigned)q, 0u, 0u, 0u, seeds[b], 0x5EEDu, r);
        double u0 = ((double)r[0] + 0.5) / 4294967296.0, u1 = ((double)r[1] + 0.5) / 4294967296.0;
        double u2 = ((double)r[2] + 0.5) / 4294967296.0, u3 = ((double)r[3] + 0.5) / 4294967296.0;
        double ra = sqrt(-2.0 * log(u0)), rb = sqrt(-2.0 * log(u2));
        double zz[4] = {ra * cos(two_pi * u1), ra * sin(two_pi * u1), rb * cos(two_pi * u3), rb * sin(two_pi * u3)};
        for (int e = 0; e < 4; ++e) {
            int i = 4 * q + e;
            if (i < n) y[i] = (float)((double)x[i] + sigma * zz[e]);
        }
    }
}

// delete a span (out shorter than in) or zero it (same length)
__global__ __launch_bounds__(256) void segment_copy_kernel(const float* __restrict__ in, const int* __restrict__ in_off,
                                                            float* __restrict__ out, const int* __restrict__ out_off,
                                                            const int* __restrict__ out_len,
                                                            const int* __restrict__ cut_start,
                                                            const int* __restrict__ cut_len, int zero_fill) {
    const int b = blockIdx.y;
    const int n = out_len[b], s0 = cut_start[b], k = cut_len[b];
    const float* x = in + in_off[b];
    float* y = out + out_off[b];
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        if (zero_fill) y[j] = (j >= s0 && j < s0 + k) ? 0.f : x[j];
        else y[j] = (j < s0) ? x[j] : x[j + k];
    }
}

// Resample's branch for sr // target_sr = k > 1 (scripts/attacks.py:275-288): keep every k-th sample, then
// np.interp back onto 0..n-1.  np.interp works in float64: slope = (y[i+1] - y[i]) / (x[i+1] - x[i]),
// value = slope * (x - x[i]) + y[i] (two roundings, no fused multiply-add), exact sample points return y[i], positions
// beyond the last kept sample return it (numpy/_core/src/multiarray/compiled_base.c arr_interp).  Output float64.
__global__ __launch_bounds__(256) void decimate_interp_kernel(const float* __restrict__ in, const int* __restrict__ off,
                                                               const int* __restrict__ len, double* __restrict__ out, int k) {
    const int b = blockIdx.y;
    const int n = len[b];
    const float* x = in + off[b];
    double* y = out + off[b];
    const int last = ((n - 1) / k) * k;                     // position of the last kept sample
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const int i0 = (j / k) * k;
        double v;
        if (j >= last) v = (double)x[last];
        else if (j == i0) v = (double)x[i0];
        else {
            const double y0 = (double)x[i0], y1 = (double)x[i0 + k];
            const double slope = __ddiv_rn(__dsub_rn(y1, y0), (double)k);
            v = __dadd_rn(__dmul_rn(slope, (double)(j - i0)), y0);
        }
        y[j] = v;
    }
}

static inline int gx(int max_len) { int g = (max_len + 1023) / 1024; return g < 1 ? 1 : g; }
void launch_decimate_interp(const float* in, const int* off, const int* len, double* out, int k, int B, int max_len,
                            hipStream_t st) {
    hipLaunchKernelGGL(decimate_interp_kernel, dim3(gx(max_len), B), dim3(256), 0, st, in, off, len, out, k);
}

void launch_pcm_quantize(const float* in, float* out, const int* off, const int* len, const unsigned long long* pmax,
                         const int* pcount, int pstride, float q, float lo, float hi, int B, int max_len,
                         hipStream_t st) {
    hipLaunchKernelGGL(pcm_quantize_kernel, dim3(gx(max_len), B), dim3(256), 0, st, in, out, off, len, pmax, pcount,
                       pstride, q, lo, hi);
}
void launch_normalize(const float* in, float* out, const int* off, const int* len, const unsigned long long* pmax,
                      const int* pcount, int pstride, int B, int max_len, hipStream_t st) {
    hipLaunchKernelGGL(normalize_kernel, dim3(gx(max_len), B), dim3(256), 0, st, in, out, off, len, pmax, pcount, pstride);
}
void launch_upfirdn(const float* in, const int* in_off, const int* in_len, float* out, const int* out_off,
                    const int* out_len, const float* h, int nh, int up, int down, int half_len, int B, int max_out,
                    hipStream_t st) {
    hipLaunchKernelGGL(upfirdn_kernel, dim3(gx(max_out), B), dim3(256), 0, st, in, in_off, in_len, out, out_off, out_len,
                       h, nh, up, down, half_len);
}
void launch_iir_full(const float* in, const int* off, const int* len, void* out, int out_f64, const double* b,
                     const double* a, const double* zi, int ncoef, int mode, double* scratch, int sstride, int B,
                     hipStream_t st) {
#define IIR(NC_)                                                                                                       \
    hipLaunchKernelGGL(iir_kernel<NC_>, dim3(B), dim3(kIirChunks + 64), 0, st, in, off, len, out, out_f64, b, a, zi, mode, \
                       scratch, sstride)
    switch (ncoef) {
        case 2: IIR(2); break; case 3: IIR(3); break; case 4: IIR(4); break; case 5: IIR(5); break;
        case 6: IIR(6); break; case 7: IIR(7); break; case 8: IIR(8); break; case 9: IIR(9); break;
        case 10: IIR(10); break; case 11: IIR(11); break; default: IIR(12); break;
    }
#undef IIR
}
void launch_gaussian_noise_full(const float* in, float* out, const int* off, const int* len, const unsigned* seeds,
                                double* power, float snr_db, int B, int max_len, hipStream_t st) {
    hipLaunchKernelGGL(power_kernel, dim3(B), dim3(256), 0, st, in, off, len, power);
    hipLaunchKernelGGL(gaussian_noise_kernel, dim3(gx(max_len / 4 + 1), B), dim3(256), 0, st, in, out, off, len, seeds,
                       power, snr_db);
}
void launch_segment_copy(const float* in, const int* in_off, float* out, const int* out_off, const int* out_len,
                         const int* cut_start, const int* cut_len, int zero_fill, int B, int max_len, hipStream_t st) {
    hipLaunchKernelGGL(segment_copy_kernel, dim3(gx(max_len), B), dim3(256), 0, st, in, in_off, out, out_off, out_len,
                       cut_start, cut_len, zero_fill);
}

}  // namespace aware

// ---------------------------------------------------------------------------------
// EXTENSION (not in the reference; BASELINE.json north_star "MP3-like quantisation
// surrogate"): per-frame log-magnitude quantisation of a full one-sided spectrum.
// Specification: oracle/aware_oracle.py::mp3_surrogate_attack.  One wave per frame:
//   fmax = max_k |X_k| ; db = 20 log10(|X| / fmax) ; q = round(db / step) * step ;
//   |X| <- fmax * 10^(q/20), zero where db < floor ; the phase is kept.
// spec: [frames][520] complex64 (bins 0..512), in place.
// ---------------------------------------------------------------------------------
namespace aware {
__global__ __launch_bounds__(256) void spectral_quantize_kernel(cf* __restrict__ spec, int nframes, float step_db,
                                                                 float floor_db) {
    const int frame = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (frame >= nframes) return;
    cf* X = spec + (size_t)frame * 520;
    float mg[9];
    float mx = 0.f;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int k = lane + 64 * r;
        mg[r] = 0.f;
        if (k <= 512) { cf x = X[k]; mg[r] = sqrtf(x.x * x.x + x.y * x.y); }
        mx = fmaxf(mx, mg[r]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    mx = fmaxf(mx, 1e-12f);
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int k = lane + 64 * r;
        if (k > 512) continue;
        const float m = fmaxf(mg[r], 1e-12f);
        const float db = 20.0f * log10f(m / mx);
        const float q = rintf(db / step_db) * step_db;          // torch.round: half to even
        float mq = mx * powf(10.0f, q / 20.0f);
        if (db < floor_db) mq = 0.f;
        const float sc = (mg[r] > 0.f) ? mq / mg[r] : 0.f;      // keep the phase
        cf x = X[k];
        X[k] = mk(x.x * sc, x.y * sc);
    }
}
// Backward of the surrogate as a DIFFERENTIABLE op (BASELINE north_star "differentiable MP3-like quantisation surrogates"):
// Y = Q(|X|) X/|X|.  The quantiser's staircase has zero derivative almost everywhere, so the magnitude path is straight-through
// (dQ/d|X| := 1 on kept bins, 0 on bins dropped below the floor; the frame maximum is treated as a constant), the phase path
// is exact (Y keeps X's phase, scaled by Q/|X|):   gX = u [ keep Re(G conj u) + i (Q/|X|) Im(G conj u) ],  u = X/|X|.
// X: the spectrum BEFORE quantisation; G = dL/dRe Y + i dL/dIm Y; specified by oracle/aware_oracle.py::mp3_surrogate_spectrum.
__global__ __launch_bounds__(256) void spectral_quantize_bwd_kernel(const cf* __restrict__ spec, const cf* __restrict__ gout,
                                                                     cf* __restrict__ gin, int nframes, float step_db, float floor_db) {
    const int frame = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (frame >= nframes) return;
    const cf* X = spec + (size_t)frame * 520;
    const cf* G = gout + (size_t)frame * 520;
    cf* O = gin + (size_t)frame * 520;
    cf xv[9];
    float mg[9];
    float mx = 0.f;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int k = lane + 64 * r;
        mg[r] = 0.f;
        xv[r] = mk(0.f, 0.f);
        if (k <= 512) { xv[r] = X[k]; mg[r] = sqrtf(xv[r].x * xv[r].x + xv[r].y * xv[r].y); }
        mx = fmaxf(mx, mg[r]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    mx = fmaxf(mx, 1e-12f);
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int k = lane + 64 * r;
        if (k > 512) continue;
        const float m = fmaxf(mg[r], 1e-12f);
        const float db = 20.0f * log10f(m / mx);
        const float q = rintf(db / step_db) * step_db;
        const bool keep = !(db < floor_db) && mg[r] > 0.f;
        const float sc = keep ? mx * powf(10.0f, q / 20.0f) / mg[r] : 0.f;
        const cf g = G[k];
        cf o = mk(0.f, 0.f);
        if (keep) {
            const float ux = xv[r].x / mg[r], uy = xv[r].y / mg[r];
            const float re = g.x * ux + g.y * uy;                 // Re(G conj u)
            const float im = (g.y * ux - g.x * uy) * sc;          // (Q/|X|) Im(G conj u)
            o = mk(ux * re - uy * im, uy * re + ux * im);         // u (re + i im)
        }
        O[k] = o;
    }
}
void launch_spectral_quantize_bwd(const void* spec, const void* gout, void* gin, int nframes, float step_db, float floor_db,
                                  hipStream_t st) {
    hipLaunchKernelGGL(spectral_quantize_bwd_kernel, dim3((nframes + 3) / 4), dim3(256), 0, st, (const cf*)spec, (const cf*)gout,
                       (cf*)gin, nframes, step_db, floor_db);
}
void launch_spectral_quantize(void* spec, int nframes, float step_db, float floor_db, hipStream_t st) {
    hipLaunchKernelGGL(spectral_quantize_kernel, dim3((nframes + 3) / 4), dim3(256), 0, st, (cf*)spec, nframes, step_db,
                       floor_db);
}

// ---------------------------------------------------------------------------------
// SNR in dB of `a` (output) against `b` (target) per clip over len[c] samples:
// 10 log10(mean(a^2) / mean((a-b)^2)), +inf when identical  (metrics/audio.py:68-89)
// One workgroup per clip, f64 accumulation in a fixed order (deterministic).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void snr_kernel(const float* __restrict__ a, const int* __restrict__ a_off,
                                                   const float* __restrict__ b, const int* __restrict__ b_off,
                                                   const int* __restrict__ len, double* __restrict__ out) {
    __shared__ double r1[4], r2[4];
    const int c = blockIdx.x, n = len[c];
    const float* x = a + a_off[c];
    const float* y = b + b_off[c];
    double s1 = 0.0, s2 = 0.0;
    for (int i0 = 0; i0 < n; i0 += 256 * 8) {
        float xv[8], yv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {                       // loads first (clamped), then use
            const int i = min(i0 + threadIdx.x + 256 * u, n - 1);
            xv[u] = x[i]; yv[u] = y[i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + (int)threadIdx.x + 256 * u < n) {
                const double p = (double)xv[u], d = (double)xv[u] - (double)yv[u];
                s1 += p * p; s2 += d * d;
            }
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double p = (r1[0] + r1[1]) + (r1[2] + r1[3]), d = (r2[0] + r2[1]) + (r2[2] + r2[3]);
        out[c] = (d == 0.0) ? (double)INFINITY : 10.0 * log10(p / d);
    }
}
void launch_snr(const float* a, const int* a_off, const float* b, const int* b_off, const int* len, double* out, int B,
                hipStream_t st) {
    hipLaunchKernelGGL(snr_kernel, dim3(B), dim3(256), 0, st, a, a_off, b, b_off, len, out);
}


// ---------------------------------------------------------------------------------
// Phase vocoder (time-scale modification at a fixed hop): resamples a one-sided STFT in time by `rate`
// (frame t of the output sits at input position t*rate: magnitudes interpolated linearly, phases advanced by the
// measured per-bin phase increment wrapped to (-pi, pi]) -- the textbook algorithm, as in librosa.phase_vocoder.
// The reference's TimeStretch / PitchShift shell out to the rubberband binary (scripts/attacks.py:208-252, absent
// here): this is a stand-in specified by oracle/aware_oracle.py::phase_vocoder, parity with rubberband unpinned.
// One thread per (clip, bin): the phase accumulation is a sequential scan over time; f64 arithmetic.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void phase_vocoder_kernel(const cf* __restrict__ in, const int* __restrict__ fin,
                                                            cf* __restrict__ out, const int* __restrict__ fout, double rate) {
    const int c = blockIdx.y, k = blockIdx.x * 64 + threadIdx.x;
    if (k > 512) return;
    const int f0 = fin[c], T = fin[c + 1] - f0, g0 = fout[c], To = fout[c + 1] - g0;
    const double TWO_PI = 6.283185307179586476925286766559;
    const double adv = TWO_PI * (double)kHop * (double)k / (double)kNfft;       // expected phase advance per hop
    const cf* D = in + (size_t)f0 * 520 + k;
    cf* O = out + (size_t)g0 * 520 + k;
    cf d0 = D[0];
    double acc = atan2((double)d0.y, (double)d0.x);
    for (int t = 0; t < To; ++t) {
        const double step = (double)t * rate;
        const int i = (int)floor(step);
        const double alpha = step - (double)i;
        cf c0 = mk(0.f, 0.f), c1 = mk(0.f, 0.f);
        if (i < T) c0 = D[(size_t)i * 520];
        if (i + 1 < T) c1 = D[(size_t)(i + 1) * 520];
        const double m0 = sqrt((double)c0.x * c0.x + (double)c0.y * c0.y), m1 = sqrt((double)c1.x * c1.x + (double)c1.y * c1.y);
        const double mag = (1.0 - alpha) * m0 + alpha * m1;
        O[(size_t)t * 520] = mk((float)(mag * cos(acc)), (float)(mag * sin(acc)));
        double dp = atan2((double)c1.y, (double)c1.x) - atan2((double)c0.y, (double)c0.x) - adv;
        dp -= TWO_PI * rint(dp / TWO_PI);                                       // wrap to [-pi, pi]
        acc += adv + dp;
    }
    if (k < 8) {                                       // keep the 7 pad columns of the 520-wide rows finite
        for (int t = 0; t < To; ++t) out[(size_t)(g0 + t) * 520 + 513 + (k < 7 ? k : 6)] = mk(0.f, 0.f);
    }
}
void launch_phase_vocoder(const void* in, const int* fin, void* out, const int* fout, double rate, int B, hipStream_t st) {
    hipLaunchKernelGGL(phase_vocoder_kernel, dim3(9, B), dim3(64), 0, st, (const cf*)in, fin, (cf*)out, fout, rate);
}

}  // namespace aware
