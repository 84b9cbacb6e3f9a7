// Framed STFT / iSTFT kernels and their adjoints for gfx950 (MI355X).
//
// One 256-thread workgroup = 4 wavefronts transforms up to 16 frames of one clip.
// The signal chunk (analysis) or the overlap-add buffer (synthesis) lives in LDS, so
// every sample is read from / written to HBM once per kernel with 16-byte-coalesced
// row accesses; the 75 % frame overlap is served from LDS.  Each wave runs the
// 1024-point real FFT of fft512.hpp on one frame at a time.
//
// Reference operations replaced (all in /root/reference/src/AWARE):
//   analysis  <- utils/audio/stft.py:27-28 (torch.stft, center/reflect, hann, onesided)
//                + STFTDecomposer :54-55 (abs; the unit phasor S/|S| replaces angle)
//                + WaveformNormalizer utils/audio/waveform.py:18-19 fused into the load
//   synthesis <- STFTAssembler stft.py:61-62 (mag*exp(i phase), phasor precomputed)
//                + ISTFT :47-48 (irfft, window, overlap-add, / envelope, trim)
//   *_bwd     <- what torch autograd derives for the two above inside
//                embedding/multibit_embedder.py:95-111, written out by hand; the
//                NAdam + clamp + best-snapshot epilogue replaces :112-122.
#include "common.hpp"
#include "dsp_args.hpp"
#include "kernels.h"

namespace aware {

// ---------------------------------------------------------------------------------
// |x| maximum per clip, as per-segment partials (4096 samples per workgroup)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void absmax_partial_kernel(const float* __restrict__ sig,
                                                                   const int* __restrict__ sig_off,
                                                                   const int* __restrict__ sig_len,
                                                                   unsigned long long* __restrict__ pmax,
                                                                   int pstride) {
    __shared__ unsigned long long red[4];
    const int b = blockIdx.y;
    const int n = sig_len[b];
    const int s0 = blockIdx.x * 4096;
    if (s0 >= n) return;
    const float* x = sig + sig_off[b];
    unsigned long long v = 0;
    for (int i = s0 + threadIdx.x; i < min(n, s0 + 4096); i += kThreads) v = umax64(v, pack_max(fabsf(x[i]), i));
    v = wave_max64(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) pmax[(size_t)b * pstride + blockIdx.x] = umax64(umax64(red[0], red[1]), umax64(red[2], red[3]));
}

// ---------------------------------------------------------------------------------
// Analysis: frames -> windowed real FFT -> epilogue
// ---------------------------------------------------------------------------------

// FULLOUT: write the full one-sided spectrum (generic STFT plug-in) instead of the band outputs
template <int MODE, bool FULLOUT>
__global__ __launch_bounds__(kThreads) void analysis_kernel(AnalysisArgs a) {
    __shared__ float chunk[kChunk];
    __shared__ cf scratch[4][kFftScratch];
    __shared__ unsigned long long red[4];
    __shared__ double dred[4];

    const int b = blockIdx.y;
    const int f0 = a.frame_off[b];
    const int T = a.frame_off[b + 1] - f0;
    const int t0 = blockIdx.x * kFramesPerWG;
    if (t0 >= T) return;
    const int nfr = min(kFramesPerWG, T - t0);
    const int n = a.sig_len[b];
    const float* x = a.sig + a.sig_off[b];
    const int tid = threadIdx.x;

    // ---- per-clip scalars -----------------------------------------------------------
    float m = 1.f, m2 = 1.f;
    unsigned kmax = 0xFFFFFFFFu;
    if (a.pmax) {
        ClipNorm cn = clip_norm_from_partials(a.pmax + (size_t)b * a.pstride, a.pcount[b], red);
        m = cn.m;
        m2 = (MODE == AN_ADJ || a.double_norm) ? cn.m2 : 1.f;
        kmax = cn.k;
    }
    float adot = 0.f, smax = 0.f;
    if (MODE == AN_ADJ && a.pdot) {       // (null: plain ISTFT backward of the plug-in seam, no normaliser behind it)
        // A = sum_j g2[j]*y2[j] (fixed summation order) and the sign of the max sample
        double s = 0.0;
        const double* pd = a.pdot + (size_t)b * a.pstride;
        for (int i = tid; i < a.pcount[b]; i += kThreads) s += pd[i];
        s = wave_sum_d(s);
        if ((tid & 63) == 0) dred[tid >> 6] = s;
        __syncthreads();
        adot = (float)(dred[0] + dred[1] + dred[2] + dred[3]);
        float yk = a.yraw[a.sig_off[b] + kmax];
        smax = (yk > 0.f) ? 1.f : ((yk < 0.f) ? -1.f : 0.f);
    }

    // ---- stage the signal chunk: padded positions [256*t0, 256*t0 + 256*(nfr-1)+1024) --
    const int p0 = kHop * t0;
    const int cnt = kHop * (nfr - 1) + kNfft;
    {
        // all loads of the chunk are issued before any of them is used (a load inside a per-sample branch
        // would wait for its own latency 19 times over)
        constexpr int NIT = kChunk / kThreads;
        const bool small_t = T < 4;                    // clips shorter than the envelope tables assume
        const float inv_m = 1.0f / m, inv_m2 = 1.0f / m2, inv_mm2 = 1.0f / (m * m2);
        float xin[NIT], env[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int p = p0 + tid + kThreads * it;
            int src = p - kHalf;
            if (MODE == AN_NORM) {
                if (src < 0) src = -src;
                if (src >= n) src = 2 * (n - 1) - src;
            }
            xin[it] = x[min(max(src, 0), n - 1)];
            if (MODE == AN_ADJ) env[it] = small_t ? 1.f : a.plan.env_tab[ola_envelope_index(p, T)];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + kThreads * it;
            if (i < cnt) {
                const int p = p0 + i;
                const int src = p - kHalf;
                float v;
                if (MODE == AN_NORM) {
                    v = xin[it];
                    if (a.pmax) { v = v * inv_m; if (a.double_norm) v = v * inv_m2; }
                } else {
                    // adjoint of (trim, / envelope): zero outside the kept region
                    if (src >= 0 && src < n) {
                        float g = xin[it];
                        if ((unsigned)src == kmax) g -= adot * smax;
                        g = g * inv_mm2;
                        v = g * fast_rcp(small_t ? ola_envelope_loop(a.plan.window2, p, T) : env[it]);
                    } else {
                        v = 0.f;
                    }
                }
                chunk[i] = v;
            }
        }
    }
    __syncthreads();

    // ---- per-wave FFTs ---------------------------------------------------------------
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave index is wave-uniform: keep it scalar
    cf* s = scratch[wave];
    FftLaneConst fc;
    fft_lane_const(lane, a.plan.tw512, fc);
    float wr[16];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        wr[2 * r] = a.plan.window[2 * (lane + 64 * r)];
        wr[2 * r + 1] = a.plan.window[2 * (lane + 64 * r) + 1];
    }
    const int band_lo = a.plan.band_lo, nband = a.plan.nband;
    float4 sc = make_float4(0.f, 0.f, 1.f, 0.f);
    float inv_bc2 = 1.f;
    int improved = 0;
    if (MODE == AN_ADJ && a.do_step) {
        // the read-out kernel of this iteration already advanced the counter; clamped to the table (the C ABI
        // refuses more than num_iterations steps, so the clamp only guards a misuse of the launcher)
        sc = a.sched[min(max(*a.step - 1, 0), a.sched_len - 1)];
        inv_bc2 = 1.0f / sc.z;
        improved = a.improved[b];
    }

    // The adjoint epilogue reads six operands per bin that do not depend on the FFT: issue
    // those loads before the transform so their latency hides behind it (bins of the band sit
    // in registers r = 0..4 when band_lo < 64, the model card's case).
    const bool pre_ok = (MODE == AN_ADJ) && a.do_step && band_lo < 64;
#pragma unroll 1
    for (int fr = wave; fr < nfr; fr += 4) {
        cf v[8];
        const size_t row = (size_t)(f0 + t0 + fr);
        cf preP[5];
        float preM[5], preV[5], preC[5], preL[5], preH[5];
        if (MODE == AN_ADJ) {
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int f = lane + 64 * r - band_lo;
                preP[r] = mk(0.f, 0.f); preM[r] = preV[r] = preC[r] = preL[r] = preH[r] = 0.f;
                if (pre_ok && f >= 0 && f < nband) {
                    const size_t idx = row * kFS + f;
                    preP[r] = a.phasor[idx]; preM[r] = a.mom[idx]; preV[r] = a.vel[idx]; preC[r] = a.coef[idx];
                    preL[r] = a.lo[idx]; preH[r] = a.hi[idx];
                }
            }
        }
        const float2* c2 = reinterpret_cast<const float2*>(chunk + kHop * fr);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float2 q = c2[lane + 64 * r];
            v[r] = mk(q.x * wr[2 * r], q.y * wr[2 * r + 1]);
        }
        fft512_wave<-1>(lane, v, fc, s);
        rfft_split_store(lane, v, s);
        wave_sync();
        if (FULLOUT) {
            cf* out = a.full + row * 520;
            // AN_ADJ: adjoint of irfft (torch.istft's C2R transform): dL/dX[k] = (c_k / 1024) rfft(.)[k] with c_k = 2 on
            // the interior bins and 1 at DC / Nyquist (whose imaginary parts the transform ignores)
            const float si = (MODE == AN_ADJ) ? (1.0f / 512.0f) : 1.0f, se = (MODE == AN_ADJ) ? (1.0f / 1024.0f) : 1.0f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int k = lane + 64 * r;
                cf X = rfft_split_bin(k, v[r], s, a.plan.tw1024);
                const float sk = k == 0 ? se : si;
                X = mk(X.x * sk, (MODE == AN_ADJ && k == 0) ? 0.f : X.y * sk);
                out[k] = X;
            }
            if (lane == 0) out[512] = mk(rfft_split_nyquist(s) * se, 0.f);
        }
        if (!FULLOUT) {
            // band bins k = band_lo + f, f < nband (<= 256)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int k = lane + 64 * r;
                const int f = k - band_lo;
                if (f < 0 || f >= kFS) continue;
                const size_t idx = row * kFS + f;
                if (f >= nband) {
                    if (MODE == AN_NORM) {
                        if (a.mag) a.mag[idx] = 0.f;
                        if (a.unit) a.unit[idx] = mk(0.f, 0.f);
                    }
                    continue;
                }
                cf X = rfft_split_bin(k, v[r], s, a.plan.tw1024);
                if (MODE == AN_NORM) {
                    const float mg = fast_sqrt(X.x * X.x + X.y * X.y);
                    const float im = fast_rcp(mg);
                    if (a.mag) a.mag[idx] = mg;
                    if (a.unit) a.unit[idx] = (mg > 0.f) ? mk(X.x * im, X.y * im) : mk(a.unit_default, 0.f);
                } else {
                    // dL/dc = Re(G conj P) with G = (2/N) rfft(.)  [adjoint of irfft on interior bins]
                    const bool pre = pre_ok && r < 5;
                    cf P = pre ? preP[r < 5 ? r : 0] : a.phasor[idx];
                    float g = (X.x * P.x + X.y * P.y) * (1.0f / 512.0f);
                    if (a.grad_out) a.grad_out[idx] = g;
                    if (a.do_step) {
                        // torch.optim.NAdam single-tensor step + clamp + best snapshot
                        float mo, ve, p, blo, bhi;
                        if (pre) { mo = preM[r < 5 ? r : 0]; ve = preV[r < 5 ? r : 0]; p = preC[r < 5 ? r : 0];
                                   blo = preL[r < 5 ? r : 0]; bhi = preH[r < 5 ? r : 0]; }
                        else { mo = a.mom[idx]; ve = a.vel[idx]; p = a.coef[idx]; blo = a.lo[idx]; bhi = a.hi[idx]; }
                        nadam_clamp_update(p, mo, ve, g, blo, bhi, sc.x, sc.y, inv_bc2, a.hyp);
                        a.mom[idx] = mo; a.vel[idx] = ve; a.coef[idx] = p;
                        if (improved) a.best[idx] = p;
                    }
                }
            }
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------------
// Synthesis: spectrum -> inverse real FFT -> window -> overlap-add -> epilogue
// ---------------------------------------------------------------------------------

// INP: 0 = full one-sided spectrum, 1 = band inside bins 1..256 (prefetched), 2 = any band
template <int MODE, int INP>
__global__ __launch_bounds__(kThreads) void synth_kernel(SynthArgs a) {
    __shared__ float ola[kSynthChunk];
    __shared__ cf scratch[4][kFftScratch];
    __shared__ unsigned long long red[4];
    __shared__ double dred[4];

    const int b = blockIdx.y;
    const int f0 = a.frame_off[b];
    const int T = a.frame_off[b + 1] - f0;
    const int nblk = T - 1;                     // output hop blocks
    int nseg, jb0, jb1;
    synth_segment(nblk, blockIdx.x, a.run_blocks, nseg, jb0, jb1);
    if ((int)blockIdx.x >= nseg || T < 1) return;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave index is wave-uniform: keep it scalar

    // frames that touch the padded range this workgroup needs
    int tlo = max(0, jb0 - 1);
    int thi = min(T - 1, jb1 + 1);
    if (MODE == SY_ADJ) {
        if (blockIdx.x == 0) tlo = 0;
        if ((int)blockIdx.x == nseg - 1) thi = T - 1;
    }
    const int nfr = thi - tlo + 1;              // <= kSynthBlocks + 3 = 19 by construction
    const int pbase = kHop * tlo;               // padded position of ola[0]

    ClipNorm cn;
    cn.m = 1.f; cn.m2 = 1.f; cn.k = 0;
    if (MODE == SY_ADJ && a.pmax_in) cn = clip_norm_from_partials(a.pmax_in + (size_t)b * a.pstride, a.pcount[b], red);

    // twiddles live in LDS in this kernel (28 VGPRs less per lane -> one more wave per SIMD)
    __shared__ cf tw1s[512];
    __shared__ cf tw2s[64];
    fft_fill_tables(tid, kThreads, a.plan.tw512, tw1s, tw2s);
    for (int i = tid; i < kSynthChunk; i += kThreads) ola[i] = 0.f;
    __syncthreads();

    cf* s = scratch[wave];
    // irfft's 1/1024 (1/2 in the merge, 1/512 here); the adjoint of the forward rfft is 512*irfft
    const float scale = (MODE == SY_FWD) ? (1.0f / 512.0f) : 1.0f;
    const float2* win2p = reinterpret_cast<const float2*>(a.plan.window);   // 4 KB, L1-resident
    const int band_lo = a.plan.band_lo, nband = a.plan.nband;

    // kSynthRounds rounds; in round r wave w owns frame r + kSynthRounds*w: concurrently processed frames are
    // at least 4 hops = 1024 samples apart, so the overlap-add needs no atomics and its
    // summation order is fixed.
    // Band inputs of one frame.  When the band lies inside bins 1..256 (the model card: 32..256)
    // register slot r needs exactly one input bin: its own bin k = lane+64r for k <= 256, the
    // partner bin 512-k otherwise (k = 256 is its own partner).  Those 3 floats per slot are
    // prefetched: the loads for round r4+1 are issued before round r4's transform.
    constexpr bool full_in = (INP == 0);
    constexpr bool compact = (INP == 1);
    float inA[8];
    cf inP[8];
    auto load_band = [&](int fi) {
        const size_t row = (size_t)(f0 + tlo + fi);
        const float* A = a.amp + row * kFS;
        const cf* P = a.ph + row * kFS;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            // unconditional loads from a clamped index (the row always has kFS entries); bins
            // outside the band are zeroed by the amplitude
            const int k = lane + 64 * r;
            const int f = ((k <= 256) ? k : 512 - k) - band_lo;
            const int fc_ = min(max(f, 0), kFS - 1);
            const float am = A[fc_];
            inP[r] = P[fc_];
            inA[r] = (f >= 0 && f < nband) ? am : 0.f;
        }
    };
    cf mc[8];                                       // merge constants of this lane's eight bins (compact input only)
    if (compact) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const cf w = a.plan.tw1024[lane + 64 * r];                  // (cos t, -sin t), t = 2 pi k / 1024
            mc[r] = (r < 4) ? mk(0.5f * (1.f + w.y), 0.5f * w.x) : mk(0.5f * (1.f - w.y), -0.5f * w.x);
        }
    }
    if (compact && kSynthRounds * wave < nfr) load_band(kSynthRounds * wave);
#pragma unroll 1
    for (int r4 = 0; r4 < kSynthRounds; ++r4) {
        const int fi = r4 + kSynthRounds * wave;
        if (fi < nfr) {
            const size_t row = (size_t)(f0 + tlo + fi);
            cf v[8];
            if (full_in) {
                const cf* X = a.full + row * 520;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int k = lane + 64 * r;
                    cf xk = X[k], xp = X[512 - k];
                    if (k == 0) { xk.y = 0.f; xp.y = 0.f; }      // C2R ignores Im of DC / Nyquist
                    // adjoint of the forward rfft = 512*irfft on the interior bins; DC / Nyquist are not doubled by
                    // irfft, so they enter twice as large
                    if (MODE == SY_ADJ && k == 0) { xk.x *= 2.f; xp.x *= 2.f; }
                    v[r] = irfft_merge_bin(k, xk, xp, a.plan.tw1024);
                }
            } else if (compact) {
                // irfft merge with one of the two inputs known to be zero (band inside bins 1..256):
                //   k < 256:  Z[k] = X[k] * (1 + i conj W^k)/2        k >= 256:  Z[k] = conj(X[512-k]) * (1 - i conj W^k)/2
                // (k = 256 is its own partner and both forms give conj X[256]); the constants sit in registers
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float xr = inA[r] * inP[r].x, xi = (r < 4) ? inA[r] * inP[r].y : -(inA[r] * inP[r].y);
                    v[r] = mk(xr * mc[r].x - xi * mc[r].y, xr * mc[r].y + xi * mc[r].x);
                }
                if (r4 < kSynthRounds - 1 && fi + 1 < nfr) load_band(fi + 1);
            } else {
                const float* A = a.amp + row * kFS;
                const cf* P = a.ph + row * kFS;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int k = lane + 64 * r;
                    const int f = k - band_lo, fp = (512 - k) - band_lo;
                    cf xk = mk(0.f, 0.f), xp = mk(0.f, 0.f);
                    if (f >= 0 && f < nband) { float am = A[f]; cf p = P[f]; xk = mk(am * p.x, am * p.y); }
                    if (fp >= 0 && fp < nband) { float am = A[fp]; cf p = P[fp]; xp = mk(am * p.x, am * p.y); }
                    v[r] = irfft_merge_bin(k, xk, xp, a.plan.tw1024);
                }
            }
            fft512_wave_t<1>(lane, v, tw1s, tw2s, s);
            float2* o2 = reinterpret_cast<float2*>(ola + kHop * fi);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float2 q = o2[lane + 64 * r];
                const float2 w = win2p[lane + 64 * r];
                q.x += v[r].x * (w.x * scale);
                q.y += v[r].y * (w.y * scale);
                o2[lane + 64 * r] = q;
            }
        }
        __syncthreads();
    }

    // ---- output pass ------------------------------------------------------------------
    const int Ny = kHop * nblk;
    float* out = a.out + sig_offset(a.frame_off, b);
    const int j0 = kHop * jb0, j1 = kHop * jb1;
    constexpr int NOUT = kSynthBlocks;              // (j1 - j0) / 256 <= 16 passes; loads batched ahead of their use
    const bool small_t = T < 4;
    if (MODE == SY_FWD) {
        const float* add = a.add ? a.add + sig_offset(a.frame_off, b) : nullptr;
        unsigned long long best = 0;
        float addv[NOUT], envv[NOUT];
#pragma unroll
        for (int it = 0; it < NOUT; ++it) {
            const int j = j0 + tid + kThreads * it;
            const int jc = j < j1 ? j : j0;
            envv[it] = small_t ? 1.f : a.plan.env_tab[ola_envelope_index(kHalf + jc, T)];
            addv[it] = add ? add[jc] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NOUT; ++it) {
            const int j = j0 + tid + kThreads * it;
            if (j < j1) {
                const int p = kHalf + j;
                float v = ola[p - pbase] * fast_rcp(small_t ? ola_envelope_loop(a.plan.window2, p, T) : envv[it]);
                if (add) v += addv[it];
                out[j] = v;
                best = umax64(best, pack_max(fabsf(v), (unsigned)j));
            }
        }
        if (a.pmax) {
            best = wave_max64(best);
            if (lane == 0) red[wave] = best;
            __syncthreads();
            if (tid == 0) a.pmax[(size_t)b * a.pstride + blockIdx.x] = umax64(umax64(red[0], red[1]), umax64(red[2], red[3]));
        }
    } else if (a.sig_len) {
        // backward of torch.stft(center=True) for a clip of ANY length n = 256 (T-1) + r, r < 256 (utils/audio/stft.py:27-28
        // under autograd): the padded signal has n + 1024 samples, the frames cover padded positions [0, 256 (T-1) + 1024);
        // the left pad (p < 512) mirrors about sample 0, the right pad (p >= n + 512) about sample n - 1.  The first / last
        // segment of a clip holds its pad and every sample the pad folds onto (segments are >= 3 hop blocks long).
        const int n = a.sig_len[b];
        float* og = a.out + a.sig_off[b];
        const int j1e = ((int)blockIdx.x == nseg - 1) ? n : j1;
        const int pend = kHop * (T - 1) + kNfft;                 // end of the frames' cover
        for (int j = j0 + tid; j < j1e; j += kThreads) {
            float g = ola[kHalf + j - pbase];
            if (j >= 1 && j <= kHalf) g += ola[kHalf - j - pbase];
            const int p = 2 * (n - 1) - j + kHalf;                // the pad position that reflects onto sample j
            if (j <= n - 2 && p >= n + kHalf && p < pend) g += ola[p - pbase];
            og[j] = g;
        }
    } else {
        // adjoint of reflect padding: fold the two 512-sample pads back, then the partial
        // dot product with the normalised forward signal for the normaliser's backward
        // (a.yraw null: plain STFT backward of the plug-in seam -- no normaliser in front, no dot product)
        const float* y = a.yraw ? a.yraw + sig_offset(a.frame_off, b) : nullptr;
        const float inv_m = 1.0f / cn.m, inv_m2 = 1.0f / cn.m2;
        double acc = 0.0;
        float yv[NOUT];
#pragma unroll
        for (int it = 0; it < NOUT; ++it) {
            const int j = j0 + tid + kThreads * it;
            yv[it] = y ? y[j < j1 ? j : j0] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NOUT; ++it) {
            const int j = j0 + tid + kThreads * it;
            if (j < j1) {
                float g = ola[kHalf + j - pbase];
                if (j >= 1 && j <= kHalf) g += ola[kHalf - j - pbase];
                if (j >= Ny - kHalf - 1 && j <= Ny - 2) g += ola[2 * Ny + kHalf - 2 - j - pbase];
                out[j] = g;
                float y2 = (yv[it] * inv_m) * inv_m2;
                acc += (double)g * (double)y2;
            }
        }
        acc = wave_sum_d(acc);
        if (lane == 0) dred[wave] = acc;
        __syncthreads();
        if (tid == 0 && a.pdot) a.pdot[(size_t)b * a.pstride + blockIdx.x] = dred[0] + dred[1] + dred[2] + dred[3];
    }
}

// ---------------------------------------------------------------------------------
// small elementwise kernels of the embed set-up / tear-down
// ---------------------------------------------------------------------------------
// bounds (multibit_embedder.py:157-160, :89-90) and optimiser state reset
__global__ void embed_prepare_kernel(const float* __restrict__ c0, float* __restrict__ coef, float* __restrict__ lo,
                                     float* __restrict__ hi, float* __restrict__ mom, float* __restrict__ vel,
                                     float* __restrict__ best, float* __restrict__ c0_keep, float ratio, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float c = c0[i];
    coef[i] = c;
    box_bounds(c, ratio, lo[i], hi[i]);
    if (c0_keep) c0_keep[i] = c;
    mom[i] = 0.f;
    vel[i] = 0.f;
    best[i] = c;
}

// y_oob[j] = x[j]/m - band[j]   (constant out-of-band part of the synthesis)
__global__ void oob_residual_kernel(const float* __restrict__ audio, const int* __restrict__ in_off,
                                    const unsigned long long* __restrict__ pmax, const int* __restrict__ pcount, int pstride,
                                    const float* __restrict__ band, const int* __restrict__ frame_off, float* __restrict__ oob) {
    __shared__ unsigned long long red[4];
    const int b = blockIdx.y;
    ClipNorm cn = clip_norm_from_partials(pmax + (size_t)b * pstride, pcount[b], red);
    const int Ny = kHop * (frame_off[b + 1] - frame_off[b] - 1);
    const int so = sig_offset(frame_off, b);
    const float* x = audio + in_off[b];
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < Ny; j += gridDim.x * blockDim.x)
        oob[so + j] = x[j] / cn.m - band[so + j];
}

// final waveform: normalise once and rescale by the caller's signed max
// (multibit_embedder.py:185-194 + service/embed.py:69,73)
__global__ void finish_kernel(const float* __restrict__ yraw, const int* __restrict__ frame_off,
                              const unsigned long long* __restrict__ pmax, const int* __restrict__ pcount, int pstride,
                              const float* __restrict__ rescale, float* __restrict__ out, const int* __restrict__ out_off) {
    __shared__ unsigned long long red[4];
    const int b = blockIdx.y;
    ClipNorm cn = clip_norm_from_partials(pmax + (size_t)b * pstride, pcount[b], red);
    const int Ny = kHop * (frame_off[b + 1] - frame_off[b] - 1);
    const int so = sig_offset(frame_off, b);
    const float r = rescale ? rescale[b] : 1.f;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < Ny; j += gridDim.x * blockDim.x)
        out[out_off[b] + j] = r * (yraw[so + j] / cn.m);
}

// ---------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------
static inline dim3 grid2(int x, int y) { return dim3((unsigned)x, (unsigned)y, 1); }

void launch_absmax_partials(const float* sig, const int* sig_off, const int* sig_len, unsigned long long* pmax,
                            int pstride, int B, int max_len, hipStream_t st) {
    int nx = (max_len + 4095) / 4096;
    hipLaunchKernelGGL(absmax_partial_kernel, grid2(nx, B), dim3(kThreads), 0, st, sig, sig_off, sig_len, pmax, pstride);
}

void launch_analysis(const AnalysisLaunch& L, hipStream_t st) {
    AnalysisArgs a{};
    a.plan = L.plan;
    a.frame_off = L.frame_off;
    a.sig = L.sig; a.sig_off = L.sig_off; a.sig_len = L.sig_len;
    a.pmax = L.pmax; a.pcount = L.pcount; a.pstride = L.pstride;
    a.double_norm = L.double_norm; a.unit_default = L.unit_default;
    a.mag = L.mag; a.unit = (cf*)L.unit; a.full = (cf*)L.full;
    a.yraw = L.yraw; a.pdot = L.pdot; a.phasor = (const cf*)L.phasor;
    a.coef = L.coef; a.mom = L.mom; a.vel = L.vel; a.lo = L.lo; a.hi = L.hi; a.best = L.best;
    a.improved = L.improved; a.sched = (const float4*)L.sched; a.sched_len = L.sched_len > 0 ? L.sched_len : 1; a.step = L.step;
    a.grad_out = L.grad_out; a.do_step = L.do_step;
    a.hyp = make_float4(L.hyp[0], L.hyp[1], L.hyp[2], L.hyp[3]);
    int nx = (L.max_frames + kFramesPerWG - 1) / kFramesPerWG;
    if (L.adjoint && L.full)
        hipLaunchKernelGGL((analysis_kernel<AN_ADJ, true>), grid2(nx, L.B), dim3(kThreads), 0, st, a);
    else if (L.adjoint)
        hipLaunchKernelGGL((analysis_kernel<AN_ADJ, false>), grid2(nx, L.B), dim3(kThreads), 0, st, a);
    else if (L.full)
        hipLaunchKernelGGL((analysis_kernel<AN_NORM, true>), grid2(nx, L.B), dim3(kThreads), 0, st, a);
    else
        hipLaunchKernelGGL((analysis_kernel<AN_NORM, false>), grid2(nx, L.B), dim3(kThreads), 0, st, a);
}

void launch_synth(const SynthLaunch& L, hipStream_t st) {
    SynthArgs a{};
    a.plan = L.plan;
    a.frame_off = L.frame_off;
    a.amp = L.amp; a.ph = (const cf*)L.ph; a.full = (const cf*)L.full;
    a.out = L.out; a.add = L.add; a.pmax = L.pmax; a.pstride = L.pstride;
    a.yraw = L.yraw; a.pmax_in = L.pmax_in; a.pcount = L.pcount; a.pdot = L.pdot;
    a.sig_off = L.sig_off; a.sig_len = L.sig_len;
    int nblk = L.max_frames - 1;
    a.run_blocks = (L.run_blocks >= 1 && L.run_blocks <= kSynthBlocks) ? L.run_blocks : kSynthBlocks;
    int nx = (nblk + a.run_blocks - 1) / a.run_blocks;
    if (nx < 1) nx = 1;
    const bool compact = L.plan.band_lo >= 1 && L.plan.band_lo + L.plan.nband <= 257;
    if (L.adjoint && L.full) {
        hipLaunchKernelGGL((synth_kernel<SY_ADJ, 0>), grid2(nx, L.B), dim3(kThreads), 0, st, a);
    } else if (L.adjoint) {
        if (compact) hipLaunchKernelGGL((synth_kernel<SY_ADJ, 1>), grid2(nx, L.B), dim3(kThreads), 0, st, a);
        else hipLaunchKernelGGL((synth_kernel<SY_ADJ, 2>), grid2(nx, L.B), dim3(kThreads), 0, st, a);
    } else if (L.full) {
        hipLaunchKernelGGL((synth_kernel<SY_FWD, 0>), grid2(nx, L.B), dim3(kThreads), 0, st, a);
    } else {
        if (compact) hipLaunchKernelGGL((synth_kernel<SY_FWD, 1>), grid2(nx, L.B), dim3(kThreads), 0, st, a);
        else hipLaunchKernelGGL((synth_kernel<SY_FWD, 2>), grid2(nx, L.B), dim3(kThreads), 0, st, a);
    }
}

void launch_embed_prepare(const float* c0, float* coef, float* lo, float* hi, float* mom, float* vel, float* best,
                          float* c0_keep, float ratio, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(embed_prepare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c0, coef, lo, hi, mom,
                       vel, best, c0_keep, ratio, n);
}

void launch_oob_residual(const float* audio, const int* in_off, const unsigned long long* pmax, const int* pcount,
                         int pstride, const float* band, const int* frame_off, float* oob, int B, int max_frames,
                         hipStream_t st) {
    int nx = max(1, (kHop * (max_frames - 1) + 1023) / 1024);
    hipLaunchKernelGGL(oob_residual_kernel, grid2(nx, B), dim3(256), 0, st, audio, in_off, pmax, pcount, pstride, band,
                       frame_off, oob);
}

void launch_finish(const float* yraw, const int* frame_off, const unsigned long long* pmax, const int* pcount,
                   int pstride, const float* rescale, float* out, const int* out_off, int B, int max_frames,
                   hipStream_t st) {
    int nx = max(1, (kHop * (max_frames - 1) + 1023) / 1024);
    hipLaunchKernelGGL(finish_kernel, grid2(nx, B), dim3(256), 0, st, yraw, frame_off, pmax, pcount, pstride, rescale,
                       out, out_off);
}

}  // namespace aware
