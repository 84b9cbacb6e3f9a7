// Streaming wave kernels for the framed STFT / iSTFT and their adjoints (gfx950, MI355X).
//
// The workgroup-staged kernels of dsp_kernels.hip put a 16-frame chunk of signal (analysis) or a 16-hop
// overlap-add buffer (synthesis) into LDS and transform it with four waves in lock step: a barrier per round,
// 38-46 KB of LDS and 128-168 VGPRs per workgroup, three waves per SIMD.  Measured on MI355X they are latency bound
// (26 % issue-active, 55 % of the wave's life parked in s_waitcnt / s_barrier): every wave of a workgroup meets the
// same global-load and LDS latencies at the same time.
//
// Here ONE WAVE streams a run of consecutive frames of one clip and nothing is shared between waves except
// read-only tables, so there is no barrier after the table fill:
//   * analysis: consecutive frames overlap by 768 of their 1024 samples -- the wave keeps the frame's samples in
//     16 registers per lane and loads only the new 256-sample quarter per frame (two 8-byte loads per lane, issued
//     before the transform of the current frame);
//   * synthesis: the overlap-add runs in registers.  After the inverse transform lane L / register r holds the
//     samples 2(L+64r), 2(L+64r)+1, so quarter q of the frame is the register pair (2q, 2q+1): a rolling
//     accumulator of three quarters (12 registers) receives the frame, the finished 256-sample hop block leaves as
//     two coalesced 512-byte stores, frames are added in ascending order (fixed summation order, no atomics);
//   * the reflect-padding fold of the synthesis adjoint is moved to its consumer: the synthesis adjoint writes the four
//     pad blocks of a clip to a 4 KB side buffer and the analysis adjoint adds them back on load (only the frames at
//     the two ends of a clip take that path), which keeps the run of a wave strictly streaming;
//   * the bin k <-> 512-k pairing of the real-FFT split uses cross-lane reads (ds_bpermute: lane 64-L holds the
//     partner) instead of a third pass through LDS memory.
// Per workgroup of four waves: 27-31 KB of LDS (FFT exchange scratch + twiddle / window / merge tables), no barrier in
// the loop, registers sized for five to six waves per SIMD.
//
// Same reference operations as dsp_kernels.hip (utils/audio/stft.py:27-28,47-48,54-55,61-62, utils/audio/waveform.py:18-19,
// embedding/multibit_embedder.py:95-122); serves plans whose band lies inside bins 1..256 (the model card: 32..256).
#include "common.hpp"
#include "dsp_args.hpp"
#include "kernels.h"

namespace aware {

constexpr int kSW = kStreamWaves;      // waves per workgroup (they share only the read-only tables)
constexpr int kSThreads = 64 * kSW;

// per-clip normaliser state from the per-run partial maxima, reduced by ONE wave (no barrier)
__device__ __forceinline__ ClipNorm clip_norm_wave(const unsigned long long* part, int nseg, int lane) {
    unsigned long long v = 0;
    for (int i = lane; i < nseg; i += 64) v = umax64(v, part[i]);
    v = wave_max64(v);
    ClipNorm c;
    const float raw = __uint_as_float((unsigned)(v >> 32));
    c.k = 0xFFFFFFFFu - (unsigned)(v & 0xFFFFFFFFu);
    c.m = raw + 1e-8f;
    c.m2 = raw / c.m + 1e-8f;
    return c;
}
__device__ __forceinline__ double dot_wave(const double* part, int nseg, int lane) {
    double s = 0.0;
    for (int i = lane; i < nseg; i += 64) s += part[i];
    return wave_sum_d(s);
}

// reciprocal envelope of the two hop halves in the interior of a clip (padded position p >= 768, p < 256 T):
// env_tab[768 + (p & 255)], p = 2 lane (+1) and 128 + 2 lane (+1)
struct LaneEnv {
    float2 lo, hi;
};
__device__ __forceinline__ LaneEnv lane_env(const PlanDev& pl, int lane) {
    LaneEnv e;
    e.lo = make_float2(fast_rcp(pl.env_tab[768 + 2 * lane]), fast_rcp(pl.env_tab[768 + 2 * lane + 1]));
    e.hi = make_float2(fast_rcp(pl.env_tab[768 + 128 + 2 * lane]), fast_rcp(pl.env_tab[768 + 128 + 2 * lane + 1]));
    return e;
}
__device__ __forceinline__ float env_at(const PlanDev& pl, int p, int T) {
    return T < 4 ? ola_envelope_loop(pl.window2, p, T) : pl.env_tab[ola_envelope_index(p, T)];
}

// ---------------------------------------------------------------------------------------------------------
// Analysis: a wave transforms frames t0 .. t0 + nfr - 1 of one clip
// ---------------------------------------------------------------------------------------------------------
// L1: the push_extremes + L1 objective (EXTENSION) adds l1_weight * sign(c - c0) / (nband T) to the gradient
// MELF (AN_NORM only): the magnitudes go through the mel filter bank here -- a filter is a run of at most kMelTapsB adjacent
// band columns, 450 non-zero weights in all for the card's bank -- and the frame's 128 mel values are written instead of its
// 256 magnitudes (the dense K = 256 GEMM of the mel block and its operand never exist)
template <int MODE, bool L1, bool MELF = false>
__global__ __launch_bounds__(kSThreads, MODE == AN_ADJ ? 3 : 4) void analysis_stream_kernel(AnalysisArgs a, int run_frames) {
    __shared__ float magrow_s[MELF ? kSW : 1][MELF ? kFS : 1];
    __shared__ cf tw1s[512];
    __shared__ cf tw2s[64];
    __shared__ float2 wins[512];
    __shared__ cf scratch[kSW][kFftScratch];
    const int tid = threadIdx.x;
    fft_fill_tables(tid, kSThreads, a.plan.tw512, tw1s, tw2s);
    for (int i = tid; i < 512; i += kSThreads) wins[i] = make_float2(a.plan.window[2 * i], a.plan.window[2 * i + 1]);
    __syncthreads();

    int b = blockIdx.y, wgx = blockIdx.x;
    if (a.wg_tab) { const int e = a.wg_tab[blockIdx.x]; b = e >> 12; wgx = e & 4095; }
    const int f0 = a.frame_off[b];
    const int T = a.frame_off[b + 1] - f0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int t0 = (wgx * kSW + wave) * run_frames;
    if (t0 >= T) return;
    const int nfr = min(run_frames, T - t0);
    const int n = a.sig_len[b];
    const float* x = a.sig + a.sig_off[b];
    const bool al8 = (reinterpret_cast<uintptr_t>(x) & 7) == 0;

    float m = 1.f, m2 = 1.f;
    unsigned kmax = 0xFFFFFFFFu;
    if (a.pmax) {
        const ClipNorm cn = clip_norm_wave(a.pmax + (size_t)b * a.pstride, a.pcount[b], lane);
        m = cn.m;
        m2 = (MODE == AN_ADJ || a.double_norm) ? cn.m2 : 1.f;
        kmax = cn.k;
    }
    const float inv_m = 1.0f / m, inv_m2 = 1.0f / m2, inv_mm2 = 1.0f / (m * m2);
    float corr = 0.f;                       // (sum_j g2[j] y2[j]) * sign(y[kmax]): the normalisers' backward at the max sample
    const float* gpL = nullptr;
    const float* gpR = nullptr;
    LaneEnv le;
    le.lo = le.hi = make_float2(1.f, 1.f);
    if (MODE == AN_ADJ) {
        const float adot = (float)dot_wave(a.pdot + (size_t)b * a.pstride, a.pcount[b], lane);
        const float yk = a.yraw[a.sig_off[b] + kmax];
        corr = adot * ((yk > 0.f) ? 1.f : ((yk < 0.f) ? -1.f : 0.f));
        gpL = a.gpad + (size_t)b * 1024;
        gpR = gpL + 512;
        le = lane_env(a.plan, lane);
    }

    // 128 padded samples starting at P (a multiple of 128): lane L gets positions P + 2L, P + 2L + 1
    auto load_half = [&](int P) -> float2 {
        const int s_lo = P - kHalf;                  // first source sample of the wave
        float2 v;
        if (MODE == AN_NORM) {
            if (al8 && s_lo >= 0 && s_lo + 128 <= n) {
                v = *reinterpret_cast<const float2*>(x + s_lo + 2 * lane);
            } else {                                 // reflect padding of torch.stft(center=True) at the two clip ends
                int s0 = s_lo + 2 * lane, s1 = s0 + 1;
                s0 = s0 < 0 ? -s0 : s0;
                s1 = s1 < 0 ? -s1 : s1;
                s0 = s0 >= n ? 2 * (n - 1) - s0 : s0;
                s1 = s1 >= n ? 2 * (n - 1) - s1 : s1;
                v.x = x[min(max(s0, 0), n - 1)];
                v.y = x[min(max(s1, 0), n - 1)];
            }
            if (a.pmax) {
                v.x = v.x * inv_m;
                v.y = v.y * inv_m;
                if (a.double_norm) { v.x = v.x * inv_m2; v.y = v.y * inv_m2; }
            }
            return v;
        }
        // AN_ADJ: adjoint of (reflect pad, trim, / envelope) and of the two stacked normalisers
        if (al8 && T >= 4 && s_lo >= 513 && s_lo + 127 <= n - 514 && P >= 768 && P + 127 < kHop * T) {
            float2 g = *reinterpret_cast<const float2*>(x + s_lo + 2 * lane);
            const unsigned s = (unsigned)(s_lo + 2 * lane);
            if (s == kmax) g.x -= corr;
            if (s + 1 == kmax) g.y -= corr;
            g.x = g.x * inv_mm2;
            g.y = g.y * inv_mm2;
            const float2 re = (P & 128) ? le.hi : le.lo;
            return make_float2(g.x * re.x, g.y * re.y);
        }
        float o[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int p = P + 2 * lane + e;
            const int src = p - kHalf;
            const bool valid = src >= 0 && src < n;
            float g = x[min(max(src, 0), n - 1)];
            // reflect-pad parts of the synthesis adjoint (written by synth_stream_kernel<SY_ADJ>), folded in here
            const bool cl = src >= 1 && src <= kHalf;
            const bool cr = src >= n - kHalf - 1 && src <= n - 2;
            const float pl = gpL[min(max(kHalf - src, 0), 511)];
            const float pr = gpR[min(max(n - 2 - src, 0), 511)];
            if (cl) g += pl;
            if (cr) g += pr;
            if ((unsigned)src == kmax) g -= corr;
            g = g * inv_mm2;
            o[e] = valid ? g * fast_rcp(env_at(a.plan, p, T)) : 0.f;
        }
        return make_float2(o[0], o[1]);
    };

    cf* s = scratch[wave];
    const int band_lo = a.plan.band_lo, nband = a.plan.nband;
    cf w1024[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) w1024[r] = a.plan.tw1024[lane + 64 * r];
    float4 sc = make_float4(0.f, 0.f, 1.f, 0.f);
    float inv_bc2 = 1.f;
    int improved = 0;
    if (MODE == AN_ADJ && a.do_step) {
        sc = a.sched[min(max(*a.step - 1, 0), a.sched_len - 1)];   // the read-out kernel already advanced the counter
        inv_bc2 = 1.0f / sc.z;
        improved = a.improved[b];
    }

    const float l1g = (MODE == AN_ADJ && L1) ? a.l1_weight / (float)(nband * T) : 0.f;
    // MELF: this lane's two filters (mel channels lane and lane + 64): weights in registers, first column of the support
    float wA[kMelTapsA], wB[kMelTapsB];
    int sA = 0, sB = 0;
    if (MELF) {
#pragma unroll
        for (int j = 0; j < kMelTapsA; ++j) wA[j] = a.melf_w[(size_t)lane * kMelTapsB + j];
#pragma unroll
        for (int j = 0; j < kMelTapsB; ++j) wB[j] = a.melf_w[(size_t)(lane + 64) * kMelTapsB + j];
        sA = a.melf_s[lane];
        sB = a.melf_s[lane + 64];
        // the tail of the wave's magnitude row (columns nband .. kFS - 1, whatever the band) stays zero for the whole run
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int f = lane + 64 * r - band_lo;
            if (f >= nband && f < kFS) magrow_s[MELF ? wave : 0][f] = 0.f;
        }
    }
    float2 raw[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) raw[r] = load_half(kHop * t0 + 128 * r);

#pragma unroll 1
    for (int fr = 0; fr < nfr; ++fr) {
        const int t = t0 + fr;
        const size_t row = (size_t)(f0 + t);
        // operands of the optimiser epilogue do not depend on the transform: request them first
        cf preP[5];
        float preM[5], preV[5], preC[5], pre0[5];
        if (MODE == AN_ADJ) {
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int f = lane + 64 * r - band_lo;
                const size_t idx = row * kFS + (size_t)min(max(f, 0), kFS - 1);       // clamped, masked at use
                preP[r] = a.phasor[idx];
                if (a.do_step) { preM[r] = a.mom[idx]; preV[r] = a.vel[idx]; preC[r] = a.coef[idx]; }
                else { preM[r] = preV[r] = 0.f; preC[r] = L1 ? a.coef[idx] : 0.f; }
                pre0[r] = (a.do_step || L1) ? a.c0[idx] : 0.f;     // the box [lo, hi] is a function of c0: one operand, not two
            }
        }
        cf v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float2 w = wins[lane + 64 * r];
            v[r] = mk(raw[r].x * w.x, raw[r].y * w.y);
        }
        // the next frame shares three quarters with this one: roll, and request its new quarter now
#pragma unroll
        for (int r = 0; r < 6; ++r) raw[r] = raw[r + 2];
        // (unconditional: behind the last frame of the run the quarter is simply not used -- positions past the clip are
        // reflected / masked by load_half like any other)
        raw[6] = load_half(kHop * (t + 1) + 768);
        raw[7] = load_half(kHop * (t + 1) + 896);
        fft512_wave_t<-1>(lane, v, tw1s, tw2s, s);

        // real-FFT split: X[k] needs Z[k] (own register) and Z[512-k] = lane (64-L)&63, register 7-r (lane 0: own 8-r)
        const int src_lane = (64 - lane) & 63;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            cf zp = mk(__shfl(v[7 - r].x, src_lane), __shfl(v[7 - r].y, src_lane));
            if (lane == 0) zp = (r == 0) ? v[0] : v[8 - r];
            const int k = lane + 64 * r;
            const int f = k - band_lo;
            if (f < 0 || f >= nband) continue;
            const cf zk = v[r];
            const cf e = mk(0.5f * (zk.x + zp.x), 0.5f * (zk.y - zp.y));
            const cf d = mk(0.5f * (zk.x - zp.x), 0.5f * (zk.y + zp.y));
            const cf wd = cmul(w1024[r], d);
            const cf X = mk(e.x + wd.y, e.y - wd.x);
            const size_t idx = row * kFS + f;
            if (MODE == AN_NORM) {
                const float mg = fast_sqrt(X.x * X.x + X.y * X.y);
                const float im = fast_rcp(mg);
                if (MELF) magrow_s[wave][f] = mg;
                else if (a.mag) a.mag[idx] = mg;
                if (a.unit) a.unit[idx] = (mg > 0.f) ? mk(X.x * im, X.y * im) : mk(a.unit_default, 0.f);
            } else {
                // dL/dc = Re(G conj P) with G = (2/N) rfft(.)  [adjoint of irfft on interior bins]
                const cf P = preP[r];
                float g = (X.x * P.x + X.y * P.y) * (1.0f / 512.0f);
                if (L1) {
                    // EXTENSION: + l1_weight * d/dc mean|c - c0| (mean over the clip's nband * T variables; sign(0) = 0)
                    const float dc = preC[r] - pre0[r];
                    g += l1g * ((dc > 0.f) ? 1.f : ((dc < 0.f) ? -1.f : 0.f));
                }
                if (a.grad_out) a.grad_out[idx] = g;
                if (a.do_step) {
                    // torch.optim.NAdam single-tensor step + clamp + best snapshot (multibit_embedder.py:112-122)
                    float mo = preM[r], ve = preV[r], p = preC[r];
                    float blo, bhi;
                    box_bounds(pre0[r], a.box_ratio, blo, bhi);
                    nadam_clamp_update(p, mo, ve, g, blo, bhi, sc.x, sc.y, inv_bc2, a.hyp);
                    a.mom[idx] = mo; a.vel[idx] = ve; a.coef[idx] = p;
                    if (improved) a.best[idx] = p;
                }
            }
        }
        if (MODE == AN_NORM && MELF) {
            // (columns nband..255 of the wave's magnitude row were zeroed ahead of the loop; LDS operations of one wave complete
            //  in order: no barrier between the stores above and the reads below)
            float* mr = magrow_s[MELF ? wave : 0];
            float m0 = 0.f, m1 = 0.f;
#pragma unroll
            for (int j = 0; j < kMelTapsA; ++j) m0 += wA[j] * mr[sA + j];
#pragma unroll
            for (int j = 0; j < kMelTapsB; ++j) m1 += wB[j] * mr[sB + j];
            float* mo = a.mel_out + row * 128;
            mo[lane] = m0;
            mo[lane + 64] = m1;
        }
        if (MODE == AN_NORM && a.write_pad) {
            // zero tail of the row (columns nband..255); the embed loop keeps it zero from aware_embed_create on
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int f = lane + 64 * r - band_lo;
                if (f >= nband && f < kFS) {
                    const size_t idx = row * kFS + f;
                    if (a.mag) a.mag[idx] = 0.f;
                    if (a.unit) a.unit[idx] = mk(0.f, 0.f);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Synthesis: a wave produces the hop blocks [jb0, jb1) of one clip from frames jb0-1 .. jb1+1
// ---------------------------------------------------------------------------------------------------------
// L1 (SY_FWD only): also emit the per-run sums of |amp - c0| for the loss value of the push_extremes + L1 objective
// MELG (SY_ADJ only): the amplitudes are dL/d|S| = (dL/dmel) * melB, expanded here from the 128 mel gradients of the frame
// through the filter bank's two taps per bin (the dense K = 128 GEMM and its [NF][256] result never exist)
template <int MODE, bool L1, bool MELG = false>
__global__ __launch_bounds__(kSThreads, 4) void synth_stream_kernel(SynthArgs a) {
    __shared__ float2 melw_s[MELG ? kFS : 1];
    __shared__ float melrow_s[MELG ? kSW : 1][MELG ? 128 : 1];
    __shared__ cf tw1s[512];
    __shared__ cf tw2s[64];
    __shared__ float2 wins[512];             // window * irfft scale, as sample pairs
    __shared__ cf mcs[512];                  // irfft merge constants of bin k (band inside bins 1..256)
    __shared__ cf scratch[kSW][kFftScratch];
    const int tid = threadIdx.x;
    // irfft's 1/1024 (1/2 in the merge, 1/512 here); the adjoint of the forward rfft is 512*irfft
    const float scale = (MODE == SY_FWD) ? (1.0f / 512.0f) : 1.0f;
    fft_fill_tables(tid, kSThreads, a.plan.tw512, tw1s, tw2s);
    for (int i = tid; i < 512; i += kSThreads) {
        wins[i] = make_float2(a.plan.window[2 * i] * scale, a.plan.window[2 * i + 1] * scale);
        const cf w = a.plan.tw1024[i];                                   // (cos t, -sin t), t = 2 pi k / 1024
        mcs[i] = (i < 256) ? mk(0.5f * (1.f + w.y), 0.5f * w.x) : mk(0.5f * (1.f - w.y), -0.5f * w.x);
    }
    if (MELG)
        for (int i = tid; i < kFS; i += kSThreads) melw_s[i] = a.melw[i];
    __syncthreads();

    int b = blockIdx.y, wgx = blockIdx.x;
    if (a.wg_tab) { const int e = a.wg_tab[blockIdx.x]; b = e >> 12; wgx = e & 4095; }
    const int f0 = a.frame_off[b];
    const int T = a.frame_off[b + 1] - f0;
    const int nblk = T - 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int run = wgx * kSW + wave;
    int nseg, jb0, jb1;
    synth_segment(nblk, run, a.run_blocks, nseg, jb0, jb1);
    if (run >= nseg || T < 1) return;
    const bool first = jb0 == 0, last = jb1 == nblk;
    const int t_lo = max(jb0 - 1, 0);
    // the last run of the adjoint keeps shifting with zero frames until the two right pad blocks have left
    const int t_hi = (MODE == SY_ADJ && last) ? T + 2 : jb1 + 1;
    const int Ny = kHop * nblk;
    const int so = sig_offset(a.frame_off, b);
    float* out = a.out + so;
    const float* add = (MODE == SY_FWD && a.add) ? a.add + so : nullptr;
    const float* y = (MODE == SY_ADJ) ? a.yraw + so : nullptr;
    float inv_m = 1.f, inv_m2 = 1.f;
    if (MODE == SY_ADJ) {
        const ClipNorm cn = clip_norm_wave(a.pmax_in + (size_t)b * a.pstride, a.pcount[b], lane);
        inv_m = 1.0f / cn.m;
        inv_m2 = 1.0f / cn.m2;
    }
    const LaneEnv le = lane_env(a.plan, lane);
    const int band_lo = a.plan.band_lo, nband = a.plan.nband;
    cf* s = scratch[wave];

    // band inputs of one frame: register slot r needs exactly one input bin -- its own bin k = lane+64r for k <= 256,
    // the partner 512-k otherwise; loads are unconditional from a clamped index, the amplitude masks the rest
    float inA[8];
    cf inP[8];
    unsigned fo[8];                          // lane-constant column of slot r (clamped) and whether it lies in the band
    unsigned inband = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = lane + 64 * r;
        const int f = ((k <= 256) ? k : 512 - k) - band_lo;
        fo[r] = (unsigned)min(max(f, 0), kFS - 1);
        if (f >= 0 && f < nband) inband |= 1u << r;
    }
    // MELG: the first mel tap of slot r's column, four per register
    unsigned mtap[2] = {0u, 0u};
    float drow[2] = {0.f, 0.f};              // the frame's 128 mel gradients, two per lane (MELG)
    if (MELG) {
#pragma unroll
        for (int r = 0; r < 8; ++r) mtap[r >> 2] |= (unsigned)a.melm[fo[r]] << (8 * (r & 3));
    }
    auto load_band = [&](int t) {
        const size_t row = (size_t)(f0 + t);
        const float* A = MELG ? nullptr : a.amp + row * kFS;
        const cf* P = a.ph + row * kFS;
        if (MELG) {
            const float* D = a.dmel + row * 128;
            drow[0] = D[lane];
            drow[1] = D[lane + 64];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            inP[r] = P[fo[r]];
            if (!MELG) {
                const float am = A[fo[r]];
                inA[r] = ((inband >> r) & 1u) ? am : 0.f;
            }
        }
    };
    // MELG: this frame's amplitudes from its mel gradients (through the wave's own 512 bytes of LDS; LDS operations of one
    // wave complete in order, so neither a barrier nor a fence is needed)
    auto expand_mel = [&]() {
        float* mr = melrow_s[MELG ? wave : 0];
        mr[lane] = drow[0];
        mr[lane + 64] = drow[1];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const unsigned m = (mtap[r >> 2] >> (8 * (r & 3))) & 0xFFu;
            const float2 w = melw_s[MELG ? fo[r] : 0];
            const float v = w.x * mr[m] + w.y * mr[m + 1];
            inA[r] = ((inband >> r) & 1u) ? v : 0.f;
            if (r == 3) __builtin_amdgcn_sched_barrier(0);      // two rounds of four slots: 16, not 32, registers in flight
        }
    };

    float2 acc[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) acc[q] = make_float2(0.f, 0.f);
    unsigned long long best = 0;
    double dot = 0.0;
    float l1 = 0.f;                          // sum of |amp - c0| over the band bins of the frames this run owns
    if (t_lo <= T - 1) load_band(t_lo);

    // operands of a block's epilogue that do not depend on the transform (requested before it)
    auto epi_operands = [&](int i, bool emit, float2& e0, float2& e1) {
        e0 = make_float2(0.f, 0.f);
        e1 = e0;
        if (emit) {
            const float* src = (MODE == SY_FWD) ? add : y;
            if (src) {
                e0 = *reinterpret_cast<const float2*>(src + kHop * i + 2 * lane);
                e1 = *reinterpret_cast<const float2*>(src + kHop * i + 128 + 2 * lane);
            }
        }
    };
    // overlap-add of one frame's windowed samples c[] and the epilogue of the hop block it completes
    auto ola_emit = [&](int i, bool emit, const float2 (&c)[8], const float2 e0, const float2 e1) {
        // overlap-add in registers: quarter q of the frame = registers 2q, 2q+1; frames arrive in ascending order
        const float2 o0 = make_float2(acc[0].x + c[0].x, acc[0].y + c[0].y);
        const float2 o1 = make_float2(acc[1].x + c[1].x, acc[1].y + c[1].y);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = make_float2(acc[q + 2].x + c[q + 2].x, acc[q + 2].y + c[q + 2].y);
        acc[4] = c[6];
        acc[5] = c[7];

        const int j0 = kHop * i + 2 * lane, j1 = j0 + 128;    // output samples of o0 (.x -> j0, .y -> j0+1) and o1
        if (MODE == SY_FWD) {
            if (emit) {
                float2 r0, r1;
                if (T >= 4 && i >= 1 && i <= T - 3) { r0 = le.lo; r1 = le.hi; }       // padded 768 <= p < 256 T
                else {
                    r0 = make_float2(fast_rcp(env_at(a.plan, kHalf + j0, T)), fast_rcp(env_at(a.plan, kHalf + j0 + 1, T)));
                    r1 = make_float2(fast_rcp(env_at(a.plan, kHalf + j1, T)), fast_rcp(env_at(a.plan, kHalf + j1 + 1, T)));
                }
                float2 v0 = make_float2(o0.x * r0.x, o0.y * r0.y), v1 = make_float2(o1.x * r1.x, o1.y * r1.y);
                if (add) { v0.x += e0.x; v0.y += e0.y; v1.x += e1.x; v1.y += e1.y; }
                *reinterpret_cast<float2*>(out + j0) = v0;
                *reinterpret_cast<float2*>(out + j1) = v1;
                best = umax64(best, pack_max(fabsf(v0.x), (unsigned)j0));
                best = umax64(best, pack_max(fabsf(v0.y), (unsigned)j0 + 1));
                best = umax64(best, pack_max(fabsf(v1.x), (unsigned)j1));
                best = umax64(best, pack_max(fabsf(v1.y), (unsigned)j1 + 1));
            }
        } else {
            if (emit) {
                *reinterpret_cast<float2*>(out + j0) = o0;
                *reinterpret_cast<float2*>(out + j1) = o1;
                dot += (double)o0.x * (double)((e0.x * inv_m) * inv_m2) + (double)o0.y * (double)((e0.y * inv_m) * inv_m2);
                dot += (double)o1.x * (double)((e1.x * inv_m) * inv_m2) + (double)o1.y * (double)((e1.y * inv_m) * inv_m2);
            } else if (first && i < 0 && i >= -2) {
                // left reflect-pad blocks (padded positions 0..511): the analysis adjoint adds pad[p] onto sample 512-p;
                // their share of sum_j g[j] y2[j] is taken here
                float* gp = a.gpad + (size_t)b * 1024;
                const int p0 = kHop * (i + 2) + 2 * lane, p1 = p0 + 128;
                *reinterpret_cast<float2*>(gp + p0) = o0;
                *reinterpret_cast<float2*>(gp + p1) = o1;
                const float ov[4] = {o0.x, o0.y, o1.x, o1.y};
                const int pp[4] = {p0, p0 + 1, p1, p1 + 1};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = kHalf - pp[e];
                    const float yv = y[min(max(j, 0), max(Ny - 1, 0))];
                    if (j >= 1 && j < Ny) dot += (double)ov[e] * (double)((yv * inv_m) * inv_m2);
                }
            } else if (last && i >= nblk && i <= nblk + 1) {
                // right reflect-pad blocks (padded positions Ny+512 ..): pad[u] is added onto sample Ny-2-u
                float* gp = a.gpad + (size_t)b * 1024 + 512;
                const int u0 = kHop * (i - nblk) + 2 * lane, u1 = u0 + 128;
                *reinterpret_cast<float2*>(gp + u0) = o0;
                *reinterpret_cast<float2*>(gp + u1) = o1;
                const float ov[4] = {o0.x, o0.y, o1.x, o1.y};
                const int uu[4] = {u0, u0 + 1, u1, u1 + 1};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = Ny - 2 - uu[e];
                    const float yv = y[min(max(j, 0), max(Ny - 1, 0))];
                    if (j >= 0) dot += (double)ov[e] * (double)((yv * inv_m) * inv_m2);
                }
            }
        }
    };

    const int t_fft = min(t_hi, T - 1);      // frames t_lo..t_fft exist; t_fft+1..t_hi only flush the accumulator
#pragma unroll 1
    for (int t = t_lo; t <= t_fft; ++t) {
        const int i = t - 2;                                  // the hop block that frame t completes
        const bool emit = i >= jb0 && i < jb1;
        float2 e0, e1;
        epi_operands(i, emit, e0, e1);
        float2 c[8];
        {
            if (MELG) expand_mel();
            if (MODE == SY_FWD && L1 && ((t >= jb0 && t < jb1) || (last && t == T - 1))) {
                // own bins of the slots: k = lane + 64 r <= 256 (r < 4, and lane 0 of r = 4)
                const float* C0 = a.c0 + (size_t)(f0 + t) * kFS;
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const int f = lane + 64 * r - band_lo;
                    const float cv = C0[min(max(f, 0), kFS - 1)];
                    if (f >= 0 && f < nband && (r < 4 || lane == 0)) l1 += fabsf(inA[r] - cv);
                }
            }
            cf v[8];
            // irfft merge with one of the two inputs known to be zero (band inside bins 1..256):
            //   k < 256:  Z[k] = X[k] * (1 + i conj W^k)/2        k >= 256:  Z[k] = conj(X[512-k]) * (1 - i conj W^k)/2
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const cf mc = mcs[lane + 64 * r];
                const float xr = inA[r] * inP[r].x, xi = (r < 4) ? inA[r] * inP[r].y : -(inA[r] * inP[r].y);
                v[r] = mk(xr * mc.x - xi * mc.y, xr * mc.y + xi * mc.x);
            }
            load_band(min(t + 1, T - 1));           // unconditional (clamped): no register copies around a branch
            fft512_wave_t<1>(lane, v, tw1s, tw2s, s);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float2 w = wins[lane + 64 * r];
                c[r] = make_float2(v[r].x * w.x, v[r].y * w.y);
            }
        }
        ola_emit(i, emit, c, e0, e1);
    }
#pragma unroll 1
    for (int t = t_fft + 1; t <= t_hi; ++t) {
        const int i = t - 2;
        const bool emit = i >= jb0 && i < jb1;
        float2 e0, e1;
        epi_operands(i, emit, e0, e1);
        float2 c[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) c[r] = make_float2(0.f, 0.f);
        ola_emit(i, emit, c, e0, e1);
    }
    if (MODE == SY_FWD) {
        if (a.pmax) {
            best = wave_max64(best);
            if (lane == 0) a.pmax[(size_t)b * a.pstride + run] = best;
        }
        if (L1) {
            const double tot = wave_sum_d((double)l1);
            if (lane == 0) a.pl1[(size_t)b * a.pstride + run] = tot;
        }
    } else {
        dot = wave_sum_d(dot);
        if (lane == 0) a.pdot[(size_t)b * a.pstride + run] = dot;
    }
}

// L1 part of the loss push_extremes + L1 (EXTENSION): l1term[b] = weight * sum_runs pl1[b][run] / (nband * T_b)
__global__ __launch_bounds__(64) void l1_reduce_kernel(const double* __restrict__ pl1, const int* __restrict__ pcount, int pstride,
                                                        const int* __restrict__ frame_off, int nband, float weight,
                                                        float* __restrict__ l1term) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const double s = dot_wave(pl1 + (size_t)b * pstride, pcount[b], lane);
    const int T = frame_off[b + 1] - frame_off[b];
    if (lane == 0) l1term[b] = weight * (float)(s / (double)(nband * T));
}
void launch_l1_reduce(const double* pl1, const int* pcount, int pstride, const int* frame_off, int nband, float weight,
                      float* l1term, int B, hipStream_t st) {
    hipLaunchKernelGGL(l1_reduce_kernel, dim3(B), dim3(64), 0, st, pl1, pcount, pstride, frame_off, nband, weight, l1term);
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------
bool stream_supported(const PlanDev& plan) { return plan.band_lo >= 1 && plan.band_lo + plan.nband <= 257; }

void launch_analysis_stream(const AnalysisLaunch& L, hipStream_t st) {
    AnalysisArgs a{};
    a.plan = L.plan;
    a.frame_off = L.frame_off;
    a.sig = L.sig; a.sig_off = L.sig_off; a.sig_len = L.sig_len;
    a.pmax = L.pmax; a.pcount = L.pcount; a.pstride = L.pstride;
    a.double_norm = L.double_norm; a.unit_default = L.unit_default;
    a.mag = L.mag; a.unit = (cf*)L.unit; a.full = nullptr;
    a.yraw = L.yraw; a.pdot = L.pdot; a.phasor = (const cf*)L.phasor;
    a.coef = L.coef; a.mom = L.mom; a.vel = L.vel; a.lo = L.lo; a.hi = L.hi; a.best = L.best;
    a.improved = L.improved; a.sched = (const float4*)L.sched; a.sched_len = L.sched_len > 0 ? L.sched_len : 1; a.step = L.step;
    a.grad_out = L.grad_out; a.do_step = L.do_step;
    a.hyp = make_float4(L.hyp[0], L.hyp[1], L.hyp[2], L.hyp[3]);
    a.gpad = L.gpad; a.write_pad = L.write_pad;
    a.c0 = L.c0; a.box_ratio = L.box_ratio; a.l1_weight = L.l1_weight;
    // Frames per wave (run length R).  A run start costs 8 loads per lane, every further frame 2, so long runs are cheaper
    // per frame; but the chip wants ~16 waves per CU, and a workgroup whose last waves have no run idles their slots.
    // Measured on 3 s clips (us, analysis / adjoint): B = 256: R = 8 75/144, 12 67/118, 16 69/136; B = 128: R = 4 40/84,
    // 8 38/79, 12 43/92; B = 64: R = 4 25/43, 8 32/59.  Rule: the longest R in [4, 16] that still gives 4096 waves, then
    // the nearest shorter R whose run count per clip is a multiple of the 4 waves of a workgroup.
    // (a batch handle applies the rule to the sum over its clips and passes R with the flat workgroup table built for it)
    int R = 4;
    dim3 grid;
    if (L.run_frames >= 4 && L.run_frames <= 16 && L.wg_tab && L.n_wg > 0) {
        R = L.run_frames;
        a.wg_tab = L.wg_tab;
        grid = dim3((unsigned)L.n_wg, 1, 1);
    } else {
        for (int cand = 16; cand >= 4; --cand)
            if ((long)L.B * ((L.max_frames + cand - 1) / cand) >= 4096) { R = cand; break; }
        for (int cand = R; cand >= 4 && cand >= R - 3; --cand)
            if (((L.max_frames + cand - 1) / cand) % kSW == 0) { R = cand; break; }
        const int runs = (L.max_frames + R - 1) / R;
        grid = dim3((unsigned)((runs + kSW - 1) / kSW), (unsigned)L.B, 1);
    }
    if (L.adjoint && a.l1_weight != 0.f) hipLaunchKernelGGL((analysis_stream_kernel<AN_ADJ, true>), grid, dim3(kSThreads), 0, st, a, R);
    else if (L.adjoint) hipLaunchKernelGGL((analysis_stream_kernel<AN_ADJ, false>), grid, dim3(kSThreads), 0, st, a, R);
    else if (L.mel_out && L.melf_w && L.melf_s) {
        a.mel_out = L.mel_out; a.melf_w = L.melf_w; a.melf_s = L.melf_s; a.mag = nullptr;
        hipLaunchKernelGGL((analysis_stream_kernel<AN_NORM, false, true>), grid, dim3(kSThreads), 0, st, a, R);
    } else hipLaunchKernelGGL((analysis_stream_kernel<AN_NORM, false>), grid, dim3(kSThreads), 0, st, a, R);
}

void launch_synth_stream(const SynthLaunch& L, hipStream_t st) {
    SynthArgs a{};
    a.plan = L.plan;
    a.frame_off = L.frame_off;
    a.amp = L.amp; a.ph = (const cf*)L.ph; a.full = nullptr;
    a.out = L.out; a.add = L.add; a.pmax = L.pmax; a.pstride = L.pstride;
    a.yraw = L.yraw; a.pmax_in = L.pmax_in; a.pcount = L.pcount; a.pdot = L.pdot; a.gpad = L.gpad;
    a.c0 = L.c0; a.pl1 = (L.c0 && L.pl1) ? L.pl1 : nullptr;
    const int nblk = L.max_frames - 1;
    a.run_blocks = (L.run_blocks >= 1 && L.run_blocks <= kSynthBlocks) ? L.run_blocks : kSynthBlocks;
    int runs = (nblk + a.run_blocks - 1) / a.run_blocks;
    if (runs < 1) runs = 1;
    dim3 grid((unsigned)((runs + kSW - 1) / kSW), (unsigned)L.B, 1);
    if (L.wg_tab && L.n_wg > 0 && a.run_blocks == L.run_blocks) {
        a.wg_tab = L.wg_tab;
        grid = dim3((unsigned)L.n_wg, 1, 1);
    }
    a.dmel = L.dmel; a.melw = (const float2*)L.melw; a.melm = L.melm;
    if (L.adjoint && L.dmel && L.melw && L.melm) hipLaunchKernelGGL((synth_stream_kernel<SY_ADJ, false, true>), grid, dim3(kSThreads), 0, st, a);
    else if (L.adjoint) hipLaunchKernelGGL((synth_stream_kernel<SY_ADJ, false>), grid, dim3(kSThreads), 0, st, a);
    else if (a.pl1) hipLaunchKernelGGL((synth_stream_kernel<SY_FWD, true>), grid, dim3(kSThreads), 0, st, a);
    else hipLaunchKernelGGL((synth_stream_kernel<SY_FWD, false>), grid, dim3(kSThreads), 0, st, a);
}

}  // namespace aware
