// Shared device helpers and internal structures for libaware_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft512.hpp"

namespace aware {

// Band-limited spectral arrays are frame-major: [global frame][kFS] with the first
// `nband` entries valid (bin k = band_lo + f) and the tail zero.  kFS = 256 keeps
// every frame row 1 KiB-aligned and makes the row the K dimension of the mel GEMM.
constexpr int kFS = 256;
// longest support (adjacent band columns) of mel filters 0..63 / 64..127 that the analysis kernel with the mel projection
// folded in takes (dsp_stream.hip; the card's bank: 5 and 12)
constexpr int kMelTapsA = 6;
constexpr int kMelTapsB = 12;
constexpr int kFramesPerWG = 16;     // frames transformed by one 256-thread workgroup
constexpr int kSynthBlocks = 16;     // hop blocks of output per synthesis workgroup (needs up to 19 frames: 3 halo)
constexpr int kSynthRounds = 5;      // ... transformed in 5 rounds of 4 waves; 187 blocks of a 3 s clip = 12 workgroups,
                                     // 768 for 64 clips = exactly the 3 resident per CU (13 blocks gave 960: a 25 % tail wave)
constexpr int kSynthChunk = (kSynthBlocks + 6) * kHop;   // overlap-add buffer: 256*(19 - 1) + 1024 samples
constexpr int kChunk = (kFramesPerWG + 3) * kHop;   // 4864 floats of LDS signal / OLA buffer
constexpr int kThreads = 256;

// Device-resident plan tables (created by aware_plan_create).
struct PlanDev {
    const cf* tw512;      // exp(-2 pi i j/512),  j < 512
    const cf* tw1024;     // exp(-2 pi i j/1024), j < 512
    const float* window;  // w[1024]
    const float* window2; // w^2[1024]
    const float* env_tab; // overlap-add envelope: head[768], interior[768 (256 used)], tail[768]
    int band_lo;          // first band bin (32)
    int nband;            // number of band bins (225)
};

// A clip's "istft-length" signal (Ny_b = 256*(T_b-1) samples) lives at float offset
// 256*(frame_off[b] - b) of any per-clip signal array (frame_off: B+1 prefix sums of T_b).
__device__ __forceinline__ int sig_offset(const int* frame_off, int b) { return kHop * (frame_off[b] - b); }

// per-clip maximum of |y| with the first index attaining it, packed so that an
// unsigned max gives (largest value, smallest index)
__device__ __forceinline__ unsigned long long pack_max(float a, unsigned idx) {
    return ((unsigned long long)__float_as_uint(a) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}
__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) { return a > b ? a : b; }

__device__ __forceinline__ unsigned long long wave_max64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned lo = __shfl_xor((unsigned)(v & 0xFFFFFFFFu), o);
        unsigned hi = __shfl_xor((unsigned)(v >> 32), o);
        v = umax64(v, ((unsigned long long)hi << 32) | lo);
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Normaliser state of one clip, rebuilt by every consumer workgroup from the
// per-segment partial maxima (deterministic, no atomics, nothing to reset).
//   m  = max|y| + 1e-8            (WaveformNormalizer, waveform.py:18-19)
//   m2 = max|y/m| + 1e-8          (the second normaliser of the plugin lists)
struct ClipNorm {
    float m, m2;
    unsigned k;      // index of the first sample attaining max|y|
};

// all threads of the block call this; `red` is a __shared__ u64[4]
__device__ __forceinline__ ClipNorm clip_norm_from_partials(const unsigned long long* part, int nseg,
                                                            unsigned long long* red) {
    unsigned long long v = 0;
    for (int i = threadIdx.x; i < nseg; i += blockDim.x) v = umax64(v, part[i]);
    v = wave_max64(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    v = umax64(umax64(red[0], red[1]), umax64(red[2], red[3]));
    __syncthreads();
    ClipNorm c;
    float raw = __uint_as_float((unsigned)(v >> 32));
    c.k = 0xFFFFFFFFu - (unsigned)(v & 0xFFFFFFFFu);
    c.m = raw + 1e-8f;
    c.m2 = raw / c.m + 1e-8f;
    return c;
}

// sum of squared windows at padded position p for a clip of T frames (reference loop form)
__device__ __forceinline__ float ola_envelope_loop(const float* __restrict__ w2, int p, int T) {
    int thi = p >> 8;                    // floor(p/256)
    int tlo = thi - 3;
    if (tlo < 0) tlo = 0;
    if (thi > T - 1) thi = T - 1;
    float e = 0.f;
    for (int t = tlo; t <= thi; ++t) e += w2[p - kHop * t];
    return e;
}

// Same value from the plan's tables (built on the host with the same summation order): the
// envelope only depends on p near the two ends of a clip and on p mod 256 in between.
__device__ __forceinline__ float ola_envelope(const PlanDev& pl, int p, int T) {
    if (T < 4) return ola_envelope_loop(pl.window2, p, T);
    if (p < 768) return pl.env_tab[p];
    const int q = p - kHop * T;
    if (q >= 0) return pl.env_tab[1536 + q];
    return pl.env_tab[768 + (p & 255)];
}

// Index of that table entry (T >= 4), branch-free so that a batch of envelope loads can be issued together;
// clamped to the table for positions a caller masks out anyway.
__device__ __forceinline__ int ola_envelope_index(int p, int T) {
    const int q = p - kHop * T;
    int idx = 768 + (p & 255);
    idx = p < 768 ? p : idx;
    idx = q >= 0 ? 1536 + q : idx;
    return min(max(idx, 0), 2303);
}

// Per-bit loss term and its derivative at the read-out (embedding/losses.py; kinds as AWARE_LOSS_* in aware_hip.h):
// 0 push_extremes :38-42, 1 mse :23-25, 2 hinge :12-14, 3 sign :68-70, 4 push_sigmoid :55-59, 5 ber :90-92 (no
// gradient), 6 push_extremes + L1 on the coefficients (EXTENSION; the L1 part is added by the caller), 7 external:
// `tg` is dL/dp itself.  inv = 1 / n_bits (the losses average over the bits).
__device__ __forceinline__ void loss_term(int kind, float p, float tg, float inv, float& lterm, float& dp) {
    if (kind == 0 || kind == 6) {
        lterm = ((p - tg) * (p - tg) - 0.1f * fabsf(p)) * inv;
        dp = (2.f * (p - tg) - 0.1f * ((p > 0.f) ? 1.f : (p < 0.f ? -1.f : 0.f))) * inv;
    } else if (kind == 1) {
        lterm = (p - tg) * (p - tg) * inv;
        dp = 2.f * (p - tg) * inv;
    } else if (kind == 2) {
        const float h = 1.f - p * tg;
        lterm = (h > 0.f ? h : 0.f) * inv;
        dp = (h > 0.f ? -tg : 0.f) * inv;
    } else if (kind == 3) {
        const float h = -p * tg;
        lterm = (h > 0.f ? h : 0.f) * inv;
        dp = (h > 0.f ? -tg : 0.f) * inv;
    } else if (kind == 4) {
        lterm = ((p - tg) * (p - tg) - 0.1f * fabsf(p - 0.5f)) * inv;
        dp = (2.f * (p - tg) - 0.1f * ((p > 0.5f) ? 1.f : (p < 0.5f ? -1.f : 0.f))) * inv;
    } else if (kind == 5) {
        const float sp = (p > 0.f) ? 1.f : (p < 0.f ? -1.f : 0.f), st_ = (tg > 0.f) ? 1.f : (tg < 0.f ? -1.f : 0.f);
        lterm = (sp != st_ ? 1.f : 0.f) * inv;
        dp = 0.f;
    } else {
        lterm = 0.f;
        dp = tg;
    }
}

// balanced split of `nblk` hop blocks into nseg = ceil(nblk / run_blocks) segments (run_blocks <= kSynthBlocks: the
// batch's choice, aware_batch::synth_run -- shorter runs when few clips would leave the chip empty)
__device__ __forceinline__ void synth_segment(int nblk, int seg, int run_blocks, int& nseg, int& jb0, int& jb1) {
    nseg = (nblk + run_blocks - 1) / run_blocks;
    if (nseg < 1) nseg = 1;
    int base = nblk / nseg, rem = nblk % nseg;
    jb0 = seg * base + (seg < rem ? seg : rem);
    jb1 = jb0 + base + (seg < rem ? 1 : 0);
}

}  // namespace aware
