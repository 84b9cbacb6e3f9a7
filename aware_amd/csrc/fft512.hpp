// 1024-point real FFT / inverse real FFT for one 64-lane wavefront (gfx950).
//
// A frame is transformed by ONE wave: the 1024 real samples are packed into a
// 512-point complex sequence z[n] = x[2n] + i x[2n+1]; each lane keeps 8 complex
// values in registers and the transform is three radix-8 passes (512 = 8*8*8,
// decimation in frequency) with two transposes through a wave-private LDS
// scratch.  The real-FFT split / merge step pairs bin k with bin 512-k through
// the same scratch.  All cross-lane traffic happens at phase boundaries, so the
// phases are plain per-lane functions: on the device the 64 lanes run them in
// lock-step, and tests/host_sim runs them lane after lane on the CPU.
//
// Register <-> index convention everywhere ("natural order"):
//     lane L, register r  <->  element  L + 64*r     (L = 0..63, r = 0..7)
//
// Replaces torch.stft / torch.istft's per-frame rfft / irfft
// (reference: src/AWARE/utils/audio/stft.py:27-28, :47-48).
#pragma once
#include <hip/hip_runtime.h>

#define AW_HD __host__ __device__ __forceinline__

namespace aware {

constexpr int kNfft = 1024;       // frame length
constexpr int kHop = 256;         // hop
constexpr int kHalf = 512;        // complex FFT size
constexpr int kWave = 64;
// wave-private LDS scratch (float2 units).  Exchange 1 uses k0*72 + l, exchange 2
// uses k0*73 + 8*q0 + n0 (both conflict-free for ds_read_b64), the split/merge
// step uses natural order 0..511.
constexpr int kFftScratch = 584;

struct cf {
    float x, y;
};
AW_HD cf mk(float a, float b) { cf r; r.x = a; r.y = b; return r; }
AW_HD cf operator+(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
AW_HD cf operator-(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
AW_HD cf cmul(cf a, cf b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
AW_HD cf cconj(cf a) { return mk(a.x, -a.y); }
// multiply by DIR*i  (DIR = -1: forward transform, +1: inverse)
template <int DIR> AW_HD cf mul_di(cf a) { return DIR > 0 ? mk(-a.y, a.x) : mk(a.y, -a.x); }

// In-register 8-point DFT: v[k] <- sum_n v[n] * exp(DIR * 2*pi*i * n*k / 8)
template <int DIR> AW_HD void radix8(cf (&v)[8]) {
    const float h = 0.70710678118654752440f;
    cf a0 = v[0] + v[4], a4 = v[0] - v[4];
    cf a1 = v[1] + v[5], a5 = v[1] - v[5];
    cf a2 = v[2] + v[6], a6 = v[2] - v[6];
    cf a3 = v[3] + v[7], a7 = v[3] - v[7];
    // twiddles w8^1, w8^2, w8^3 on the odd half
    cf t5 = mk(h * (a5.x - DIR * a5.y), h * (a5.y + DIR * a5.x));       // a5 * (1 + DIR i)/sqrt2
    cf t6 = mul_di<DIR>(a6);
    cf t7 = mk(h * (-a7.x - DIR * a7.y), h * (-a7.y + DIR * a7.x));     // a7 * (-1 + DIR i)/sqrt2
    // even outputs: DFT-4 of (a0,a1,a2,a3)
    cf c0 = a0 + a2, c2 = a0 - a2, c1 = a1 + a3, c3 = mul_di<DIR>(a1 - a3);
    v[0] = c0 + c1; v[4] = c0 - c1; v[2] = c2 + c3; v[6] = c2 - c3;
    // odd outputs: DFT-4 of (a4,t5,t6,t7)
    cf d0 = a4 + t6, d2 = a4 - t6, d1 = t5 + t7, d3 = mul_di<DIR>(t5 - t7);
    v[1] = d0 + d1; v[5] = d0 - d1; v[3] = d2 + d3; v[7] = d2 - d3;
}

// Per-lane twiddle constants, loaded once per wave from the plan's tables.
//   tw512[j]  = exp(-2*pi*i*j/512)   j = 0..511
//   tw1024[j] = exp(-2*pi*i*j/1024)  j = 0..511
struct FftLaneConst {
    cf t1[8];   // step-1 twiddle  W512^(lane*k0)
    cf t2[8];   // step-2 twiddle  W64^(n0*q0) = W512^(8*n0*q0)
};

AW_HD void fft_lane_const(int lane, const cf* tw512, FftLaneConst& c) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        c.t1[k] = tw512[(lane * k) & 511];
        c.t2[k] = tw512[(8 * (lane & 7) * k) & 511];
    }
}

template <int DIR> AW_HD cf tw_dir(cf w) { return DIR < 0 ? w : cconj(w); }

// ---- 512-point complex FFT in three lane-phases ---------------------------------
// phase A: radix-8 over r (stride 64), twiddle, scatter for transpose 1
template <int DIR> AW_HD void fft_phaseA(int lane, cf (&v)[8], const FftLaneConst& c, cf* s) {
    radix8<DIR>(v);
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) {
        cf w = (k0 == 0) ? v[0] : cmul(v[k0], tw_dir<DIR>(c.t1[k0]));
        s[k0 * 72 + lane] = w;
    }
}
// phase B: gather (k0 = lane>>3, n0 = lane&7, reg = n1), radix-8, twiddle, scatter 2
template <int DIR> AW_HD void fft_phaseB(int lane, cf (&v)[8], const FftLaneConst& c, cf* s) {
    const int k0 = lane >> 3, n0 = lane & 7;
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) v[n1] = s[k0 * 72 + 8 * n1 + n0];
    radix8<DIR>(v);
#pragma unroll
    for (int q0 = 0; q0 < 8; ++q0) {
        if (q0) v[q0] = cmul(v[q0], tw_dir<DIR>(c.t2[q0]));
    }
}
template <int DIR> AW_HD void fft_phaseB_store(int lane, const cf (&v)[8], cf* s) {
    const int k0 = lane >> 3, n0 = lane & 7;
#pragma unroll
    for (int q0 = 0; q0 < 8; ++q0) s[k0 * 73 + 8 * q0 + n0] = v[q0];
}
// phase C: gather (k0 = lane&7, q0 = lane>>3, reg = n0), radix-8 -> natural order
template <int DIR> AW_HD void fft_phaseC(int lane, cf (&v)[8], cf* s) {
    const int k0 = lane & 7, q0 = lane >> 3;
#pragma unroll
    for (int n0 = 0; n0 < 8; ++n0) v[n0] = s[k0 * 73 + 8 * q0 + n0];
    radix8<DIR>(v);
}

// ---- the same phases with the twiddles read from LDS tables instead of registers ---------
// (28 VGPRs less per lane; tw1s[k0*64 + lane] = W512^(lane*k0), tw2s[n0*8 + q0] = W64^(n0*q0);
// consecutive lanes read consecutive words / lanes with equal n0 broadcast: conflict-free)
template <int DIR> AW_HD void fft_phaseA_t(int lane, cf (&v)[8], const cf* tw1s, cf* s) {
    radix8<DIR>(v);
#pragma unroll
    for (int k0 = 0; k0 < 8; ++k0) {
        cf w = (k0 == 0) ? v[0] : cmul(v[k0], tw_dir<DIR>(tw1s[k0 * 64 + lane]));
        s[k0 * 72 + lane] = w;
    }
}
template <int DIR> AW_HD void fft_phaseB_t(int lane, cf (&v)[8], const cf* tw2s, cf* s) {
    const int k0 = lane >> 3, n0 = lane & 7;
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) v[n1] = s[k0 * 72 + 8 * n1 + n0];
    radix8<DIR>(v);
#pragma unroll
    for (int q0 = 1; q0 < 8; ++q0) v[q0] = cmul(v[q0], tw_dir<DIR>(tw2s[n0 * 8 + q0]));
}
// fill the two tables (all threads of the block; caller synchronises)
AW_HD void fft_fill_tables(int tid, int nthreads, const cf* tw512, cf* tw1s, cf* tw2s) {
    for (int i = tid; i < 512; i += nthreads) tw1s[i] = tw512[((i & 63) * (i >> 6)) & 511];
    for (int i = tid; i < 64; i += nthreads) tw2s[i] = tw512[(8 * (i >> 3) * (i & 7)) & 511];
}

// ---- real-FFT split (forward) -----------------------------------------------------
// After the forward complex FFT, lane L reg r holds Z[L+64r].  X[k], k = 0..511:
//   X[k] = E - i*W^k*D,  E = (Z[k] + conj Z[512-k])/2,  D = (Z[k] - conj Z[512-k])/2
// phase 1: every lane stores its Z in natural order; phase 2: reads the partners.
AW_HD void rfft_split_store(int lane, const cf (&v)[8], cf* s) {
#pragma unroll
    for (int r = 0; r < 8; ++r) s[lane + 64 * r] = v[r];
}
// returns X[lane + 64*r]; tw1024 = exp(-2 pi i k/1024)
AW_HD cf rfft_split_bin(int k, cf zk, const cf* s, const cf* tw1024) {
    cf zp = s[(512 - k) & 511];
    cf e = mk(0.5f * (zk.x + zp.x), 0.5f * (zk.y - zp.y));
    cf d = mk(0.5f * (zk.x - zp.x), 0.5f * (zk.y + zp.y));
    cf w = tw1024[k];
    cf wd = cmul(w, d);
    return mk(e.x + wd.y, e.y - wd.x);      // e - i*wd
}
// X[512] = Re Z[0] - Im Z[0]  (purely real)
AW_HD float rfft_split_nyquist(const cf* s) { return s[0].x - s[0].y; }

// ---- inverse real FFT merge ---------------------------------------------------------
// Given the one-sided spectrum X[0..512], build Z[k] = E + i*O with
//   E = (X[k] + conj X[512-k])/2,  O = (X[k] - conj X[512-k])/2 * W^-k
// Then z = IFFT512(Z) (unnormalised) / 512 gives x[2n] = Re z[n], x[2n+1] = Im z[n]
// with irfft's 1/1024 normalisation folded in: scale = 1/512 * (the 1/2 above).
AW_HD cf irfft_merge_bin(int k, cf xk, cf xp /* X[512-k] */, const cf* tw1024) {
    cf e = mk(0.5f * (xk.x + xp.x), 0.5f * (xk.y - xp.y));
    cf d = mk(0.5f * (xk.x - xp.x), 0.5f * (xk.y + xp.y));
    cf w = cconj(tw1024[k]);           // W^-k
    cf o = cmul(d, w);
    return mk(e.x - o.y, e.y + o.x);   // e + i*o
}

#ifdef __HIPCC__
__device__ __forceinline__ void wave_sync() {
    // LDS ops of one wave execute in order; this only stops the compiler from
    // moving LDS accesses across the phase boundary.
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
// Forward/inverse 512-point complex FFT of the wave's 8x64 values, natural order in
// and out.  `s` is this wave's private scratch (kFftScratch cf).
template <int DIR>
__device__ __forceinline__ void fft512_wave(int lane, cf (&v)[8], const FftLaneConst& c, cf* s) {
    fft_phaseA<DIR>(lane, v, c, s);
    wave_sync();
    fft_phaseB<DIR>(lane, v, c, s);
    wave_sync();
    fft_phaseB_store<DIR>(lane, v, s);
    wave_sync();
    fft_phaseC<DIR>(lane, v, s);
    wave_sync();
}
template <int DIR>
__device__ __forceinline__ void fft512_wave_t(int lane, cf (&v)[8], const cf* tw1s, const cf* tw2s, cf* s) {
    fft_phaseA_t<DIR>(lane, v, tw1s, s);
    wave_sync();
    fft_phaseB_t<DIR>(lane, v, tw2s, s);
    wave_sync();
    fft_phaseB_store<DIR>(lane, v, s);
    wave_sync();
    fft_phaseC<DIR>(lane, v, s);
    wave_sync();
}
#endif

}  // namespace aware
