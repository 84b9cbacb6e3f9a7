// Clip-aligned detector GEMM on the bf16 matrix pipe with fp32-equivalent arithmetic (gfx950 only).
//
// gfx950 has no fast path for f32 operands: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate.
// An f32 value is EXACTLY the sum of three bf16 values (8 + 8 + 8 significand bits, round-to-nearest
// residuals): a = a0 + a1 + a2.  The product a*b is then the sum of nine bf16*bf16 partial products,
// each exact in f32; the three with i + j >= 3 are below 2^-26 |a*b| (less than one f32 rounding) and
// are dropped.  The remaining six run as six v_mfma_f32_32x32x16_bf16 with f32 accumulation:
// 6/16 of the f32-MFMA time for the same k, error vs fp64 at the level of the f32 MFMA chain
// (measured in tests/test_gpu_kernels.py::test_gemm_clip_x3).
//
// Layout decisions:
//  * the weights are constant for the 400 iterations -> split and re-ordered ONCE on the host into MFMA
//    fragment order  [N/32][ceil(K/16)][plane 0..2][lane 0..63][8 bf16]; a wave loads its B fragments
//    straight from global memory into VGPRs as fully coalesced 1 KiB reads (no LDS for B: no other wave
//    of the workgroup needs the same columns);
//  * the activations are split on the fly while they are staged to LDS; the LDS image is the fragment
//    order too (ds_read_b128, lane-linear up to an XOR of the row slot that keeps the 16-byte stores of
//    a row's eight k-chunks on distinct banks);
//  * one workgroup = 4 waves = all rows of ONE clip (RG groups of 32 pooled frames) x 128 columns; a wave
//    owns every row of its 32 columns, so the InstanceNorm statistics of the fused epilogues are
//    in-register sums plus one cross-half shuffle (no LDS reduction, no extra barrier).
// Reference semantics of the epilogues: detection/modules/conv1d.py:38-42 (conv -> InstanceNorm1d -> LeakyReLU).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "kernels.h"

#ifndef X3_ABLATE
#define X3_ABLATE 0
#endif

namespace aware {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------
// host side: three-way bf16 split (round to nearest even on the f32 bits) and fragment-order packing
// ---------------------------------------------------------------------------------------------------
static inline uint16_t bf16_rne(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float v;
    memcpy(&v, &u, 4);
    return v;
}

size_t x3_packed_bytes(int N, int K) { return (size_t)(N / 32) * ((K + 15) / 16) * 3 * 64 * 8 * sizeof(uint16_t); }

// Wt: [N][K] row-major f32 (the NT operand: C = A * Wt^T), N % 32 == 0
void x3_pack(const float* Wt, int N, int K, uint16_t* out) {
    const int KS = (K + 15) / 16;
    for (int nt = 0; nt < N / 32; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int n = nt * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5) + j;
                    const float v = k < K ? Wt[(size_t)n * K + k] : 0.f;
                    const uint16_t p0 = bf16_rne(v);
                    const float r1 = v - bf16_to_f32(p0);
                    const uint16_t p1 = bf16_rne(r1);
                    const float r2 = r1 - bf16_to_f32(p1);
                    const uint16_t p2 = bf16_rne(r2);
                    const size_t base = ((size_t)(nt * KS + ks) * 3) * 512 + (size_t)lane * 8 + j;
                    out[base] = p0;
                    out[base + 512] = p1;
                    out[base + 1024] = p2;
                }
}

// ---------------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// (x, y) -> packed bf16 pairs of the three planes; residuals are exact f32 subtractions
__device__ __forceinline__ void split_pair(float x, float y, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = cvt_pk_bf16(x, y);
    const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xFFFF0000u);
    p1 = cvt_pk_bf16(rx, ry);
    const float sx = rx - __uint_as_float(p1 << 16), sy = ry - __uint_as_float(p1 & 0xFFFF0000u);
    p2 = cvt_pk_bf16(sx, sy);
}

enum { X3_PLAIN = 0, X3_FWD = 1, X3_BWD = 2 };

template <int RG, int EPI>
__global__ __launch_bounds__(256) void gemm_clip_x3_kernel(const float* __restrict__ A, int lda,
                                                            const u32x4* __restrict__ Bpk, const float* __restrict__ bias,
                                                            float* __restrict__ C, int ldc, int Tp, int N, int K, int tiles_n,
                                                            int ntiles, float* __restrict__ rstd_io,
                                                            const float* __restrict__ act) {
    constexpr int FRAG = 1024;            // one 32-row x 16-k bf16 fragment image, bytes
    constexpr int PLANE = RG * FRAG;      // the RG row groups of one (k16 step, plane)
    constexpr int KSS = 3 * PLANE;        // one k16 step
    constexpr int BUF = 4 * KSS;          // one K tile (BK = 64): RG = 3 -> 36 KiB
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];

    int id = blockIdx.x;
    if ((ntiles & 7) == 0) id = (id & 7) * (ntiles >> 3) + (id >> 3);     // contiguous run of tiles per XCD
    const int clip = id / tiles_n;
    const int bm = clip * 32 * RG;
    const int bn = (id % tiles_n) * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // staging role: row `srow` of every row group, k-chunk `sc` (8 floats) of the 64-wide K tile
    const int srow = tid >> 3, sc = tid & 7;
    const unsigned swr = (unsigned)((sc >> 1) * KSS + (sc & 1) * 512 + ((srow ^ sc) * 16));
    unsigned rbase[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) rbase[s] = (unsigned)(s * KSS + h * 512 + ((r ^ (2 * s + h)) * 16));

    const int KS = K >> 4;                // K % 64 == 0 (checked by the launcher)
    const int nkt = K >> 6;
    const u32x4* bp = Bpk + ((size_t)((bn >> 5) + wave) * KS) * 192 + lane;
    const float* ap = A + (size_t)(bm + srow) * lda + sc * 8;

    f32x16 acc[RG];
#pragma unroll
    for (int m = 0; m < RG; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

    float4 ra[RG][2];
    auto gload_g = [&](int i, int kt) {                 // row group i of K tile kt
        const float* p = ap + (size_t)(i * 32) * lda + kt * 64;
        ra[i][0] = *reinterpret_cast<const float4*>(p);
        ra[i][1] = *reinterpret_cast<const float4*>(p + 4);
    };
    auto split_store_g = [&](int i, unsigned boff) {
        uint4 q0, q1, q2;
        split_pair(ra[i][0].x, ra[i][0].y, q0.x, q1.x, q2.x);
        split_pair(ra[i][0].z, ra[i][0].w, q0.y, q1.y, q2.y);
        split_pair(ra[i][1].x, ra[i][1].y, q0.z, q1.z, q2.z);
        split_pair(ra[i][1].z, ra[i][1].w, q0.w, q1.w, q2.w);
        unsigned char* d = lds + boff + swr + i * FRAG;
        *reinterpret_cast<uint4*>(d) = q0;
        *reinterpret_cast<uint4*>(d + PLANE) = q1;
        *reinterpret_cast<uint4*>(d + 2 * PLANE) = q2;
    };
    u32x4 bq[4][3];
    auto loadB = [&](int set, int ksg) {
        ksg = ksg < KS ? ksg : KS - 1;                  // the tail prefetches re-read the last step (never used)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[set][p] = bp[(size_t)(ksg * 3 + p) * 64];
    };
    // LDS-only barrier: the pending global loads (next A tile, B fragments) stay in flight across it
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

#pragma unroll
    for (int i = 0; i < RG; ++i) gload_g(i, 0);
    loadB(0, 0);
    loadB(1, 1);
#pragma unroll
    for (int i = 0; i < RG; ++i) split_store_g(i, 0);
#pragma unroll
    for (int i = 0; i < RG; ++i) gload_g(i, nkt > 1 ? 1 : 0);
    lds_barrier();
    // A fragments are read one k16 sub-step ahead of the MFMAs that use them (two register sets)
    bf16x8 af[2][3][RG];
    auto read_frags = [&](int set, unsigned off) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int m = 0; m < RG; ++m)
                af[set][p][m] = *reinterpret_cast<const bf16x8*>(lds + off + p * PLANE + m * FRAG);
    };
    read_frags(0, rbase[0]);
    for (int kt = 0; kt < nkt; ++kt) {
        const unsigned cur = (kt & 1) * BUF, nxt = BUF - cur;
        const int ktn = kt + 2 < nkt ? kt + 2 : nkt - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#if !(X3_ABLATE & 1)
            loadB((s + 2) & 3, kt * 4 + s + 2);
#endif
            // this sub-step's share of staging K tile kt+1 (split into the other buffer) and of loading tile kt+2
#if !(X3_ABLATE & 2)
            if (s < RG) {
                split_store_g(s, nxt);
                gload_g(s, ktn);
            }
#endif
            if (s < 3) {
                read_frags((s + 1) & 1, cur + rbase[s + 1]);
            } else {
#if !(X3_ABLATE & 4)
                lds_barrier();               // tile kt+1 is complete; every wave has read all of tile kt
#endif
                read_frags(0, nxt + rbase[0]);
            }
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, bq[s][0]);
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, bq[s][1]);
            const bf16x8 b2 = __builtin_bit_cast(bf16x8, bq[s][2]);
            const int cs = s & 1;
            // smallest partial products first
#pragma unroll
            for (int m = 0; m < RG; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cs][2][m], b0, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < RG; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cs][1][m], b1, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < RG; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cs][0][m], b2, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < RG; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cs][1][m], b0, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < RG; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cs][0][m], b1, acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < RG; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cs][0][m], b0, acc[m], 0, 0, 0);
        }
    }

    // ---- epilogue: this wave holds rows m*32 + (e&3) + 8*(e>>2) + 4*h of column `col`, all rows of the clip ----
    const int col = bn + wave * 32 + r;
    const float invT = 1.0f / (float)Tp;
    if (EPI == X3_PLAIN) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int m = 0; m < RG; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                C[(size_t)(bm + row) * ldc + col] = row < Tp ? acc[m][e] + bv : 0.f;
            }
        return;
    }
    if (EPI == X3_FWD) {
        const float bv = bias ? bias[col] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < RG; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                acc[m][e] += bv;
                if (row < Tp) s += acc[m][e];
            }
        s += __shfl_xor(s, 32);
        const float mean = s * invT;
        float q = 0.f;
#pragma unroll
        for (int m = 0; m < RG; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < Tp) { const float d = acc[m][e] - mean; q += d * d; }
            }
        q += __shfl_xor(q, 32);
        const float rs = 1.0f / sqrtf(q * invT + 1e-5f);      // biased variance, eps 1e-5 (InstanceNorm1d defaults)
        if (h == 0) rstd_io[(size_t)clip * N + col] = rs;
#pragma unroll
        for (int m = 0; m < RG; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const float u = (acc[m][e] - mean) * rs;
                C[(size_t)(bm + row) * ldc + col] = row < Tp ? (u > 0.f ? u : 0.2f * u) : 0.f;
            }
        return;
    }
    // X3_BWD: acc = dL/dA of the previous block's output (read from `act`, post-activation);
    //         C = dL/dZ = rstd * (dU - mean_t dU - u * mean_t(dU*u)),  dU = acc * lrelu'(u)
    {
        const float rs = rstd_io[(size_t)clip * N + col];
        float s1 = 0.f, s2 = 0.f;
        float u[RG][16];
#pragma unroll
        for (int m = 0; m < RG; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                float du = 0.f, uv = 0.f;
                if (row < Tp) {
                    const float av = act[(size_t)(bm + row) * ldc + col];
                    uv = av > 0.f ? av : av * 5.0f;                 // invert LeakyReLU(0.2)
                    du = acc[m][e] * (av > 0.f ? 1.f : 0.2f);
                }
                acc[m][e] = du;
                u[m][e] = uv;
                s1 += du;
                s2 += du * uv;
            }
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        const float m1 = s1 * invT, m2 = s2 * invT;
#pragma unroll
        for (int m = 0; m < RG; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                C[(size_t)(bm + row) * ldc + col] = row < Tp ? rs * (acc[m][e] - m1 - u[m][e] * m2) : 0.f;
            }
    }
}

bool gemm_clip_x3_supported(int nwm, int N, int K, int lda) {
    return nwm >= 1 && nwm <= 4 && N % 128 == 0 && K % 64 == 0 && lda % 4 == 0;
}

void launch_gemm_clip_x3(const float* A, int lda, const void* Bpk, const float* bias, float* C, int ldc, int B, int nwm,
                         int Tp, int N, int K, int epi, float* rstd_io, const float* act, hipStream_t st) {
    const int tn = N / 128;
#define XK(M_, E_) hipLaunchKernelGGL((gemm_clip_x3_kernel<M_, E_>), dim3(tn * B), dim3(256), 0, st, A, lda,          \
                                      (const u32x4*)Bpk, bias, C, ldc, Tp, N, K, tn, tn * B, rstd_io, act)
#define XM(E_) switch (nwm) { case 1: XK(1, E_); break; case 2: XK(2, E_); break; case 3: XK(3, E_); break; default: XK(4, E_); break; }
    if (epi == X3_FWD) { XM(X3_FWD) } else if (epi == X3_BWD) { XM(X3_BWD) } else { XM(X3_PLAIN) }
#undef XM
#undef XK
}

}  // namespace aware
