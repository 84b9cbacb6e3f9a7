// Clip-aligned detector GEMM on the f16 matrix pipe with f32-level arithmetic: two-term operand split, three products.
//
// gfx950 has no fast path for f32 operands (v_mfma_f32_32x32x2_f32 runs at 1/16 of the 16-bit MFMA rate).  gemm_x3.hip
// writes an f32 value as three bf16 terms and runs six partial products.  This file uses the other 16-bit format:
//     a * 2^s = h + l + e,   h = RN_f16(a 2^s),  l = RN_f16(a 2^s - h),  |e| <= 2^-23 |a 2^s|
// (binary16 has 11 significand bits.  An f32 value has 24: h keeps the top 11, the residual a 2^s - h is exact in f32 -- a
// signed integer of at most 13 bits in units of the value's last place -- and l keeps 11 bits and the sign of it: exact for
// three quarters of all values, off by ONE unit in the last place of the f32 value for the rest; measured rms 2^-24.5).  The product a*b
// is h_a h_b + h_a l_b + l_a h_b, the term l_a l_b (<= 2^-22 |ab|, rms 2^-24.6) is dropped: THREE v_mfma_f32_16x16x32_f16
// with f32 accumulation per k-step, half the matrix-pipe time of the six-product kernel, 2/3 of its LDS fragment traffic and
// weight bytes.  Each partial product is exact in f32 (22 significand bits).  Error of one product against the exact a*b:
// rms 2^-23.7, worst case 2^-21 (an f32 multiply: rms 2^-25.2, worst 2^-24) -- i.e. this pipe is NOT bit-for-bit f32
// arithmetic; what it keeps is the error LEVEL of an f32 dot product, which the f32 accumulation of K terms sets in both
// cases (measured below).  The exact alternative is gemm_x3.hip (conv_pipe 2).
//
// binary16's exponent range (normal down to 2^-14) makes the scale 2^s part of the format:
//   * weights: one power of two per output channel (row of Wt), chosen when the weights are packed so that the row's
//     largest magnitude lands in [2^13, 2^14); the inverse goes into the epilogue (exact).  A weight more than 2^15 below
//     its row maximum has a subnormal l: its absolute error is then <= 2^-39 of the row maximum -- far below the f32
//     rounding of the row's large entries, which is what bounds a dot product's error;
//   * activations / gradients: one power of two per CLIP, from the clip's max |x|, which every producing kernel leaves as
//     per-wave partial maxima (K/16 floats per clip: no atomics, nothing to reset) and the consumer reduces at start-up.
// Measured against fp64 beside the f32-input MFMA kernel in tests/test_gpu_kernels.py::test_gemm_clip_x3 (mode 2) on operands
// spanning 2^10 per row: at or below the f32 chain's error on every shape and epilogue.
//
// Tiling, staging, epilogues: as gemm_x3.hip (one 512-thread workgroup = all rows of one clip x 128 columns, a wave = all
// rows x 16 columns, A split on the fly into LDS in fragment order, B fragments straight from L2, InstanceNorm statistics
// in registers).  Reference semantics of the epilogues: detection/modules/conv1d.py:38-42.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "common.hpp"
#include "split_bf16.hpp"

namespace aware {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------
// packing (device): Wt [N][K] f32 row-major (row pitch ldw) -> [N/16][KS][plane 0..1][lane 0..63][8 f16] + inverse scales
// ---------------------------------------------------------------------------------------------------
static inline size_t h2_plane_bytes(int N, int K) { return (size_t)N * (size_t)(((K + 31) / 32) * 32) * 2 * sizeof(uint16_t); }
size_t h2_packed_bytes(int N, int K) { return h2_plane_bytes(N, K) + (size_t)N * sizeof(float); }
const float* h2_inv_scale(const void* packed, int N, int K) { return (const float*)((const char*)packed + h2_plane_bytes(N, K)); }

// power-of-two scale that brings a maximum magnitude `amax` into [2^13, 2^14); 1 for zero / tiny maxima
__device__ __forceinline__ float h2_scale_for(float amax) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xFFu);
    return e < 20 ? 1.0f : __uint_as_float((unsigned)(267 - e) << 23);
}
__device__ __forceinline__ float h2_pow2_inverse(float s) {            // s is a power of two
    const unsigned e = (__float_as_uint(s) >> 23) & 0xFFu;
    return __uint_as_float((254u - e) << 23);
}

__global__ __launch_bounds__(64) void h2_row_scale_kernel(const float* __restrict__ Wt, int ldw, int K, float* __restrict__ binv) {
    const int n = blockIdx.x, lane = threadIdx.x;
    float m = 0.f;
    for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(Wt[(size_t)n * ldw + k]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) binv[n] = h2_pow2_inverse(h2_scale_for(m));
}

__global__ __launch_bounds__(64) void h2_pack_kernel(const float* __restrict__ Wt, int ldw, int N, int K, int KS,
                                                      const float* __restrict__ binv, u32x4* __restrict__ out) {
    const int ks = blockIdx.x, nt = blockIdx.y, lane = threadIdx.x;
    const int n = nt * 16 + (lane & 15), k0 = ks * 32 + 8 * (lane >> 4);
    const float s = h2_pow2_inverse(binv[n]);
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + 2 * j;
        const float x = k < K ? Wt[(size_t)n * ldw + k] * s : 0.f, y = k + 1 < K ? Wt[(size_t)n * ldw + k + 1] * s : 0.f;
        const f16x2 hh = {(_Float16)x, (_Float16)y};
        const f16x2 ll = {(_Float16)(x - (float)hh.x), (_Float16)(y - (float)hh.y)};
        h[j] = __builtin_bit_cast(unsigned, hh);
        l[j] = __builtin_bit_cast(unsigned, ll);
    }
    u32x4* o = out + ((size_t)(nt * KS + ks) * 2) * 64 + lane;
    o[0] = u32x4{h[0], h[1], h[2], h[3]};
    o[64] = u32x4{l[0], l[1], l[2], l[3]};
}

// Wt_dev: device [N][K] f32 (row pitch ldw), N % 16 == 0; out: device, h2_packed_bytes(N, K)
void launch_h2_pack(const float* Wt_dev, int ldw, int N, int K, void* out, hipStream_t st) {
    const int KS = (K + 31) / 32;
    float* binv = (float*)((char*)out + h2_plane_bytes(N, K));
    hipLaunchKernelGGL(h2_row_scale_kernel, dim3(N), dim3(64), 0, st, Wt_dev, ldw, K, binv);
    hipLaunchKernelGGL(h2_pack_kernel, dim3(KS, N / 16), dim3(64), 0, st, Wt_dev, ldw, N, K, KS, binv, (u32x4*)out);
}

// per-clip max |x| of a [clips * rows_per_clip][K] matrix into the partial layout the GEMM reads: entry 0 = the maximum,
// entries 1 .. K/16 - 1 = 0.  For operands whose producer does not leave the partials (tests, the small-batch mel kernels).
__global__ __launch_bounds__(256) void clip_amax_kernel(const float* __restrict__ A, int lda, int K, int rows_per_clip,
                                                         float* __restrict__ amax) {
    __shared__ float red[4];
    const int clip = blockIdx.x, tid = threadIdx.x;
    const int k4 = K >> 2;
    float m = 0.f;
    for (int i = tid; i < rows_per_clip * k4; i += 256) {
        const int r = i / k4, c = i % k4;
        const float4 v = *reinterpret_cast<const float4*>(A + (size_t)(clip * rows_per_clip + r) * lda + 4 * c);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    const int np = K >> 4;
    if (tid < np) amax[(size_t)clip * 64 + tid] = tid == 0 ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : 0.f;
}
void launch_clip_amax(const float* A, int lda, int K, int rows_per_clip, int B, float* amax, hipStream_t st) {
    hipLaunchKernelGGL(clip_amax_kernel, dim3(B), dim3(256), 0, st, A, lda, K, rows_per_clip, amax);
}

// ---------------------------------------------------------------------------------------------------
// the K loop of one tile (8 waves, a wave = all 32 RG rows x 16 columns)
// ---------------------------------------------------------------------------------------------------
// (x, y) scaled by s -> packed f16 pairs of the two planes (v_pk_mul_f32, v_cvt_pk_f16_f32, 2 v_cvt_f32_f16, v_pk_fma_f32,
// v_cvt_pk_f16_f32: the residual is one exact fused multiply-subtract)
__device__ __forceinline__ void h2_split_pair(float x, float y, float s, unsigned& h, unsigned& l) {
    const float tx = x * s, ty = y * s;
    const f16x2 hh = {(_Float16)tx, (_Float16)ty};
    const f16x2 ll = {(_Float16)(tx - (float)hh.x), (_Float16)(ty - (float)hh.y)};
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}

// acc[m] += (A[bm + 16 m .. +16)[0..K) * 2^sa) * (B 2^sb)^T for the wave's 16 columns bn + 16 wave ..; the caller unscales.
// `lds`: 2 * 2 * 2 * 2RG KiB of staging memory (two K tiles of 64, two K32 steps, two planes); every wave of the workgroup
// calls this with the same arguments; the caller provides a barrier between two calls that reuse `lds`.
// NW waves per workgroup, NTW 16-column tiles per wave: the workgroup's slab is 16 NW NTW columns wide and the A tile it stages
// is shared by all of them.
template <int RG, int NW, int NTW>
__device__ __forceinline__ void h2_tile_gemm(const float* __restrict__ A, int lda, const u32x4* __restrict__ Bpk, int K, int bm,
                                             int bn, unsigned char* lds, f32x4 (&acc)[2 * RG][NTW], int row_limit, float ascale) {
    constexpr int NT = 64 * NW;
    constexpr int MT = 2 * RG;            // 16-row tiles per clip
    constexpr int MH = RG;                // ... per half (the unit of the A-fragment schedule)
    constexpr int FRAG = 1024;            // one 16-row x 32-k f16 fragment image, bytes
    constexpr int PLANE = MT * FRAG;
    constexpr int KSS = 2 * PLANE;        // one K32 step
    constexpr int BUF = 2 * KSS;          // one K tile (BK = 64)
    // A staging: pass i covers rows RPP i .. RPP i + RPP - 1 of the K tile (RPP = threads / 16); thread -> (row tid >> 4, 4 floats
    // at k = 4 (tid & 15)): one fully coalesced 16-byte load per lane (16 lanes = one 256-byte row segment) and one 8-byte
    // LDS store per plane.  (1024 threads, 96 rows: the second pass has work for the first 8 waves only.)
    constexpr int RPP = NT / 16;
    constexpr int NPASS = (32 * RG + RPP - 1) / RPP;

    bm = __builtin_amdgcn_readfirstlane(bm);
    bn = __builtin_amdgcn_readfirstlane(bn);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    const int srow = tid >> 4, k4 = tid & 15, sc = k4 >> 1;      // sc: the 8-wide k chunk (one lane's share of a fragment)
    unsigned rb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) rb[t] = (unsigned)(t * KSS + kg * 256 + ((r16 ^ (4 * t + kg)) * 16));

    const int KS2 = K >> 5;
    const int nkt = K >> 6;
    const u32x4* bp = Bpk + ((size_t)((bn >> 4) + wave * NTW) * KS2) * 128;          // uniform; + lane per thread
    const float* ap = A + (size_t)bm * lda;                                          // uniform
    unsigned roff[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) roff[i] = (unsigned)(min(srow + RPP * i, row_limit - 1) * lda + k4 * 4);
    auto pass_ok = [&](int i) { return (32 * RG) % RPP == 0 || srow + RPP * i < 32 * RG; };      // wave-uniform
    // LDS slot of this thread's 4 k-values of row srow (+ 32 i: two fragment images further): the XOR of the row slot with
    // the k chunk keeps the 16 lanes of a row on 16 distinct 8-byte slots of a 128-byte bank row
    const unsigned woff = (unsigned)((sc >> 2) * KSS + (srow >> 4) * FRAG + (sc & 3) * 256 + (((srow & 15) ^ sc) * 16) + (k4 & 1) * 8);

#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

#ifndef H2_ABL
#define H2_ABL 0          // timing-only ablations of the K loop (tools/h2_ablate.sh); results are invalid when non-zero
#endif
    bool in_loop = false;
    float4 ra[NPASS];
    auto gload_c = [&](int i, int kt) {
        if ((H2_ABL & 32) && in_loop) return;
        if (pass_ok(i)) ra[i] = *reinterpret_cast<const float4*>(ap + kt * 64 + roff[i]);
    };
    auto split_store_c = [&](int i, unsigned boff) {
        if (!pass_ok(i)) return;
        uint2 qh, ql;
        if ((H2_ABL & 1) && in_loop) {
            qh = make_uint2(__float_as_uint(ra[i].x), __float_as_uint(ra[i].y));
            ql = make_uint2(__float_as_uint(ra[i].z), __float_as_uint(ra[i].w));
        } else {
            h2_split_pair(ra[i].x, ra[i].y, ascale, qh.x, ql.x);
            h2_split_pair(ra[i].z, ra[i].w, ascale, qh.y, ql.y);
        }
        unsigned char* d = lds + boff + woff + i * (RPP / 16) * FRAG;
        if ((H2_ABL & 2) && in_loop) {
            asm volatile("" :: "v"(qh.x), "v"(qh.y), "v"(ql.x), "v"(ql.y));
        } else {
            *reinterpret_cast<uint2*>(d) = qh;
            *reinterpret_cast<uint2*>(d + PLANE) = ql;
        }
    };
#ifndef H2_BDEPTH
#define H2_BDEPTH 2        // B fragment sets in flight: the fragments of K32 step s are requested H2_BDEPTH - 1 steps ahead
#endif
    constexpr int BD = H2_BDEPTH;
    u32x4 bq[BD][2][NTW];
    auto loadB = [&](int set, int ks2) {
        if ((H2_ABL & 8) && in_loop) return;
        ks2 = ks2 < KS2 ? ks2 : KS2 - 1;
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int p = 0; p < 2; ++p) bq[set][p][n] = (bp + (((size_t)n * KS2 + ks2) * 2 + p) * 64)[(unsigned)lane];
    };
    auto lds_barrier = [&]() {
        if ((H2_ABL & 16) && in_loop) return;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    // A fragments: the h plane is double-buffered (the next quarter's h fragments are requested a whole quarter -- 3 MH MFMAs
    // -- ahead), the l plane single-buffered and refilled right after its only product of the quarter (2 MH MFMAs ahead)
    f16x8 ah[2][MH], al[MH];
    auto read_h = [&](int set, unsigned off) {
        if ((H2_ABL & 4) && in_loop) return;
#pragma unroll
        for (int m = 0; m < MH; ++m) ah[set][m] = *reinterpret_cast<const f16x8*>(lds + off + m * FRAG);
    };
    auto read_l = [&](unsigned off) {
        if ((H2_ABL & 4) && in_loop) return;
#pragma unroll
        for (int m = 0; m < MH; ++m) al[m] = *reinterpret_cast<const f16x8*>(lds + off + PLANE + m * FRAG);
    };
#define H2_MFMA(a_, b_, hf_)                                                                                                   \
    _Pragma("unroll") for (int m = 0; m < MH; ++m)                                                                             \
        _Pragma("unroll") for (int n = 0; n < NTW; ++n)                                                                        \
            acc[(hf_) * MH + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16((a_)[m], __builtin_bit_cast(f16x8, (b_)[n]),       \
                                                                            acc[(hf_) * MH + m][n], 0, 0, 0)
#define H2_PIN(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

#pragma unroll
    for (int i = 0; i < NPASS; ++i) gload_c(i, 0);
#pragma unroll
    for (int d = 0; d < BD - 1; ++d) loadB(d, d);
#pragma unroll
    for (int i = 0; i < NPASS; ++i) split_store_c(i, 0);
#pragma unroll
    for (int i = 0; i < NPASS; ++i) gload_c(i, nkt > 1 ? 1 : 0);
    lds_barrier();
    read_h(0, rb[0]);
    read_l(rb[0]);
    if (H2_ABL) {            // (ablations: the second h set and the second B set are never refilled -- give them values)
        read_h(1, rb[0]);
#pragma unroll
        for (int n = 0; n < NTW; ++n) { bq[1][0][n] = bq[0][0][n]; bq[1][1][n] = bq[0][1][n]; }
    }
#ifndef H2_PRIO
#define H2_PRIO 1          // 1 = raised priority around the h_a MFMA cluster of a quarter (-1.3 % per iteration, alternating runs
                           // on one box); 3 = around the l_a cluster too (same); 2 = static priority for waves 4..7 (+0.8 %); 0 = none
#endif
#if H2_PRIO == 2
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
    in_loop = true;
    // (the loop body covers BD K tiles when BD = 3 so that the B set of a K32 step is a compile-time index)
    constexpr int KTU = (BD == 3) ? 3 : 1;
    for (int kt0 = 0; kt0 < nkt; kt0 += KTU) {
#pragma unroll
      for (int ku = 0; ku < KTU; ++ku) {
        const int kt = kt0 + ku;
        if (KTU > 1 && kt >= nkt) break;
        const unsigned cur = (kt & 1) * BUF, nxt = BUF - cur;
        const int ktn = kt + 2 < nkt ? kt + 2 : nkt - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                 // quarter = (K32 step q>>1, row half q&1)
            const int t = q >> 1, hf = q & 1;
            const int bs = (BD == 3) ? (2 * ku + t) % 3 : t;          // B set of this K32 step (step index mod BD)
            if (hf == 0) loadB((bs + BD - 1) % BD, kt * 2 + t + BD - 1);   // B fragments BD - 1 K32 steps ahead
            if (q < NPASS) {
                split_store_c(q, nxt);
                gload_c(q, ktn);
            }
            const unsigned noff = q < 3 ? cur + rb[(q + 1) >> 1] + ((q + 1) & 1) * MH * FRAG : nxt + rb[0];
            if (q < 3) { read_h((q + 1) & 1, noff); H2_PIN(0x100, MH); }
#if H2_PRIO == 3
            __builtin_amdgcn_s_setprio(1);
#endif
            H2_MFMA(al, bq[bs][0], hf);               // l_a * h_b
            H2_PIN(0x008, MH * NTW);
#if H2_PRIO == 3
            __builtin_amdgcn_s_setprio(0);
#endif
            if (q == 3) {                             // tile kt+1 is complete; every wave has finished its reads of tile kt
                lds_barrier();
                read_h(0, noff);
                H2_PIN(0x100, MH);
            }
            read_l(noff);
            H2_PIN(0x100, MH);
#if H2_PRIO == 1 || H2_PRIO == 3
            __builtin_amdgcn_s_setprio(1);
#endif
            H2_MFMA(ah[q & 1], bq[bs][1], hf);        // h_a * l_b
            H2_MFMA(ah[q & 1], bq[bs][0], hf);        // h_a * h_b
            H2_PIN(0x008, 2 * MH * NTW);
#if H2_PRIO == 1 || H2_PRIO == 3
            __builtin_amdgcn_s_setprio(0);
#endif
        }
      }
    }
#undef H2_PIN
#undef H2_MFMA
}

// ---------------------------------------------------------------------------------------------------
// the conv block / data-gradient kernel for uniform batches (epilogues as gemm_clip_x3_kernel)
// ---------------------------------------------------------------------------------------------------
// NW x NTW: 8 x 1 = 128-column slabs, two workgroups per CU; 16 x 1 (and 8 x 2, not instantiated: measured slower) = 256-column
// slabs (half the A staging per MFMA), one workgroup per CU
template <int RG, int EPI, int NW, int NTW>
__global__ __launch_bounds__(64 * NW, (NW == 16) ? 4 : (NTW == 1 && RG <= 3) ? 4 : 2) void gemm_clip_h2_kernel(
    const float* __restrict__ A, int lda, const u32x4* __restrict__ Bpk, const float* __restrict__ binv,
    const float* __restrict__ amax_in, float* __restrict__ amax_out, const float* __restrict__ bias, float* __restrict__ C, int ldc,
    int Tp, int N, int K, int tiles_n, int ntiles, float* __restrict__ rstd_io, const float* __restrict__ act,
    const u32x4* __restrict__ Lpk, float* __restrict__ zpart, int CL) {
    constexpr int MT = 2 * RG;
    constexpr int MH = RG;
    constexpr int FRAG = 1024;
    constexpr int BUF = 2 * 2 * MT * FRAG;
    constexpr int SLABW = 16 * NW * NTW;             // columns per workgroup
    constexpr int TPITCH = SLABW + 4;                // row pitch (floats) of the output tile re-laid in LDS (FWD_LAST)
    // (the FWD_LAST epilogue re-lays the output tile as f32 [32 RG][TPITCH] in the same memory, then parks NW x MH x 3
    //  partial tiles of 1 KiB there)
    constexpr int LASTB = 32 * RG * TPITCH * 4 > NW * MH * 3 * FRAG ? 32 * RG * TPITCH * 4 : NW * MH * 3 * FRAG;
    constexpr int LDSB = (EPI == X3_FWD_LAST && LASTB > 2 * BUF) ? LASTB : 2 * BUF;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDSB];

    // block -> (clip, column slab): blocks b and b + 8 share an XCD (observed round-robin placement; speed only).  An XCD takes
    // a contiguous range of clips and walks it slab-group-major, `sg` slabs at a time whose packed weights (sg * SLABW * K * 4
    // bytes) fit its L2 beside the activation rows in flight (gemm_x3.hip has the measurements).
    int id = blockIdx.x;
    int clip, slab_;
    if ((ntiles & 7) == 0) {
        const int x = id & 7, j = id >> 3, R = ntiles >> 3;
        const int nclip = R / tiles_n;
        if (nclip * tiles_n == R && nclip > 0) {
            int sg = (int)(3355443u / (unsigned)(SLABW * K * 4));
            sg = sg < 1 ? 1 : (sg > tiles_n ? tiles_n : sg);
            while (tiles_n % sg) --sg;
            const int per_group = nclip * sg;
            const int grp = j / per_group, r = j % per_group;
            clip = x * nclip + r / sg;
            slab_ = grp * sg + r % sg;
        } else {
            id = x * R + j;
            clip = id / tiles_n;
            slab_ = id % tiles_n;
        }
    } else {
        clip = id / tiles_n;
        slab_ = id % tiles_n;
    }
    const int bm = clip * 32 * RG;
    const int bn = slab_ * SLABW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;

    // the clip's scale from the producer's partial maxima (K/16 of them: one per wave of each of its workgroups)
    float am = lane < (K >> 4) ? amax_in[(size_t)clip * 64 + lane] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
    const float ascale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(h2_scale_for(am))));

    f32x4 acc[MT][NTW];
    h2_tile_gemm<RG, NW, NTW>(A, lda, Bpk, K, bm, bn, lds, acc, 32 * RG, ascale);

    // ---- epilogue: lane holds rows m*16 + 4*kg + e (e = 0..3) of columns cb + 16 n ----
    const int cb = bn + wave * (16 * NTW) + r16;
    const float ainv = h2_pow2_inverse(ascale);
    const float invT = 1.0f / (float)Tp;
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int col = cb + 16 * n;
        const float unscale = ainv * binv[col];
        float omax = 0.f;                               // max |output| of this wave's 16 columns, for the next GEMM's scale
        if (EPI == X3_PLAIN) {
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m * 16 + 4 * kg + e;
                    const float o = row < Tp ? acc[m][n][e] * unscale + bv : 0.f;
                    omax = fmaxf(omax, fabsf(o));
                    C[(size_t)(bm + row) * ldc + col] = o;
                }
        } else if (EPI == X3_FWD || EPI == X3_FWD_LAST) {
            const float bv = bias ? bias[col] : 0.f;
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m * 16 + 4 * kg + e;
                    acc[m][n][e] = acc[m][n][e] * unscale + bv;
                    if (row < Tp) s += acc[m][n][e];
                }
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const float mean = s * invT;
            float qq = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m * 16 + 4 * kg + e;
                    if (row < Tp) { const float d = acc[m][n][e] - mean; qq += d * d; }
                }
            qq += __shfl_xor(qq, 16);
            qq += __shfl_xor(qq, 32);
            const float rs = 1.0f / sqrtf(qq * invT + 1e-5f);      // biased variance, eps 1e-5 (InstanceNorm1d defaults)
            if (kg == 0) rstd_io[(size_t)clip * N + col] = rs;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m * 16 + 4 * kg + e;
                    const float u = (acc[m][n][e] - mean) * rs;
                    const float o = row < Tp ? (u > 0.f ? u : 0.2f * u) : 0.f;
                    acc[m][n][e] = o;
                    omax = fmaxf(omax, fabsf(o));
                    C[(size_t)(bm + row) * ldc + col] = o;
                }
        } else {
            // X3_BWD: acc = dL/dA of the previous block's output (read from `act`, post-activation);
            //         C = dL/dZ = rstd * (dU - mean_t dU - u * mean_t(dU*u)),  dU = acc * lrelu'(u)
            const float rs = rstd_io[(size_t)clip * N + col];
            float s1 = 0.f, s2 = 0.f;
            float u[MT][4];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m * 16 + 4 * kg + e;
                    // unconditional load (padding rows exist and hold zeros): a branch here would serialise the loads
                    const float av = act[(size_t)(bm + row) * ldc + col];
                    const bool valid = row < Tp;
                    const float uv = valid ? (av > 0.f ? av : av * 5.0f) : 0.f;                 // invert LeakyReLU(0.2)
                    const float du = valid ? acc[m][n][e] * unscale * (av > 0.f ? 1.f : 0.2f) : 0.f;
                    acc[m][n][e] = du;
                    u[m][e] = uv;
                    s1 += du;
                    s2 += du * uv;
                }
            s1 += __shfl_xor(s1, 16);
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 16);
            s2 += __shfl_xor(s2, 32);
            const float m1 = s1 * invT, m2 = s2 * invT;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m * 16 + 4 * kg + e;
                    const float o = row < Tp ? rs * (acc[m][n][e] - m1 - u[m][e] * m2) : 0.f;
                    omax = fmaxf(omax, fabsf(o));
                    C[(size_t)(bm + row) * ldc + col] = o;
                }
        }
        if (amax_out) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
            if (lane == 0) amax_out[(size_t)clip * 64 + (col >> 4)] = omax;
        }
    }
    if (EPI == X3_FWD_LAST) {
        // acc[m][n][e] holds this block's output (zero in padding rows).  The next conv block is the skinny last one (CL <= 48
        // channels): its K = this N is split over the workgroups' column slabs, so this workgroup contributes the partial
        // z_part[slab] = out[:, slab] * Wlast[:, slab]^T, computed on the bf16 pipe with the exact three-way split (operands
        // of gemm_x3.hip's pack: the tile is re-laid as A fragments (k = column) through LDS).  The slab has KT = SLABW / 32
        // K32 steps; work item (tq, mh) = (K32 step, half of the row tiles); wave w takes mh = w & 1 and the steps
        // tq = (w >> 1) + j NW/2, summing them in registers; the NW/2 partials per half then meet in LDS.
        constexpr int KT = SLABW / 32;
        const int KS2L = N >> 5, ncl = (CL + 15) >> 4;
        const int mh = wave & 1;
        __syncthreads();                                  // every wave is done with the staging buffers
        float* const T = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) T[(16 * m + 4 * kg + e) * TPITCH + (wave * NTW + n) * 16 + r16] = acc[m][n][e];
        __syncthreads();
        f32x4 zt[MH][3];
#pragma unroll
        for (int mm = 0; mm < MH; ++mm)
#pragma unroll
            for (int n = 0; n < 3; ++n) zt[mm][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < (2 * KT) / NW; ++j) {
            const int tq = (wave >> 1) + j * (NW / 2);
            bf16x8 bl[3][3];
#pragma unroll
            for (int n = 0; n < 3; ++n)
                if (n < ncl) {
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        bl[n][p] = __builtin_bit_cast(bf16x8, Lpk[(((size_t)n * KS2L + KT * slab_ + tq) * 3 + p) * 64 + lane]);
                }
#pragma unroll
            for (int mm = 0; mm < MH; ++mm) {
                const float* src = T + (16 * (mh * MH + mm) + r16) * TPITCH + 32 * tq + 8 * kg;
                const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
                uint4 q0, q1, q2;
                split_pair(x0.x, x0.y, q0.x, q1.x, q2.x);
                split_pair(x0.z, x0.w, q0.y, q1.y, q2.y);
                split_pair(x1.x, x1.y, q0.z, q1.z, q2.z);
                split_pair(x1.z, x1.w, q0.w, q1.w, q2.w);
                bf16x8 a[3];
                a[0] = __builtin_bit_cast(bf16x8, q0); a[1] = __builtin_bit_cast(bf16x8, q1); a[2] = __builtin_bit_cast(bf16x8, q2);
#pragma unroll
                for (int term = 0; term < 6; ++term) {
                    const int pa = term == 0 ? 2 : (term == 1 || term == 3) ? 1 : 0;
                    const int pb = term == 2 ? 2 : (term == 1 || term == 4) ? 1 : 0;
#pragma unroll
                    for (int n = 0; n < 3; ++n)
                        if (n < ncl) zt[mm][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[pa], bl[n][pb], zt[mm][n], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                  // all fragment reads done: the buffer becomes the partial store
#pragma unroll
        for (int mm = 0; mm < MH; ++mm)
#pragma unroll
            for (int n = 0; n < 3; ++n)
                *reinterpret_cast<f32x4*>(lds + (size_t)((wave * MH + mm) * 3 + n) * FRAG + lane * 16) = zt[mm][n];
        __syncthreads();
        if (wave < MT) {
            const int smh = wave / MH, smm = wave % MH;   // this wave finishes row tile `wave`
            float* zp = zpart + (size_t)slab_ * ((size_t)(ntiles / tiles_n) * 32 * RG * CL) + (size_t)(bm + 16 * wave + 4 * kg) * CL;
#pragma unroll
            for (int n = 0; n < 3; ++n)
                if (n < ncl) {
                    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < NW / 2; ++q)
                        t += *reinterpret_cast<const f32x4*>(lds + (size_t)(((2 * q + smh) * MH + smm) * 3 + n) * FRAG + lane * 16);
                    if (16 * n + r16 < CL) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) zp[(size_t)e * CL + 16 * n + r16] = t[e];
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Ragged batches: the same conv block / data-gradient GEMM for clips of ANY length in ONE launch (the structure of
// gemm_ragged_x3_kernel, gemm_x3.hip: a workgroup owns one clip x 128 columns and walks the clip's pooled rows in chunks of two
// or three 32-row groups; a clip of one chunk gets the single-pass fused epilogue, a longer one two passes with the
// InstanceNorm statistics carried in registers).  The clip's scale comes from amax_in as in the uniform kernel; the partial
// maxima of the output (per 16-column group, over all chunks) go to amax_out.
// ---------------------------------------------------------------------------------------------------
constexpr int kRaggedRGh = 3;           // largest chunk, in 32-row groups
extern __shared__ __attribute__((aligned(16))) unsigned char h2_dyn_lds[];

template <class T>
__device__ __forceinline__ T* h2_uniform_ptr(T* p) {       // (see uniform_ptr in gemm_x3.hip)
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(__attribute__((address_space(1))) T*)(((unsigned long long)hi << 32) | lo);
}

// one chunk: rows [bm, bm + 32 RG) of which `rows` are valid.  st0/st1/st2: forward (count, mean, M2) of the column; backward
// (unused, sum dU, sum dU*u) in-lane partial sums; omax: running max |stored value| of this lane's column (single-pass clips)
template <int RG, int EPI>
__device__ __attribute__((noinline)) void h2_ragged_chunk(const bool SINGLE, const float* __restrict__ A, int lda,
                                                          const u32x4* __restrict__ Bpk, const float* __restrict__ bias,
                                                          float* __restrict__ C, int ldc, int N, int K, int bm, int rows,
                                                          int store_rows, int bn, float* __restrict__ rstd_clip,
                                                          const float* __restrict__ act, float ascale, float unscale, float& st0,
                                                          float& st1, float& st2, float& omax) {
    unsigned char* lds = h2_dyn_lds;
    A = h2_uniform_ptr(A); Bpk = h2_uniform_ptr(Bpk); bias = h2_uniform_ptr(bias); C = h2_uniform_ptr(C);
    rstd_clip = h2_uniform_ptr(rstd_clip); act = h2_uniform_ptr(act);
    lda = __builtin_amdgcn_readfirstlane(lda); ldc = __builtin_amdgcn_readfirstlane(ldc);
    N = __builtin_amdgcn_readfirstlane(N); K = __builtin_amdgcn_readfirstlane(K);
    bm = __builtin_amdgcn_readfirstlane(bm); bn = __builtin_amdgcn_readfirstlane(bn);
    rows = __builtin_amdgcn_readfirstlane(rows); store_rows = __builtin_amdgcn_readfirstlane(store_rows);
    ascale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ascale)));
    constexpr int MT = 2 * RG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    f32x4 acc[MT][1];
    h2_tile_gemm<RG, 8, 1>(A, lda, Bpk, K, bm, bn, lds, acc, store_rows, ascale);
    const int col = bn + wave * 16 + r16;
    const float invR = 1.0f / (float)rows;
    if (EPI == X3_FWD) {
        const float bv = bias ? bias[col] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[m][0][e] = acc[m][0][e] * unscale + bv;
                if (m * 16 + 4 * kg + e < rows) s += acc[m][0][e];
            }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const float mean = s * invR;
        float qq = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (m * 16 + 4 * kg + e < rows) { const float d = acc[m][0][e] - mean; qq += d * d; }
        qq += __shfl_xor(qq, 16);
        qq += __shfl_xor(qq, 32);
        if (SINGLE) {
            const float rs = 1.0f / sqrtf(qq * invR + 1e-5f);      // biased variance, eps 1e-5 (InstanceNorm1d defaults)
            if (kg == 0) rstd_clip[col] = rs;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m * 16 + 4 * kg + e;
                    const float u = (acc[m][0][e] - mean) * rs;
                    const float o = row < rows ? (u > 0.f ? u : 0.2f * u) : 0.f;
                    omax = fmaxf(omax, fabsf(o));
                    if (row < store_rows) C[(size_t)(bm + row) * ldc + col] = o;
                }
        } else {
            // raw conv output now, statistics merged across the clip's chunks (Chan et al.)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m * 16 + 4 * kg + e;
                    if (row < store_rows) C[(size_t)(bm + row) * ldc + col] = row < rows ? acc[m][0][e] : 0.f;
                }
            const float nc = (float)rows, nt = st0 + nc, dl = mean - st1;
            st2 = st2 + qq + dl * dl * (st0 * nc / nt);
            st1 = st1 + dl * (nc / nt);
            st0 = nt;
        }
    } else {      // X3_BWD
        float s1 = 0.f, s2 = 0.f;
        float u[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m * 16 + 4 * kg + e;
                const float av = act[(size_t)(bm + min(row, store_rows - 1)) * ldc + col];   // unconditional (clamped, masked below)
                const bool valid = row < rows;
                const float uv = valid ? (av > 0.f ? av : av * 5.0f) : 0.f;                 // invert LeakyReLU(0.2)
                const float du = valid ? acc[m][0][e] * unscale * (av > 0.f ? 1.f : 0.2f) : 0.f;
                acc[m][0][e] = du;
                u[m][e] = uv;
                s1 += du;
                s2 += du * uv;
            }
        if (SINGLE) {
            const float rs = rstd_clip[col];
            s1 += __shfl_xor(s1, 16);
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 16);
            s2 += __shfl_xor(s2, 32);
            const float m1 = s1 * invR, m2 = s2 * invR;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m * 16 + 4 * kg + e;
                    const float o = row < rows ? rs * (acc[m][0][e] - m1 - u[m][e] * m2) : 0.f;
                    omax = fmaxf(omax, fabsf(o));
                    if (row < store_rows) C[(size_t)(bm + row) * ldc + col] = o;
                }
        } else {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m * 16 + 4 * kg + e;
                    if (row < store_rows) C[(size_t)(bm + row) * ldc + col] = acc[m][0][e];          // dU (zero in padding rows)
                }
            st1 += s1;
            st2 += s2;
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 4) void gemm_ragged_h2_kernel(const float* __restrict__ A, int lda, const u32x4* __restrict__ Bpk,
                                                                const float* __restrict__ binv, const float* __restrict__ amax_in,
                                                                float* __restrict__ amax_out, const float* __restrict__ bias,
                                                                float* __restrict__ C, int ldc, const int* __restrict__ frame_off,
                                                                const int* __restrict__ pool_off, const int* __restrict__ order, int N,
                                                                int K, int tiles_n, int ntiles, float* __restrict__ rstd_io,
                                                                const float* __restrict__ act) {
    // blocks b and b + 8 share an XCD (observed round-robin placement; speed only): the slabs of one clip stay on one XCD,
    // clips are dealt to the XCDs round-robin and dispatched longest first (gemm_ragged_x3_kernel has the measurements)
    int clip, slab;
    {
        const int id = blockIdx.x, nclips = ntiles / tiles_n;
        if ((nclips & 7) == 0) {
            const int j = id >> 3;
            clip = (j / tiles_n) * 8 + (id & 7);
            slab = j % tiles_n;
        } else {
            clip = id / tiles_n;
            slab = id % tiles_n;
        }
        if (order) clip = order[clip];
    }
    const int bn = slab * 128;
    const int Tp = (frame_off[clip + 1] - frame_off[clip]) / 2;
    const int row0 = pool_off[clip];
    if (Tp < 1) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int col = bn + wave * 16 + r16;
    float am = lane < (K >> 4) ? amax_in[(size_t)clip * 64 + lane] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
    const float ascale = h2_scale_for(am);
    const float unscale = h2_pow2_inverse(ascale) * binv[col];
    const int G = (Tp + 31) >> 5;                                   // 32-row groups of the clip
    const int nchunk = (G + kRaggedRGh - 1) / kRaggedRGh;
    const int gbase = G / nchunk, grem = G % nchunk;                // balanced: the first `grem` chunks take one group more
    float* rstd_clip = rstd_io + (size_t)clip * N;
    float st0 = 0.f, st1 = 0.f, st2 = 0.f, omax = 0.f;
    const bool single = nchunk == 1;
    int g0 = 0;
    for (int c = 0; c < nchunk; ++c) {
        const int ng = gbase + (c < grem ? 1 : 0);
        const int bm = row0 + 32 * g0;
        const int rows = min(32 * ng, Tp - 32 * g0);
        if (c) __syncthreads();                                     // every wave is done with the previous chunk's staging memory
        if (ng <= 2) h2_ragged_chunk<2, EPI>(single, A, lda, Bpk, bias, C, ldc, N, K, bm, rows, 32 * ng, bn, rstd_clip, act, ascale, unscale, st0, st1, st2, omax);
        else h2_ragged_chunk<3, EPI>(single, A, lda, Bpk, bias, C, ldc, N, K, bm, rows, 32 * ng, bn, rstd_clip, act, ascale, unscale, st0, st1, st2, omax);
        g0 += ng;
    }
    if (single) {
        if (amax_out) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
            if (lane == 0) amax_out[(size_t)clip * 64 + (col >> 4)] = omax;
        }
        return;
    }
    // ---- pass 2 over the raw tile this workgroup wrote: ROW-MAJOR (lane = 4 consecutive columns, half a wave = one 512-byte
    // row segment), the per-column statistics handed over through LDS ----
    const float invT = 1.0f / (float)Tp;
    const int npad = 32 * G;
    float* cstat = reinterpret_cast<float*>(h2_dyn_lds);           // [2][128]; the staging memory is free now
    __syncthreads();                                                // ... once every wave has left its last chunk
    if (EPI == X3_FWD) {
        const float rs = 1.0f / sqrtf(st2 * invT + 1e-5f);
        if (kg == 0) { rstd_clip[col] = rs; cstat[wave * 16 + r16] = st1; cstat[128 + wave * 16 + r16] = rs; }
    } else {
        float s1 = st1, s2 = st2;
        s1 += __shfl_xor(s1, 16);
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 16);
        s2 += __shfl_xor(s2, 32);
        if (kg == 0) { cstat[wave * 16 + r16] = s1 * invT; cstat[128 + wave * 16 + r16] = s2 * invT; }
    }
    __syncthreads();                                                // statistics in LDS; every wave's raw rows are visible
    const int c4 = (lane & 31) * 4, rr = 2 * wave + (lane >> 5);
    const float4 q0 = *reinterpret_cast<const float4*>(cstat + c4), q1 = *reinterpret_cast<const float4*>(cstat + 128 + c4);
    float* const Cw = C + (size_t)row0 * ldc + bn + c4;
    float pm = 0.f;                                                 // max |value| of this lane's 4 columns
    if (EPI == X3_FWD) {
        for (int r0 = rr; r0 < npad; r0 += 64) {                    // four rows per lane in flight
            float4 z[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = *reinterpret_cast<const float4*>(Cw + (size_t)min(r0 + 16 * j, npad - 1) * ldc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = r0 + 16 * j;
                if (r >= npad) continue;
                auto f = [&](float v, float mean, float rs) {
                    const float u = (v - mean) * rs;
                    const float o = (r < Tp) ? (u > 0.f ? u : 0.2f * u) : 0.f;
                    pm = fmaxf(pm, fabsf(o));
                    return o;
                };
                *reinterpret_cast<float4*>(Cw + (size_t)r * ldc) =
                    make_float4(f(z[j].x, q0.x, q1.x), f(z[j].y, q0.y, q1.y), f(z[j].z, q0.z, q1.z), f(z[j].w, q0.w, q1.w));
            }
        }
    } else {
        const float4 rs4 = *reinterpret_cast<const float4*>(rstd_clip + bn + c4);
        const float* const Aw = act + (size_t)row0 * ldc + bn + c4;
        for (int r0 = rr; r0 < npad; r0 += 64) {
            float4 du[4], av[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t o = (size_t)min(r0 + 16 * j, npad - 1) * ldc;
                du[j] = *reinterpret_cast<const float4*>(Cw + o);
                av[j] = *reinterpret_cast<const float4*>(Aw + o);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = r0 + 16 * j;
                if (r >= npad) continue;
                auto f = [&](float d, float a, float rs, float m1, float m2) {
                    const float uv = a > 0.f ? a : a * 5.0f;
                    const float o = (r < Tp) ? rs * (d - m1 - uv * m2) : 0.f;
                    pm = fmaxf(pm, fabsf(o));
                    return o;
                };
                *reinterpret_cast<float4*>(Cw + (size_t)r * ldc) =
                    make_float4(f(du[j].x, av[j].x, rs4.x, q0.x, q1.x), f(du[j].y, av[j].y, rs4.y, q0.y, q1.y),
                                f(du[j].z, av[j].z, rs4.z, q0.z, q1.z), f(du[j].w, av[j].w, rs4.w, q0.w, q1.w));
            }
        }
    }
    if (amax_out) {
        // the 16-column group of a lane is (lane & 31) >> 2: its four lanes in both half-waves, then the eight waves through LDS
        pm = fmaxf(pm, __shfl_xor(pm, 1));
        pm = fmaxf(pm, __shfl_xor(pm, 2));
        pm = fmaxf(pm, __shfl_xor(pm, 32));
        float* gm = cstat + 256;                                    // [8 waves][8 groups]
        if ((lane & 35) == 0) gm[wave * 8 + ((lane & 31) >> 2)] = pm;
        __syncthreads();
        if (threadIdx.x < 8) {
            float m = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) m = fmaxf(m, gm[w * 8 + threadIdx.x]);
            amax_out[(size_t)clip * 64 + (bn >> 4) + threadIdx.x] = m;
        }
    }
}

// epi: 1 forward, 2 backward (as launch_gemm_ragged_x3); amax_in / amax_out as launch_gemm_clip_h2
void launch_gemm_ragged_h2(const float* A, int lda, const void* Bpk, const float* amax_in, float* amax_out, const float* bias,
                           float* C, int ldc, int B, const int* frame_off, const int* pool_off, const int* order, int N, int K,
                           int epi, float* rstd_io, const float* act, hipStream_t st) {
    const int tn = N / 128;
    const float* binv = h2_inv_scale(Bpk, N, K);
    constexpr size_t kLds = 2 * 2 * 2 * (2 * kRaggedRGh) * 1024;      // two K tiles of the tallest chunk
    if (epi == X3_FWD)
        hipLaunchKernelGGL((gemm_ragged_h2_kernel<X3_FWD>), dim3(tn * B), dim3(512), kLds, st, A, lda, (const u32x4*)Bpk, binv, amax_in,
                           amax_out, bias, C, ldc, frame_off, pool_off, order, N, K, tn, tn * B, rstd_io, act);
    else
        hipLaunchKernelGGL((gemm_ragged_h2_kernel<X3_BWD>), dim3(tn * B), dim3(512), kLds, st, A, lda, (const u32x4*)Bpk, binv, amax_in,
                           amax_out, bias, C, ldc, frame_off, pool_off, order, N, K, tn, tn * B, rstd_io, act);
}

// per-clip max |x| over the clip's pooled rows (ragged layout), into the partial layout: entry 0 = the maximum, 1 .. K/16-1 = 0
__global__ __launch_bounds__(256) void ragged_amax_kernel(const float* __restrict__ A, int lda, int K, const int* __restrict__ frame_off,
                                                           const int* __restrict__ pool_off, float* __restrict__ amax) {
    __shared__ float red[4];
    const int clip = blockIdx.x, tid = threadIdx.x;
    const int Tp = (frame_off[clip + 1] - frame_off[clip]) / 2, row0 = pool_off[clip];
    const int k4 = K >> 2;
    float m = 0.f;
    for (int i = tid; i < Tp * k4; i += 256) {
        const int r = i / k4, c = i % k4;
        const float4 v = *reinterpret_cast<const float4*>(A + (size_t)(row0 + r) * lda + 4 * c);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    const int np = K >> 4;
    if (tid < np) amax[(size_t)clip * 64 + tid] = tid == 0 ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : 0.f;
}
void launch_ragged_amax(const float* A, int lda, int K, const int* frame_off, const int* pool_off, int B, float* amax, hipStream_t st) {
    hipLaunchKernelGGL(ragged_amax_kernel, dim3(B), dim3(256), 0, st, A, lda, K, frame_off, pool_off, amax);
}

bool gemm_clip_h2_supported(int nwm, int N, int K, int lda) {
    return nwm >= 1 && nwm <= 4 && N % 128 == 0 && K % 64 == 0 && K <= 1024 && lda % 4 == 0;
}

// Wave arrangement: 8 waves x 1 tile (128-column slabs, two workgroups per CU).  Measured alternatives on the five launches of
// an iteration at B = 256 (kernel trace of the launches alone, us): 8 x 1 543; 16 x 1 (256-column slabs, one 1024-thread
// workgroup per CU: half the A staging per MFMA) 510 -- but no difference inside the embed loop (1.025 vs 1.026 ms per iteration,
// alternating runs on one box); 8 x 2 (two tiles per wave, 150 VGPRs) 548; the same GEMM on v_mfma_f32_32x32x16_f16 (a wave =
// all rows x 32 columns: half the MFMA issues, LDS fragment reads and staging per multiply-add, at the 128-VGPR limit) 565;
// B fragments two K32 steps ahead instead of one: no change; the split arithmetic of a quarter spread behind its MFMAs (two
// vector instructions per MFMA, sched_group_barrier) instead of ahead of them: +3 % (the scheduler then exposes the fragment
// reads).  DESIGN.md section 4 has the counters behind this.
int gemm_clip_h2_slab_width(int, int, int) { return 128; }

// Bpk: launch_h2_pack image of Wt [N][K]; amax_in: [B][64] partial maxima of A's clips (K/16 valid per clip); amax_out: the
// same for C ([B][64], N/16 written per clip) or null; lastpk / zpart / CL as launch_gemm_clip_x3 (gemm_x3.hip's pack), with
// N / gemm_clip_h2_slab_width(nwm, N, B) partial slabs
void launch_gemm_clip_h2(const float* A, int lda, const void* Bpk, const float* amax_in, float* amax_out, const float* bias,
                         float* C, int ldc, int B, int nwm, int Tp, int N, int K, int epi, float* rstd_io, const float* act,
                         hipStream_t st, const void* lastpk, float* zpart, int CL) {
    const float* binv = h2_inv_scale(Bpk, N, K);
    if (epi == X3_FWD && lastpk && zpart) epi = X3_FWD_LAST;
    const int tn = N / 128;
#define HK(M_, E_, W_) hipLaunchKernelGGL((gemm_clip_h2_kernel<M_, E_, W_, 1>), dim3(tn * B), dim3(64 * W_), 0, st, A, lda,         \
                                          (const u32x4*)Bpk, binv, amax_in, amax_out, bias, C, ldc, Tp, N, K, tn, tn * B, rstd_io,  \
                                          act, (const u32x4*)lastpk, zpart, CL)
#define HE(M_, W_) if (epi == X3_FWD) { HK(M_, X3_FWD, W_); } else if (epi == X3_BWD) { HK(M_, X3_BWD, W_); }                        \
                   else if (epi == X3_FWD_LAST) { HK(M_, X3_FWD_LAST, W_); } else { HK(M_, X3_PLAIN, W_); }
    switch (nwm) { case 1: HE(1, 8) break; case 2: HE(2, 8) break; case 3: HE(3, 8) break; default: HE(4, 8) break; }
#undef HE
#undef HK
}

}  // namespace aware
