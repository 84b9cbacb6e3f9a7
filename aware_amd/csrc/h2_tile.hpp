// Shared by gemm_h2.hip and gemm_h2p.hip: the f16 two-term operand split (see gemm_h2.hip's header for the arithmetic) and the
// K loop of one clip x 128-column tile with the f32 A operand split on the fly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "split_bf16.hpp"

namespace aware {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// power-of-two scale that brings a maximum magnitude `amax` into [2^13, 2^14); 1 for zero / tiny maxima
__device__ __forceinline__ float h2_scale_for(float amax) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xFFu);
    return e < 20 ? 1.0f : __uint_as_float((unsigned)(267 - e) << 23);
}
__device__ __forceinline__ float h2_pow2_inverse(float s) {            // s is a power of two
    const unsigned e = (__float_as_uint(s) >> 23) & 0xFFu;
    return __uint_as_float((254u - e) << 23);
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
// PERM: the k order inside a K32 step is the one of gemm_h2p.hip's fragment images (lane chunk kg holds k = 4 kg .. 4 kg + 3 and
// 16 + 4 kg .. 16 + 4 kg + 3; Bpk packed with launch_h2_pack(..., perm = true)).  SWAP: the MFMA takes the weights as its first
// operand: acc[m] then holds rows m*16 + (lane & 15) of columns bn + 16 wave + 4 (lane >> 4) + e  (e = 0..3) -- four consecutive
// output channels of one row per lane, which is the A-fragment layout of the NEXT GEMM under PERM.
template <int RG, int NW, int NTW, bool PERM = false, bool SWAP = false>
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
    const int srow = tid >> 4, k4 = tid & 15;
    // sc: the lane chunk (4 t + kg) this thread's four k values belong to; shalf: which 8-byte half of the chunk
    const int sc = PERM ? ((k4 >> 3) * 4 + (k4 & 3)) : (k4 >> 1), shalf = PERM ? ((k4 >> 2) & 1) : (k4 & 1);
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
    const unsigned woff = (unsigned)((sc >> 2) * KSS + (srow >> 4) * FRAG + (sc & 3) * 256 + (((srow & 15) ^ sc) * 16) + shalf * 8);

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
            acc[(hf_) * MH + m][n] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, (b_)[n]), (a_)[m],  \
                                                                                   acc[(hf_) * MH + m][n], 0, 0, 0)             \
                                          : __builtin_amdgcn_mfma_f32_16x16x32_f16((a_)[m], __builtin_bit_cast(f16x8, (b_)[n]),  \
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

}  // namespace aware
