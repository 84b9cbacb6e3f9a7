// The conv-block chain of a uniform batch on the f16 two-term pipe with the activations kept PRE-SPLIT between the layers.
//
// gemm_h2.hip reads its A operand as f32 rows and every one of the N/128 workgroups of a clip splits the same elements into
// binary16 (h, l) pairs again: at N = 1024 each element is split eight times, 45 vector instructions and three LDS stores
// per wave and K tile beside 36 MFMAs (profiles/r03_gemm_h2_ablation.txt: the split, the stores and the loads behind them
// are a third of the kernel).  Here the PRODUCER of an operand writes it once as "planes": per clip the A-fragment images
// of the consuming GEMM,
//     [K/32 k-steps][2 planes: h, l][MT row tiles of 16][64 lanes][8 binary16]           (K * MT * 64 bytes per clip =
//                                                                                         the bytes of the f32 rows it replaces)
// scaled by one power of two per (clip, 128-column slab) -- the slab is one producing workgroup, which knows its own maximum;
// scales [B][8] travel beside the image.  The consumer copies a K tile (64 k: 4 MT KiB, contiguous) into LDS with LDS-DMA
// loads (global_load_lds_dwordx4: no registers, no split, no ds_write), reads fragments with ds_read_b128 as before, and
// multiplies its accumulators by the ratio of consecutive slab scales where the slab changes (exact: powers of two).
//
// What makes the producer side free is the operand ORDER of the MFMA.  With the weights as the first operand the result tile
// has one activation ROW per lane and four consecutive output CHANNELS in the lane's registers -- and "four consecutive k of
// one row per lane" is exactly half an A-fragment chunk.  So the k order inside a K32 step is defined as
//     lane chunk kg, element j:  k = 4 kg + j (j < 4),  16 + 4 kg + (j - 4)  (j >= 4)
// (any order works as long as weights and activations agree: launch_h2_pack(..., perm = true)), and a wave of the producing
// GEMM writes its 16 channels x 96 rows straight from the accumulators as 8-byte halves of the lane chunks: no transposition,
// no LDS.  InstanceNorm statistics reduce over rows = over the 16 lanes of a DPP row (four DPP steps).
//
// Chain of one embed iteration (capi.hip, det_forward_backward):
//   x0 f32 --conv0 [f32 A, split on the fly]--> P(out0) --conv1--> P(out1) --conv2 [FWD_LAST, classic operand order: f32 out2
//   + split-K partials of the last conv]--> read-out --> P(dZ2) --bwd2 [act = P(out1)]--> P(dZ1) --bwd1 [act = P(out0)]--> dZ0 f32
// Arithmetic, error model and the per-channel weight scales: gemm_h2.hip.  Reference semantics: detection/modules/conv1d.py:38-42.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "common.hpp"
#include "split_bf16.hpp"
#include "h2_tile.hpp"

namespace aware {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

static inline size_t h2p_clip_bytes(int rg, int K) { return (size_t)K * (2 * rg) * 64; }
size_t h2p_planes_bytes(int B, int nwm, int K) { return (size_t)B * h2p_clip_bytes(nwm, K); }

__device__ __forceinline__ float readlane_f(float x, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}

// ---- DPP reductions over the 16 lanes of a row (lanes 16 g .. 16 g + 15): every lane ends with the row's result ----
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float x) {
    x += dpp_f<0xB1>(x);      // quad_perm [1,0,3,2]
    x += dpp_f<0x4E>(x);      // quad_perm [2,3,0,1]
    x += dpp_f<0x141>(x);     // row_half_mirror
    x += dpp_f<0x140>(x);     // row_mirror
    return x;
}
__device__ __forceinline__ float row16_max(float x) {
    x = fmaxf(x, dpp_f<0xB1>(x));
    x = fmaxf(x, dpp_f<0x4E>(x));
    x = fmaxf(x, dpp_f<0x141>(x));
    x = fmaxf(x, dpp_f<0x140>(x));
    return x;
}

// ---------------------------------------------------------------------------------------------------
// f32 rows <-> planes (tests, and operands that have no producing kernel of this file)
// ---------------------------------------------------------------------------------------------------
// one workgroup per (clip, 128-column slab): slab maximum -> scale -> the slab's 4 K32 steps x MT row tiles x 2 planes
template <int RG>
__global__ __launch_bounds__(512) void h2p_from_f32_kernel(const float* __restrict__ A, int lda, int K, unsigned char* __restrict__ P,
                                                            float* __restrict__ pscale) {
    constexpr int MT = 2 * RG;
    __shared__ float red[8];
    const int ns = K >> 7;
    const int clip = blockIdx.x / ns, slab = blockIdx.x % ns;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const float* a = A + (size_t)clip * 32 * RG * lda + slab * 128;
    float m = 0.f;
    for (int i = tid; i < 32 * RG * 32; i += 512) {
        const float4 v = *reinterpret_cast<const float4*>(a + (size_t)(i >> 5) * lda + 4 * (i & 31));
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
    const float s = h2_scale_for(m);
    if (tid == 0) pscale[clip * 8 + slab] = s;
    unsigned char* pc = P + (size_t)clip * K * MT * 64;
    for (int f = wave; f < 4 * MT; f += 8) {               // fragment (k-step ksl of the slab, row tile mt)
        const int ksl = f / MT, mt = f % MT;
        const float* src = a + (size_t)(16 * mt + r16) * lda + 32 * ksl + 4 * kg;
        const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 16);
        uint4 h, l;
        h2_split_pair(x0.x, x0.y, s, h.x, l.x);
        h2_split_pair(x0.z, x0.w, s, h.y, l.y);
        h2_split_pair(x1.x, x1.y, s, h.z, l.z);
        h2_split_pair(x1.z, x1.w, s, h.w, l.w);
        unsigned char* d = pc + ((size_t)((slab * 4 + ksl) * 2) * MT + mt) * 1024 + lane * 16;
        *reinterpret_cast<uint4*>(d) = h;
        *reinterpret_cast<uint4*>(d + MT * 1024) = l;
    }
}
template <int RG>
__global__ __launch_bounds__(256) void h2p_to_f32_kernel(const unsigned char* __restrict__ P, const float* __restrict__ pscale, int K,
                                                          float* __restrict__ C, int ldc) {
    constexpr int MT = 2 * RG;
    const int ks2 = K >> 5;
    const long gid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // (clip, k-step, row tile)
    const int lane = threadIdx.x & 63, r16 = lane & 15, kg = lane >> 4;
    const int mt = (int)(gid % MT), ks = (int)((gid / MT) % ks2), clip = (int)(gid / ((long)MT * ks2));
    const unsigned char* s = P + (size_t)clip * K * MT * 64 + ((size_t)(ks * 2) * MT + mt) * 1024 + lane * 16;
    const f16x8 h = *reinterpret_cast<const f16x8*>(s), l = *reinterpret_cast<const f16x8*>(s + MT * 1024);
    const float inv = h2_pow2_inverse(pscale[clip * 8 + (ks >> 2)]);
    float* d = C + (size_t)(clip * 32 * RG + 16 * mt + r16) * ldc + 32 * ks + 4 * kg;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ((float)h[j] + (float)l[j]) * inv;
    *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(d + 16) = make_float4(v[4], v[5], v[6], v[7]);
}
void launch_h2p_from_f32(const float* A, int lda, int K, int B, int nwm, void* P, float* pscale, hipStream_t st) {
    const dim3 g(B * (K / 128));
#define F(M_) hipLaunchKernelGGL(h2p_from_f32_kernel<M_>, g, dim3(512), 0, st, A, lda, K, (unsigned char*)P, pscale)
    switch (nwm) { case 1: F(1); break; case 2: F(2); break; case 3: F(3); break; default: F(4); break; }
#undef F
}
void launch_h2p_to_f32(const void* P, const float* pscale, int K, int B, int nwm, float* C, int ldc, hipStream_t st) {
    const dim3 g((unsigned)((size_t)B * (K / 32) * (2 * nwm) / 4));
#define F(M_) hipLaunchKernelGGL(h2p_to_f32_kernel<M_>, g, dim3(256), 0, st, (const unsigned char*)P, pscale, K, C, ldc)
    switch (nwm) { case 1: F(1); break; case 2: F(2); break; case 3: F(3); break; default: F(4); break; }
#undef F
}

// ---------------------------------------------------------------------------------------------------
// the K loop of one tile with A taken from planes (8 waves, a wave = all 32 RG rows x 16 columns)
// ---------------------------------------------------------------------------------------------------
// Ap: the clip's planes; ratios: lane j = the factor the accumulators take before slab j (power of two; lane 0 unused).
// LDS: two K tiles of 4 MT KiB.  Loads are counted by hand (the weight fragments are inline-asm loads: with an LDS-DMA in
// flight the compiler would wait for vmcnt(0) at every use of an ordinary load's result):
//   quarter 0 of tile kt issues  B(2kt+1) [2 loads]  then  G(kt+1) [RG LDS-DMA loads into the other buffer];
//   quarter 2 issues B(2kt+2) [2];   the barrier at the end of quarter 3 needs G(kt+1) complete = at most 2 outstanding;
//   the first use of B(2kt) / B(2kt+1) allows RG + 2 younger loads outstanding.
#define H2P_WAITB(n_, a_, b_) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a_), "+v"(b_) : "n"(n_) : "memory")
template <int RG, bool SWAP>
__device__ __forceinline__ void h2p_tile_gemm(const unsigned char* __restrict__ Ap, const u32x4* __restrict__ Bpk, int K, int bn,
                                              unsigned char* lds, f32x4 (&acc)[2 * RG], float ratios) {
    constexpr int MT = 2 * RG, MH = RG, FRAG = 1024, PLANE = MT * FRAG, KSS = 2 * PLANE, BUF = 2 * KSS;
    constexpr int NG = RG;                          // 4 MT fragment images per K tile over 8 waves
    bn = __builtin_amdgcn_readfirstlane(bn);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KS2 = K >> 5, nkt = K >> 6;
    const u32x4* bp = Bpk + ((size_t)((bn >> 4) + wave) * KS2) * 128 + lane;
    const unsigned char* ag = Ap + (size_t)(wave * FRAG + lane * 16);
    const unsigned rl = (unsigned)lane * 16;
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 bq[2][2];
    auto loadB = [&](int set, int ks2) {
        ks2 = ks2 < KS2 ? ks2 : KS2 - 1;
        const u32x4* p = bp + (size_t)ks2 * 128;
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:1024"
                     : "=&v"(bq[set][0]), "=&v"(bq[set][1]) : "v"(p) : "memory");
    };
    // (inline asm, like the weight loads: an LDS-DMA the compiler can see makes it wait for vmcnt(0) before every LDS read that
    //  may alias the destination -- i.e. all of them once the buffer index is a loop variable)
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lvoid_t*)lds) + (unsigned)wave * FRAG;
    auto dma = [&](int kt, unsigned boff) {
        kt = kt < nkt ? kt : nkt - 1;
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const unsigned char* g = ag + (size_t)kt * BUF + i * 8 * FRAG;
            const unsigned la = lbase + boff + i * 8 * FRAG;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(la) : "memory", "m0");
        }
    };
    f16x8 ah[2][MH], al[MH];
    auto read_h = [&](int set, unsigned off) {
#pragma unroll
        for (int m = 0; m < MH; ++m) ah[set][m] = *reinterpret_cast<const f16x8*>(lds + off + rl + m * FRAG);
    };
    auto read_l = [&](unsigned off) {
#pragma unroll
        for (int m = 0; m < MH; ++m) al[m] = *reinterpret_cast<const f16x8*>(lds + off + rl + PLANE + m * FRAG);
    };
#define H2P_MFMA(a_, b_, hf_)                                                                                                        \
    _Pragma("unroll") for (int m = 0; m < MH; ++m)                                                                                   \
        acc[(hf_) * MH + m] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b_), (a_)[m], acc[(hf_) * MH + m], 0, 0, 0) \
                                   : __builtin_amdgcn_mfma_f32_16x16x32_f16((a_)[m], __builtin_bit_cast(f16x8, b_), acc[(hf_) * MH + m], 0, 0, 0)
#define H2P_PIN(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

    dma(0, 0);
    loadB(0, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    read_h(0, 0);
    read_l(0);
    for (int kt = 0; kt < nkt; ++kt) {
        const unsigned cur = (kt & 1) * BUF, nxt = BUF - cur;
        if (kt > 0 && (kt & 1) == 0) {              // a new 128-column slab of the producer: its scale may differ
            const float r = readlane_f(ratios, kt >> 1);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] *= r;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {               // quarter = (K32 step q >> 1, row half q & 1)
            const int t = q >> 1, hf = q & 1;
            if (hf == 0) loadB(t ^ 1, kt * 2 + t + 1);
            if (q == 0) dma(kt + 1, nxt);
            const unsigned noff = q < 3 ? cur + ((q + 1) >> 1) * KSS + ((q + 1) & 1) * MH * FRAG : nxt;
            if (q < 3) { read_h((q + 1) & 1, noff); H2P_PIN(0x100, MH); }
            if (hf == 0) H2P_WAITB(NG + 2, bq[t][0], bq[t][1]);
            H2P_MFMA(al, bq[t][0], hf);             // l_a * h_b
            H2P_PIN(0x008, MH);
            if (q == 3) {                           // tile kt+1 has landed; every wave has finished its reads of tile kt
                asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                read_h(0, noff);
                H2P_PIN(0x100, MH);
            }
            read_l(noff);
            H2P_PIN(0x100, MH);
            __builtin_amdgcn_s_setprio(1);
            H2P_MFMA(ah[q & 1], bq[t][1], hf);      // h_a * l_b
            H2P_MFMA(ah[q & 1], bq[t][0], hf);      // h_a * h_b
            H2P_PIN(0x008, 2 * MH);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    // the clamped prefetches of the last tile: their destination registers stay reserved until they have landed
    H2P_WAITB(0, bq[0][0], bq[0][1]);
#undef H2P_PIN
#undef H2P_MFMA
}

// ---------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------
// APL: A from planes (Ain = planes, sin = scales [B][8]) or from f32 rows (Ain = float [B*32RG][lda], sin = partial maxima
// [B][64] as gemm_h2.hip).  SWAP: weights-first MFMA and the row-per-lane epilogue; outputs: Cf (f32 rows, pitch ldc) and / or
// Pout + sout (planes of the N output columns).  !SWAP (X3_FWD_LAST only): gemm_h2.hip's epilogue -- f32 rows + the split-K
// partials of the last conv.  X3_BWD: actP / sact = planes of the forward activation of the block being differentiated.
template <int RG, int EPI, bool APL, bool SWAP>
__global__ __launch_bounds__(512, (RG <= 3) ? 4 : 2) void gemm_clip_h2p_kernel(
    const void* __restrict__ Ain, int lda, const float* __restrict__ sin, const u32x4* __restrict__ Bpk, const float* __restrict__ binv,
    const float* __restrict__ bias, float* __restrict__ Cf, int ldc, unsigned char* __restrict__ Pout, float* __restrict__ sout, int Tp,
    int N, int K, int tiles_n, int ntiles, float* __restrict__ rstd_io, const unsigned char* __restrict__ actP,
    const float* __restrict__ sact, const u32x4* __restrict__ Lpk, float* __restrict__ zpart, int CL) {
    constexpr int NW = 8, MT = 2 * RG, MH = RG, FRAG = 1024;
    constexpr int BUF = 2 * 2 * MT * FRAG;
    constexpr int SLABW = 128, TPITCH = SLABW + 4;
    constexpr int LASTB = 32 * RG * TPITCH * 4 > NW * MH * 3 * FRAG ? 32 * RG * TPITCH * 4 : NW * MH * 3 * FRAG;
    constexpr int LDSB = (EPI == X3_FWD_LAST && LASTB > 2 * BUF) ? LASTB : 2 * BUF;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDSB];
    __shared__ float redmax[NW];

    // block -> (clip, column slab): as gemm_clip_h2_kernel (an XCD takes a contiguous range of clips, slab-group-major)
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

    f32x4 acc[MT];
    float ainv;
    if (APL) {
        // running scale of the accumulators: slab j arrives scaled by s_j.  s_run follows s_j but never rises more than 2^50 above
        // the smallest scale seen (the slab with the largest values): a slab that far below another contributes less than 2^-50
        // of a term of the sum and may be mis-weighted; without the cap the rescaling could overflow
        const int ns = K >> 7;
        const float sj = lane < ns ? sin[clip * 8 + lane] : 1.f;
        float srun = readlane_f(sj, 0), smin = srun, rat = 1.f;
#pragma unroll
        for (int j = 1; j < 8; ++j)
            if (j < ns) {
                const float s = readlane_f(sj, j);
                smin = fminf(smin, s);
                const float snew = fminf(s, smin * 1.125899906842624e15f);
                const float r = snew * h2_pow2_inverse(srun);
                if (lane == j) rat = r;
                srun = snew;
            }
        ainv = h2_pow2_inverse(srun);
        h2p_tile_gemm<RG, SWAP>((const unsigned char*)Ain + (size_t)clip * K * MT * 64, Bpk, K, bn, lds, acc, rat);
    } else {
        float am = lane < (K >> 4) ? sin[(size_t)clip * 64 + lane] : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
        const float ascale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(h2_scale_for(am))));
        ainv = h2_pow2_inverse(ascale);
        f32x4 acc2[MT][1];
        h2_tile_gemm<RG, 8, 1, true, SWAP>((const float*)Ain, lda, Bpk, K, bm, bn, lds, acc2, 32 * RG, ascale);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = acc2[m][0];
    }
    const float invT = 1.0f / (float)Tp;

    if (SWAP) {
        // ---- lane holds row m*16 + r16 of channels n0 .. n0 + 3 ----
        const int n0 = bn + wave * 16 + 4 * kg;
        const float4 bi4 = *reinterpret_cast<const float4*>(binv + n0);
        const float us[4] = {ainv * bi4.x, ainv * bi4.y, ainv * bi4.z, ainv * bi4.w};
        float omax = 0.f;
        if (EPI == X3_FWD) {
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) b4 = *reinterpret_cast<const float4*>(bias + n0);
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
            float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bool valid = m * 16 + r16 < Tp;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[m][e] = acc[m][e] * us[e] + bv[e];
                    if (valid) s[e] += acc[m][e];
                }
            }
            float mean[4], rs[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) mean[e] = row16_sum(s[e]) * invT;
            float qq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bool valid = m * 16 + r16 < Tp;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (valid) { const float d = acc[m][e] - mean[e]; qq[e] += d * d; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) rs[e] = 1.0f / sqrtf(row16_sum(qq[e]) * invT + 1e-5f);   // biased variance, eps 1e-5
            if (r16 == 0) *reinterpret_cast<float4*>(rstd_io + (size_t)clip * N + n0) = make_float4(rs[0], rs[1], rs[2], rs[3]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bool valid = m * 16 + r16 < Tp;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = (acc[m][e] - mean[e]) * rs[e];
                    const float o = valid ? (u > 0.f ? u : 0.2f * u) : 0.f;
                    acc[m][e] = o;
                    omax = fmaxf(omax, fabsf(o));
                }
            }
        } else if (EPI == X3_BWD) {
            // acc = dL/dA of the previous block's output; result dL/dZ = rstd * (dU - mean_t dU - u * mean_t(dU*u)), dU = acc * lrelu'(u)
            const float4 r4 = *reinterpret_cast<const float4*>(rstd_io + (size_t)clip * N + n0);
            const float rs[4] = {r4.x, r4.y, r4.z, r4.w};
            const float asinv = h2_pow2_inverse(sact[clip * 8 + slab_]);
            const unsigned char* ap = actP + (size_t)clip * N * MT * 64 + ((size_t)((bn >> 5) + (wave >> 1)) * 2 * MT) * FRAG + lane * 16 +
                                      8 * (wave & 1);
            float u[MT][4];
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
            uint2 hq[MT], lq[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {          // (padding rows exist and hold zeros)
                hq[m] = *reinterpret_cast<const uint2*>(ap + m * FRAG);
                lq[m] = *reinterpret_cast<const uint2*>(ap + (MT + m) * FRAG);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bool valid = m * 16 + r16 < Tp;
                const f16x4 h4 = __builtin_bit_cast(f16x4, hq[m]), l4 = __builtin_bit_cast(f16x4, lq[m]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float av = ((float)h4[e] + (float)l4[e]) * asinv;
                    const float uv = valid ? (av > 0.f ? av : av * 5.0f) : 0.f;                 // invert LeakyReLU(0.2)
                    const float du = valid ? acc[m][e] * us[e] * (av > 0.f ? 1.f : 0.2f) : 0.f;
                    acc[m][e] = du;
                    u[m][e] = uv;
                    s1[e] += du;
                    s2[e] += du * uv;
                }
            }
            float m1[4], m2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { m1[e] = row16_sum(s1[e]) * invT; m2[e] = row16_sum(s2[e]) * invT; }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bool valid = m * 16 + r16 < Tp;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o = valid ? rs[e] * (acc[m][e] - m1[e] - u[m][e] * m2[e]) : 0.f;
                    acc[m][e] = o;
                    omax = fmaxf(omax, fabsf(o));
                }
            }
        } else {      // X3_PLAIN
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) b4 = *reinterpret_cast<const float4*>(bias + n0);
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bool valid = m * 16 + r16 < Tp;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o = valid ? acc[m][e] * us[e] + bv[e] : 0.f;
                    acc[m][e] = o;
                    omax = fmaxf(omax, fabsf(o));
                }
            }
        }
        if (Cf) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
                *reinterpret_cast<float4*>(Cf + (size_t)(bm + m * 16 + r16) * ldc + n0) = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
        }
        if (Pout) {
            omax = row16_max(omax);
            omax = fmaxf(omax, __shfl_xor(omax, 16));
            omax = fmaxf(omax, __shfl_xor(omax, 32));
            if (lane == 0) redmax[wave] = omax;
            __syncthreads();
            float mx = redmax[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) mx = fmaxf(mx, redmax[w]);
            const float so = h2_scale_for(mx);
            if (tid == 0) sout[clip * 8 + slab_] = so;
            unsigned char* op = Pout + (size_t)clip * N * MT * 64 + ((size_t)((bn >> 5) + (wave >> 1)) * 2 * MT) * FRAG + lane * 16 +
                                8 * (wave & 1);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                uint2 h, l;
                h2_split_pair(acc[m][0], acc[m][1], so, h.x, l.x);
                h2_split_pair(acc[m][2], acc[m][3], so, h.y, l.y);
                *reinterpret_cast<uint2*>(op + m * FRAG) = h;
                *reinterpret_cast<uint2*>(op + (MT + m) * FRAG) = l;
            }
        }
    } else {
        // ---- classic layout (gemm_clip_h2_kernel's epilogue): lane holds rows m*16 + 4 kg + e of column cb; X3_FWD_LAST ----
        const int col = bn + wave * 16 + r16;
        const float unscale = ainv * binv[col];
        const float bv = bias ? bias[col] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m * 16 + 4 * kg + e;
                acc[m][e] = acc[m][e] * unscale + bv;
                if (row < Tp) s += acc[m][e];
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
                if (row < Tp) { const float d = acc[m][e] - mean; qq += d * d; }
            }
        qq += __shfl_xor(qq, 16);
        qq += __shfl_xor(qq, 32);
        const float rs = 1.0f / sqrtf(qq * invT + 1e-5f);
        if (kg == 0) rstd_io[(size_t)clip * N + col] = rs;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m * 16 + 4 * kg + e;
                const float u = (acc[m][e] - mean) * rs;
                const float o = row < Tp ? (u > 0.f ? u : 0.2f * u) : 0.f;
                acc[m][e] = o;
                Cf[(size_t)(bm + row) * ldc + col] = o;
            }
        if (EPI == X3_FWD_LAST) {
            // split-K partials of the skinny last conv on the exact bf16x3 arithmetic: see gemm_clip_h2_kernel
            constexpr int KT = SLABW / 32;
            const int KS2L = N >> 5, ncl = (CL + 15) >> 4;
            const int mh = wave & 1;
            __syncthreads();
            float* const T = reinterpret_cast<float*>(lds);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) T[(16 * m + 4 * kg + e) * TPITCH + wave * 16 + r16] = acc[m][e];
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
            __syncthreads();
#pragma unroll
            for (int mm = 0; mm < MH; ++mm)
#pragma unroll
                for (int n = 0; n < 3; ++n)
                    *reinterpret_cast<f32x4*>(lds + (size_t)((wave * MH + mm) * 3 + n) * FRAG + lane * 16) = zt[mm][n];
            __syncthreads();
            if (wave < MT) {
                const int smh = wave / MH, smm = wave % MH;
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
}

bool gemm_clip_h2p_supported(int nwm, int N, int K) {
    return nwm >= 1 && nwm <= 4 && N % 128 == 0 && K % 128 == 0 && K <= 1024;
}

// A / sin: planes + scales [B][8] when a_planes, else f32 rows (pitch lda) + partial maxima [B][64].  Bpk: launch_h2_pack(perm).
// epi X3_FWD / X3_BWD / X3_PLAIN: outputs Cf (f32 rows, may be null) and / or Pout + sout (may be null).  epi X3_FWD with lastpk:
// f32 rows + zpart (a_planes only).  X3_BWD: actP / sact = planes of the forward activation (N columns).
void launch_gemm_clip_h2p(const void* A, int lda, const float* sin, bool a_planes, const void* Bpk, const float* bias, float* Cf,
                          int ldc, void* Pout, float* sout, int B, int nwm, int Tp, int N, int K, int epi, float* rstd_io,
                          const void* actP, const float* sact, hipStream_t st, const void* lastpk, float* zpart, int CL) {
    const float* binv = h2_inv_scale(Bpk, N, K);
    const int tn = N / 128;
#define HK(M_, E_, A_, S_) hipLaunchKernelGGL((gemm_clip_h2p_kernel<M_, E_, A_, S_>), dim3(tn * B), dim3(512), 0, st, A, lda, sin,     \
                                              (const u32x4*)Bpk, binv, bias, Cf, ldc, (unsigned char*)Pout, sout, Tp, N, K, tn, tn * B, \
                                              rstd_io, (const unsigned char*)actP, sact, (const u32x4*)lastpk, zpart, CL)
#define HE(M_)                                                                                                                        \
    if (epi == X3_FWD && lastpk && zpart) { HK(M_, X3_FWD_LAST, true, false); }                                                       \
    else if (epi == X3_FWD && a_planes) { HK(M_, X3_FWD, true, true); }                                                               \
    else if (epi == X3_FWD) { HK(M_, X3_FWD, false, true); }                                                                          \
    else if (epi == X3_BWD) { HK(M_, X3_BWD, true, true); }                                                                           \
    else { HK(M_, X3_PLAIN, true, true); }
    switch (nwm) { case 1: HE(1) break; case 2: HE(2) break; case 3: HE(3) break; default: HE(4) break; }
#undef HE
#undef HK
}

}  // namespace aware
