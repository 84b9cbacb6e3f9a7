// Clip-aligned detector GEMM on the bf16 matrix pipe with fp32-equivalent arithmetic (gfx950 only).
//
// gfx950 has no fast path for f32 operands: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate.
// An f32 value is EXACTLY the sum of three bf16 values (8 + 8 + 8 significand bits, round-to-nearest
// residuals): a = a0 + a1 + a2.  The product a*b is then the sum of nine bf16*bf16 partial products,
// each exact in f32; the three with i + j >= 3 are below 2^-26 |a*b| (less than one f32 rounding) and
// are dropped.  The remaining six run as six v_mfma_f32_16x16x32_bf16 with f32 accumulation:
// 6/16 of the f32-MFMA time for the same k, error vs fp64 at (or below) the level of the f32 MFMA chain
// (measured in tests/test_gpu_kernels.py::test_gemm_clip_x3).
//
// Layout decisions:
//  * the weights are constant for the 400 iterations -> split and re-ordered ONCE on the host into MFMA
//    fragment order  [N/16][K/32][plane 0..2][lane 0..63][8 bf16]  (lane l <-> column l&15, k = 8*(l>>4)+j);
//    a wave loads its B fragments straight from global memory into VGPRs as fully coalesced 1 KiB reads,
//    one K32 step ahead (no LDS for B: no other wave of the workgroup needs the same columns);
//  * the activations are split on the fly while they are staged to LDS (two buffers, one barrier per
//    64-wide K tile, which waits on LDS traffic only: global loads stay in flight across it); the LDS image
//    is the fragment order too (ds_read_b128, lane-linear up to an XOR of the row slot that keeps the
//    16-byte stores of a row's eight k-chunks on distinct banks); A fragments are read one half K32 step
//    ahead of the MFMAs that use them;
//  * one workgroup = 8 waves = all rows of ONE clip (RG groups of 32 pooled frames) x 128 columns; a wave
//    owns every row of its 16 columns (126 VGPRs: four waves per SIMD), so the InstanceNorm statistics of
//    the fused epilogues are in-register sums plus two cross-lane shuffles (no LDS reduction);
//  * the 16x16x32 MFMA shape rather than 32x32x16: measured 5-25 % faster here (the narrower wave tile
//    doubles the waves per SIMD on the half-filled grids, and the chip holds a higher clock on it).
// Reference semantics of the epilogues: detection/modules/conv1d.py:38-42 (conv -> InstanceNorm1d -> LeakyReLU).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "kernels.h"
#include "common.hpp"
#include "split_bf16.hpp"

namespace aware {

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

// either operand order (x3_pack: k16 steps of 32-column tiles; x3_pack: k32 steps of 16-column tiles), K rounded up to 32
size_t x3_packed_bytes(int N, int K) { return (size_t)N * (size_t)(((K + 31) / 32) * 32) * 3 * sizeof(uint16_t); }

// Wt: [N][K] row-major f32 (the NT operand: C = A * Wt^T), N % 16 == 0; out: x3_packed_bytes(N, K)

void x3_pack(const float* Wt, int N, int K, uint16_t* out) {
    const int KS = (K + 31) / 32;
    for (int nt = 0; nt < N / 16; ++nt)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int n = nt * 16 + (lane & 15), k = ks * 32 + 8 * (lane >> 4) + j;
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

// the same pack on the device (detector training: the weights change every step): element (n, k) of the [N][K] operand is
// Wt[n * ldw + k] for n < Nvalid, k < Kvalid and 0 otherwise (rows / k zero-padded as the read-out operands need)
__global__ __launch_bounds__(64) void x3_pack_dev_kernel(const float* __restrict__ Wt, int ldw, int Nvalid, int Kvalid, int KS,
                                                          u32x4* __restrict__ out) {
    const int ks = blockIdx.x, nt = blockIdx.y, lane = threadIdx.x;
    const int n = nt * 16 + (lane & 15), k0 = ks * 32 + 8 * (lane >> 4);
    unsigned p0[4], p1[4], p2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + 2 * j;
        const float x = (n < Nvalid && k < Kvalid) ? Wt[(size_t)n * ldw + k] : 0.f;
        const float y = (n < Nvalid && k + 1 < Kvalid) ? Wt[(size_t)n * ldw + k + 1] : 0.f;
        split_pair(x, y, p0[j], p1[j], p2[j]);
    }
    u32x4* o = out + ((size_t)(nt * KS + ks) * 3) * 64 + lane;
    o[0] = u32x4{p0[0], p0[1], p0[2], p0[3]};
    o[64] = u32x4{p1[0], p1[1], p1[2], p1[3]};
    o[128] = u32x4{p2[0], p2[1], p2[2], p2[3]};
}
void launch_x3_pack_dev(const float* Wt_dev, int ldw, int Nvalid, int Kvalid, int N, int K, void* out, hipStream_t st) {
    const int KS = (K + 31) / 32;
    hipLaunchKernelGGL(x3_pack_dev_kernel, dim3(KS, N / 16), dim3(64), 0, st, Wt_dev, ldw, Nvalid, Kvalid, KS, (u32x4*)out);
}

// The K loop of one tile: acc[m][n] += A[bm + 16 m .. +16)[0..K) * B^T for the wave's 16*NTW columns starting at
// bn + wave*16*NTW.  `lds` is the workgroup's staging memory (2 * 2 * 3 * 2RG KiB); every wave of the workgroup calls this
// with the same arguments.  The caller provides a barrier between two calls that reuse `lds`.
// row_limit: rows of A that may be read from bm on (rows beyond it are read from the last permitted row; their products
// land in accumulator rows the caller does not store).
template <int RG, int NW>
__device__ __forceinline__ void x3_tile_gemm(const float* __restrict__ A, int lda, const u32x4* __restrict__ Bpk, int K, int bm,
                                             int bn, unsigned char* lds, f32x4 (&acc)[2 * RG][8 / NW], int row_limit) {
    constexpr int NT = 64 * NW;
    constexpr int NTW = 8 / NW;           // 16-column tiles per wave (slab = 128 columns)
    constexpr int MT = 2 * RG;            // 16-row tiles per clip
    constexpr int MH = RG;                // ... per half (the unit of A-fragment double buffering)
    constexpr int FRAG = 1024;            // one 16-row x 32-k bf16 fragment image, bytes
    constexpr int PLANE = MT * FRAG;
    constexpr int KSS = 3 * PLANE;        // one K32 step
    constexpr int BUF = 2 * KSS;          // one K tile (BK = 64)
    constexpr int NCH = (RG * 256 + NT - 1) / NT;     // 8-float chunks of the A tile per thread

    // wave-uniform values are forced into SGPRs so that every global load is `saddr + 32-bit voffset` (two VGPRs less
    // per pointer than 64-bit VGPR addresses: the loop runs at the 128-VGPR limit of four waves per SIMD)
    bm = __builtin_amdgcn_readfirstlane(bm);
    bn = __builtin_amdgcn_readfirstlane(bn);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    const int srow = tid >> 3, sc = tid & 7;          // chunk i of this thread: row srow + (NT/8)*i, k = 8*sc
    unsigned rb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) rb[t] = (unsigned)(t * KSS + kg * 256 + ((r16 ^ (4 * t + kg)) * 16));

    const int KS2 = K >> 5;
    const int nkt = K >> 6;
    const u32x4* bp = Bpk + ((size_t)((bn >> 4) + wave * NTW) * KS2) * 192;          // uniform; + lane per thread
    const float* ap = A + (size_t)bm * lda;                                          // uniform
    unsigned roff[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) roff[i] = (unsigned)(min(srow + (NT / 8) * i, row_limit - 1) * lda + sc * 8);

#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 ra[NCH][2];
    auto chunk_ok = [&](int i) { return (RG * 256) % NT == 0 || tid + NT * i < RG * 256; };
    auto gload_c = [&](int i, int kt) {
        if (chunk_ok(i)) {
            const float* p = ap + kt * 64;
            ra[i][0] = *reinterpret_cast<const float4*>(p + roff[i]);
            ra[i][1] = *reinterpret_cast<const float4*>(p + roff[i] + 4);
        }
    };
    auto split_store_c = [&](int i, unsigned boff) {
        if (chunk_ok(i)) {
            const int row = srow + (NT / 8) * i;
            uint4 q0, q1, q2;
            split_pair(ra[i][0].x, ra[i][0].y, q0.x, q1.x, q2.x);
            split_pair(ra[i][0].z, ra[i][0].w, q0.y, q1.y, q2.y);
            split_pair(ra[i][1].x, ra[i][1].y, q0.z, q1.z, q2.z);
            split_pair(ra[i][1].z, ra[i][1].w, q0.w, q1.w, q2.w);
            unsigned char* d = lds + boff + (sc >> 2) * KSS + (row >> 4) * FRAG + (sc & 3) * 256 + (((row & 15) ^ sc) * 16);
            *reinterpret_cast<uint4*>(d) = q0;
            *reinterpret_cast<uint4*>(d + PLANE) = q1;
            *reinterpret_cast<uint4*>(d + 2 * PLANE) = q2;
        }
    };
    u32x4 bq[2][3][NTW];
    auto loadB = [&](int set, int ks2) {
        ks2 = ks2 < KS2 ? ks2 : KS2 - 1;
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int p = 0; p < 3; ++p) bq[set][p][n] = (bp + ((size_t)n * KS2 + ks2) * 192 + p * 64)[(unsigned)lane];
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // A fragments are single-buffered and refilled on a rolling schedule: the six partial products of a quarter go
    // smallest first -- (a2,b0) (a1,b1) (a0,b2) (a1,b0) (a0,b1) (a0,b0) -- so plane 2 is free after the first product,
    // plane 1 after the fourth and plane 0 after the sixth; the next quarter's fragments are read into each plane as soon
    // as it is free, 6-15 MFMAs before their first use.  (Double-buffering all three planes needs 72 VGPRs and pushed
    // the kernel past the 128-VGPR budget of four waves per SIMD: the compiler then issued each read right before its use.)
    bf16x8 af[3][MH];
    auto read_plane = [&](int p, unsigned off) {       // off: buffer + rb[K32 step] + half * MH * FRAG
#pragma unroll
        for (int m = 0; m < MH; ++m) af[p][m] = *reinterpret_cast<const bf16x8*>(lds + off + p * PLANE + m * FRAG);
    };
    auto product = [&](int pa, int pb, int t, int hf) {
#pragma unroll
        for (int m = 0; m < MH; ++m)
#pragma unroll
            for (int n = 0; n < NTW; ++n)
                acc[hf * MH + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[pa][m], __builtin_bit_cast(bf16x8, bq[t][pb][n]),
                                                                              acc[hf * MH + m][n], 0, 0, 0);
    };
    // the issue order is pinned (0x008 = MFMA, 0x100 = LDS read): left alone, the scheduler bunches the reads
#define X3_PIN(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

#pragma unroll
    for (int i = 0; i < NCH; ++i) gload_c(i, 0);
    loadB(0, 0);
#pragma unroll
    for (int i = 0; i < NCH; ++i) split_store_c(i, 0);
#pragma unroll
    for (int i = 0; i < NCH; ++i) gload_c(i, nkt > 1 ? 1 : 0);
    lds_barrier();
#pragma unroll
    for (int p = 0; p < 3; ++p) read_plane(p, rb[0]);
    for (int kt = 0; kt < nkt; ++kt) {
        const unsigned cur = (kt & 1) * BUF, nxt = BUF - cur;
        const int ktn = kt + 2 < nkt ? kt + 2 : nkt - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                 // quarter = (K32 step q>>1, row half q&1)
            const int t = q >> 1, hf = q & 1;
            if (hf == 0) loadB((t + 1) & 1, kt * 2 + t + 1);          // B fragments one K32 step ahead
            if (q < NCH) {
                split_store_c(q, nxt);
                gload_c(q, ktn);
            }
            const unsigned noff = q < 3 ? cur + rb[(q + 1) >> 1] + ((q + 1) & 1) * MH * FRAG : nxt + rb[0];
            product(2, 0, t, hf);
            X3_PIN(0x008, MH * NTW);
            if (q == 3) lds_barrier();       // tile kt+1 is complete; every wave has issued and finished its reads of tile kt
            read_plane(2, noff);
            X3_PIN(0x100, MH);
            product(1, 1, t, hf);
            product(0, 2, t, hf);
            product(1, 0, t, hf);
            X3_PIN(0x008, 3 * MH * NTW);
            read_plane(1, noff);
            X3_PIN(0x100, MH);
            product(0, 1, t, hf);
            product(0, 0, t, hf);
            X3_PIN(0x008, 2 * MH * NTW);
            read_plane(0, noff);
            X3_PIN(0x100, MH);
        }
    }
#undef X3_PIN
}

// (RG <= 3: two 8-wave workgroups per CU = four waves per SIMD need <= 128 VGPRs; RG = 4 does not fit that)
template <int RG, int EPI, int NW>
__global__ __launch_bounds__(64 * NW, (RG <= 3 && NW == 8) ? 4 : 2) void gemm_clip_x3_kernel(const float* __restrict__ A, int lda,
                                                                 const u32x4* __restrict__ Bpk, const float* __restrict__ bias,
                                                                 float* __restrict__ C, int ldc, int Tp, int N, int K,
                                                                 int tiles_n, int ntiles, float* __restrict__ rstd_io,
                                                                 const float* __restrict__ act,
                                                                 const u32x4* __restrict__ Lpk, float* __restrict__ zpart,
                                                                 int CL, int Mrows) {
    // Mrows (X3_PLAIN only): rows of A and C that exist; the last row block may be partial (reads clamped, stores masked)
    constexpr int NTW = 8 / NW;           // 16-column tiles per wave (slab = 128 columns)
    constexpr int MT = 2 * RG;            // 16-row tiles per clip
    constexpr int MH = RG;
    constexpr int FRAG = 1024;
    constexpr int BUF = 2 * 3 * MT * FRAG;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];

    // Blocks b and b + 8 share an XCD (observed round-robin placement; speed only).  An XCD takes a contiguous range of clips and
    // walks it slab-group-major: `sg` column slabs at a time whose packed weights (sg * 128 * K * 6 bytes) fit its 4 MB L2
    // together with the activation rows in flight, all clips of the range for that group, then the next group.  Clip-major
    // order streamed all 6.3 MB of a 1024 x 1024 layer through the L2 for every clip (PMC: 565 MB fetched per launch for 206 MB
    // of operands); this order fetches the weights once per XCD and the activation rows once per group.
    int id = blockIdx.x;
    int clip, slab_;
    if ((ntiles & 7) == 0) {
        const int x = id & 7, j = id >> 3, R = ntiles >> 3;          // XCD, index inside its range, workgroups per XCD
        const int nclip = R / tiles_n;                                // clips per XCD (ntiles = clips * tiles_n, clips % 8 == 0 here
        if (nclip * tiles_n == R && nclip > 0) {                     //  whenever the batch size is a multiple of 8)
            int sg = (int)(3355443u / (unsigned)(128 * K * 6));      // slabs whose weights fit 3.2 MB
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
    const int bn = slab_ * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;

    f32x4 acc[MT][NTW];
    x3_tile_gemm<RG, NW>(A, lda, Bpk, K, bm, bn, lds, acc, EPI == X3_PLAIN ? min(32 * RG, Mrows - bm) : 32 * RG);

    // ---- epilogue: lane holds rows m*16 + 4*kg + e (e = 0..3) of columns cb + n*16 + r16 ----
    const int cb = bn + wave * (16 * NTW) + r16;
    const float invT = 1.0f / (float)Tp;
    if (EPI == X3_PLAIN && NTW == 1 && (ldc & 3) == 0) {
        // The tile leaves in ROW-MAJOR order (16 bytes per lane, half a wave = one 512-byte row segment), not in the
        // accumulator's layout (4 rows x 64 bytes per wave instruction, which streams at about half the rate: measured on
        // the read-out kernel, 3.0 vs 6 TB/s).  The staging LDS is free now; pitch 132 floats keeps both sides conflict-free.
        float (*T)[132] = reinterpret_cast<float (*)[132]>(lds);
        const float bv = bias ? bias[cb] : 0.f;
        __syncthreads();                                // every wave has left the K loop
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) T[16 * m + 4 * kg + e][16 * wave + r16] = acc[m][0][e] + bv;
        __syncthreads();
        const int c4 = (lane & 31) * 4, rr = 2 * wave + (lane >> 5);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int row = rr + 16 * j;
            float4 o = *reinterpret_cast<const float4*>(&T[row][c4]);
            if (row >= Tp) o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bm + row < Mrows) *reinterpret_cast<float4*>(C + (size_t)(bm + row) * ldc + bn + c4) = o;
        }
        return;
    }
    // (the forward / backward epilogues keep the accumulator-layout stores: behind the K loop of a second resident workgroup
    //  they are hidden -- the row-major form measured the same time on the three conv blocks)
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int col = cb + n * 16;
        if (EPI == X3_PLAIN) {
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m * 16 + 4 * kg + e;
                    if (bm + row < Mrows) C[(size_t)(bm + row) * ldc + col] = row < Tp ? acc[m][n][e] + bv : 0.f;
                }
        } else if (EPI == X3_FWD || EPI == X3_FWD_LAST) {
            const float bv = bias ? bias[col] : 0.f;
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = m * 16 + 4 * kg + e;
                    acc[m][n][e] += bv;
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
                    const float du = valid ? acc[m][n][e] * (av > 0.f ? 1.f : 0.2f) : 0.f;
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
                    C[(size_t)(bm + row) * ldc + col] = row < Tp ? rs * (acc[m][n][e] - m1 - u[m][e] * m2) : 0.f;
                }
        }
    }
    if (EPI == X3_FWD_LAST && NTW == 1) {
        // acc[m][0][e] holds this block's output (zero in padding rows).  The next conv block is the skinny last one
        // (CL <= 64 channels): its K = this N is split over the column slabs, so this workgroup contributes the partial
        // z_part[slab] = out[:, slab] * Wlast[:, slab]^T.  The output tile is re-laid as A fragments (k = column) in LDS.
        // Work split: wave w takes K32 step t = w>>1 of the slab's 128 columns and half mh = w&1 of the row tiles (all
        // column tiles of the last conv, CL <= 48); the four t-partials are then summed through LDS.
        const int slab = bn >> 7, KS2L = N >> 5, ncl = (CL + 15) >> 4;
        const int tq = wave >> 1, mh = wave & 1;
        bf16x8 bl[3][3];
#pragma unroll
        for (int n = 0; n < 3; ++n)
            if (n < ncl) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bl[n][p] = __builtin_bit_cast(bf16x8, Lpk[(((size_t)n * KS2L + 4 * slab + tq) * 3 + p) * 64 + lane]);
            }
        __syncthreads();                                  // every wave is done with the staging buffers
        // the output tile goes through LDS as f32 [row][column], row pitch 132 floats (conflict-free 4-byte stores from
        // the accumulator layout); each wave reads its A fragments back as 8 consecutive columns per lane and splits them
        float* const T = reinterpret_cast<float*>(lds);
        constexpr int TP = 132;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) T[(16 * m + 4 * kg + e) * TP + 16 * wave + r16] = acc[m][0][e];
        __syncthreads();
        f32x4 zt[MH][3];
#pragma unroll
        for (int mm = 0; mm < MH; ++mm) {
#pragma unroll
            for (int n = 0; n < 3; ++n) zt[mm][n] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* src = T + (16 * (mh * MH + mm) + r16) * TP + 32 * tq + 8 * kg;
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
        __syncthreads();                                  // all fragment reads done: the buffer becomes the partial store
#pragma unroll
        for (int mm = 0; mm < MH; ++mm)
#pragma unroll
            for (int n = 0; n < 3; ++n)
                *reinterpret_cast<f32x4*>(lds + (size_t)((wave * MH + mm) * 3 + n) * FRAG + lane * 16) = zt[mm][n];
        __syncthreads();
        if (wave < MT) {
            const int smh = wave / MH, smm = wave % MH;   // this wave finishes row tile `wave`
            float* zp = zpart + (size_t)slab * ((size_t)(ntiles / tiles_n) * 32 * RG * CL) + (size_t)(bm + 16 * wave + 4 * kg) * CL;
#pragma unroll
            for (int n = 0; n < 3; ++n)
                if (n < ncl) {
                    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
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
// Ragged batches: the same conv block (or data-gradient GEMM) for clips of ANY length in ONE launch.
// A workgroup owns one clip x 128 columns, as above, but takes the clip's rows from the batch tables (pool_off is
// 32-aligned per clip, Tp = frames / 2) and walks them in chunks of at most three 32-row groups (72 KB of staging memory
// and <= 128 VGPRs: two workgroups per CU at any clip length).  A clip that fits one chunk gets the same single-pass
// epilogue as the uniform kernel (bit-identical results).  A longer clip is done in two passes over its chunks:
//   pass 1  GEMM of the chunk, raw result to C, per-column statistics carried in registers across the chunks
//           (forward: count / mean / M2 merged with Chan's formula; backward: the two sums of the InstanceNorm backward);
//   pass 2  the workgroup re-reads its own raw tile (L2-resident, written by the same lanes) and applies the
//           normalisation + LeakyReLU (forward) or the InstanceNorm backward (backward) in place.
// No second launch, no inter-workgroup traffic, the per-(clip, channel) statistics never leave the registers.
// Reference: detection/modules/conv1d.py:38-42 and its autograd.
// ---------------------------------------------------------------------------------------------------
constexpr int kRaggedRG = 3;           // largest chunk, in 32-row groups

// one chunk: rows [bm, bm + 32 RG) of which `rows` are valid.  SINGLE: the clip is this chunk.
// st0/st1/st2: forward (count, mean, M2) of the column; backward (unused, sum dU, sum dU*u) in-lane partial sums
extern __shared__ __attribute__((aligned(16))) unsigned char x3_dyn_lds[];    // the ragged kernel's staging memory (dynamic LDS)

template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    // a wave-uniform pointer to GLOBAL memory that arrived as a function argument: in VGPRs and in the generic address
    // space (flat loads count on vmcnt AND lgkmcnt, so every wait in the K loop became a full drain of both).  Back to an
    // SGPR pair, and through address space 1 so that the loads are global_load again.
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(__attribute__((address_space(1))) T*)(((unsigned long long)hi << 32) | lo);
}

// (not inlined: each tile height keeps its own register allocation -- inlined side by side the two K loops cost the
// kernel 20-40 spilled VGPRs inside the loop)
template <int RG, int EPI>
__device__ __attribute__((noinline)) void x3_ragged_chunk(const bool SINGLE, const float* __restrict__ A, int lda, const u32x4* __restrict__ Bpk,
                                                const float* __restrict__ bias, float* __restrict__ C, int ldc, int N, int K,
                                                int bm, int rows, int store_rows, int bn,
                                                float* __restrict__ rstd_clip, const float* __restrict__ act, float& st0,
                                                float& st1, float& st2) {
    unsigned char* lds = x3_dyn_lds;
    // the arguments of a non-inlined function arrive in VGPRs; all of these are wave-uniform and go back to SGPRs (the K
    // loop runs at the 128-VGPR limit: left in VGPRs they were spilled and reloaded inside it)
    A = uniform_ptr(A); Bpk = uniform_ptr(Bpk); bias = uniform_ptr(bias); C = uniform_ptr(C);
    rstd_clip = uniform_ptr(rstd_clip); act = uniform_ptr(act);
    lda = __builtin_amdgcn_readfirstlane(lda); ldc = __builtin_amdgcn_readfirstlane(ldc);
    N = __builtin_amdgcn_readfirstlane(N); K = __builtin_amdgcn_readfirstlane(K);
    bm = __builtin_amdgcn_readfirstlane(bm); bn = __builtin_amdgcn_readfirstlane(bn);
    rows = __builtin_amdgcn_readfirstlane(rows); store_rows = __builtin_amdgcn_readfirstlane(store_rows);
    // rows: valid rows of the chunk; store_rows (a multiple of 32, <= 32 RG): rows of the clip's allocation under this tile
    constexpr int MT = 2 * RG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    f32x4 acc[MT][1];
    x3_tile_gemm<RG, 8>(A, lda, Bpk, K, bm, bn, lds, acc, store_rows);
    const int col = bn + wave * 16 + r16;
    const float invR = 1.0f / (float)rows;
    if (EPI == X3_FWD) {
        const float bv = bias ? bias[col] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[m][0][e] += bv;
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
                    if (row < store_rows) C[(size_t)(bm + row) * ldc + col] = row < rows ? (u > 0.f ? u : 0.2f * u) : 0.f;
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
                const float du = valid ? acc[m][0][e] * (av > 0.f ? 1.f : 0.2f) : 0.f;
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
                    if (row < store_rows) C[(size_t)(bm + row) * ldc + col] = row < rows ? rs * (acc[m][0][e] - m1 - u[m][e] * m2) : 0.f;
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
__global__ __launch_bounds__(512, 4) void gemm_ragged_x3_kernel(const float* __restrict__ A, int lda, const u32x4* __restrict__ Bpk,
                                                                const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                                const int* __restrict__ frame_off, const int* __restrict__ pool_off,
                                                                const int* __restrict__ order, int N, int K, int tiles_n, int ntiles,
                                                                float* __restrict__ rstd_io, const float* __restrict__ act) {
    // blocks b and b + 8 share an XCD (observed round-robin placement; speed only): the slabs of one clip stay on one XCD
    // (its rows are read once into that L2), and clips are dealt to the XCDs round-robin -- batches arrive sorted by
    // length, a contiguous range per XCD would give one XCD all the long clips
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
        if (order) clip = order[clip];                              // dispatch position -> clip: longest clips first
    }
    const int bn = slab * 128;
    const int Tp = (frame_off[clip + 1] - frame_off[clip]) / 2;
    const int row0 = pool_off[clip];
    if (Tp < 1) return;
    const int G = (Tp + 31) >> 5;                                   // 32-row groups of the clip
    const int nchunk = (G + kRaggedRG - 1) / kRaggedRG;
    const int gbase = G / nchunk, grem = G % nchunk;                // balanced: the first `grem` chunks take one group more
    float* rstd_clip = rstd_io + (size_t)clip * N;
    float st0 = 0.f, st1 = 0.f, st2 = 0.f;
    const bool single = nchunk == 1;
    int g0 = 0;
    for (int c = 0; c < nchunk; ++c) {
        const int ng = gbase + (c < grem ? 1 : 0);
        const int bm = row0 + 32 * g0;
        const int rows = min(32 * ng, Tp - 32 * g0);
        if (c) __syncthreads();                                     // every wave is done with the previous chunk's staging memory
        // two tile heights only (a one-group chunk runs as a two-group tile whose second group is padding: the K-order of
        // every output element is the same at any tile height, so the results do not depend on the choice)
        if (ng <= 2) x3_ragged_chunk<2, EPI>(single, A, lda, Bpk, bias, C, ldc, N, K, bm, rows, 32 * ng, bn, rstd_clip, act, st0, st1, st2);
        else x3_ragged_chunk<3, EPI>(single, A, lda, Bpk, bias, C, ldc, N, K, bm, rows, 32 * ng, bn, rstd_clip, act, st0, st1, st2);
        g0 += ng;
    }
    if (single) return;
    // ---- pass 2 over the raw tile this workgroup wrote: ROW-MAJOR (lane = 4 consecutive columns, half a wave = one 512-byte
    // row segment), the per-column statistics handed over through LDS.  In the accumulator's layout (the lane that wrote a
    // value reads it back: 4 rows x 64 bytes per wave instruction) this pass streamed at about half the rate. ----
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int col = bn + wave * 16 + r16;
    const float invT = 1.0f / (float)Tp;
    const int npad = 32 * G;
    float* cstat = reinterpret_cast<float*>(x3_dyn_lds);           // [2][128]; the staging memory is free now
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
                    return (r < Tp) ? (u > 0.f ? u : 0.2f * u) : 0.f;
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
                    return (r < Tp) ? rs * (d - m1 - uv * m2) : 0.f;
                };
                *reinterpret_cast<float4*>(Cw + (size_t)r * ldc) =
                    make_float4(f(du[j].x, av[j].x, rs4.x, q0.x, q1.x), f(du[j].y, av[j].y, rs4.y, q0.y, q1.y),
                                f(du[j].z, av[j].z, rs4.z, q0.z, q1.z), f(du[j].w, av[j].w, rs4.w, q0.w, q1.w));
            }
        }
    }
}

// epi: 1 forward (conv + InstanceNorm + LeakyReLU), 2 backward (data gradient + InstanceNorm/LeakyReLU backward of the
// previous block); frame_off / pool_off: the batch's device tables
void launch_gemm_ragged_x3(const float* A, int lda, const void* Bpk, const float* bias, float* C, int ldc, int B,
                           const int* frame_off, const int* pool_off, const int* order, int N, int K, int epi, float* rstd_io,
                           const float* act, hipStream_t st) {
    const int tn = N / 128;
    constexpr size_t kLds = 2 * 2 * 3 * (2 * kRaggedRG) * 1024;      // two K tiles of the tallest chunk
    if (epi == X3_FWD)
        hipLaunchKernelGGL((gemm_ragged_x3_kernel<X3_FWD>), dim3(tn * B), dim3(512), kLds, st, A, lda, (const u32x4*)Bpk, bias, C, ldc,
                           frame_off, pool_off, order, N, K, tn, tn * B, rstd_io, act);
    else
        hipLaunchKernelGGL((gemm_ragged_x3_kernel<X3_BWD>), dim3(tn * B), dim3(512), kLds, st, A, lda, (const u32x4*)Bpk, bias, C, ldc,
                           frame_off, pool_off, order, N, K, tn, tn * B, rstd_io, act);
}

// ---------------------------------------------------------------------------------------------------
// Small-batch (latency) variant of the conv block: when clips x column slabs cannot fill the chip (fewer than ~256
// workgroups of the kernel above, e.g. the single clip of the service API), one workgroup takes only 16 columns of
// one clip and its 8 waves split K (K32 steps wave, wave+8, ...): every wave multiplies all rows of the clip with A
// split in registers straight from global memory -- no LDS staging, no barrier in the loop -- and the 8 partial
// tiles meet in LDS.  64 workgroups per clip at N = 1024 instead of 8; same arithmetic, same epilogues.
// ---------------------------------------------------------------------------------------------------
template <int RG, int EPI>
__global__ __launch_bounds__(512) void gemm_clip_x3_small_kernel(const float* __restrict__ A, int lda,
                                                                  const u32x4* __restrict__ Bpk, const float* __restrict__ bias,
                                                                  float* __restrict__ C, int ldc, int Tp, int N, int K,
                                                                  int tiles_n, float* __restrict__ rstd_io,
                                                                  const float* __restrict__ act) {
    constexpr int MT = 2 * RG;
    constexpr int FRAG = 1024;
    __shared__ __attribute__((aligned(16))) unsigned char part[8 * MT * FRAG];
    __shared__ float red[2][8][16];

    const int clip = blockIdx.x / tiles_n, nt = blockIdx.x % tiles_n;          // nt: 16-column tile
    const int bm = clip * 32 * RG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int KS2 = K >> 5;
    const int nsteps = (KS2 - wave + 7) >> 3;             // K32 steps wave, wave + 8, ...

    f32x4 zp[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) zp[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nsteps > 0) {
        const float* abase = A + (size_t)(bm + r16) * lda + 8 * kg;
        const u32x4* bp = Bpk + (size_t)nt * KS2 * 192 + lane;
        float4 an[MT][2];
        u32x4 bb[2][3];
        auto loadA = [&](int m, int i) {
            i = i < nsteps ? i : nsteps - 1;
            const float* p = abase + (size_t)(16 * m) * lda + (wave + 8 * i) * 32;
            an[m][0] = *reinterpret_cast<const float4*>(p);
            an[m][1] = *reinterpret_cast<const float4*>(p + 4);
        };
        auto loadB = [&](int set, int i) {
            i = i < nsteps ? i : nsteps - 1;
#pragma unroll
            for (int p = 0; p < 3; ++p) bb[set][p] = bp[(size_t)(wave + 8 * i) * 192 + p * 64];
        };
        auto step = [&](int set, int i) {
            loadB(set ^ 1, i + 1);
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, bb[set][0]), b1 = __builtin_bit_cast(bf16x8, bb[set][1]),
                         b2 = __builtin_bit_cast(bf16x8, bb[set][2]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                uint4 q0, q1, q2;
                split_pair(an[m][0].x, an[m][0].y, q0.x, q1.x, q2.x);
                split_pair(an[m][0].z, an[m][0].w, q0.y, q1.y, q2.y);
                split_pair(an[m][1].x, an[m][1].y, q0.z, q1.z, q2.z);
                split_pair(an[m][1].z, an[m][1].w, q0.w, q1.w, q2.w);
                loadA(m, i + 1);                       // the same registers, one whole step ahead of their use
                const bf16x8 a0 = __builtin_bit_cast(bf16x8, q0), a1 = __builtin_bit_cast(bf16x8, q1),
                             a2 = __builtin_bit_cast(bf16x8, q2);
                zp[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0, zp[m], 0, 0, 0);
                zp[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, zp[m], 0, 0, 0);
                zp[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b2, zp[m], 0, 0, 0);
                zp[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, zp[m], 0, 0, 0);
                zp[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, zp[m], 0, 0, 0);
                zp[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, zp[m], 0, 0, 0);
            }
        };
#pragma unroll
        for (int m = 0; m < MT; ++m) loadA(m, 0);
        loadB(0, 0);
        for (int i = 0; i < nsteps; i += 2) {
            step(0, i);
            if (i + 1 < nsteps) step(1, i + 1);
            else { bb[0][0] = bb[1][0]; bb[0][1] = bb[1][1]; bb[0][2] = bb[1][2]; }
        }
    }
    // the 8 partial tiles meet in LDS; wave m (< MT) finishes the 16 rows of row tile m
#pragma unroll
    for (int m = 0; m < MT; ++m) *reinterpret_cast<f32x4*>(part + (size_t)(wave * MT + m) * FRAG + lane * 16) = zp[m];
    __syncthreads();
    const bool wv = wave < MT;
    f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wv) {
#pragma unroll
        for (int src = 0; src < 8; ++src) z += *reinterpret_cast<const f32x4*>(part + (size_t)(src * MT + wave) * FRAG + lane * 16);
    }
    // column sums over the clip's rows: in-lane -> the wave's four row groups -> across the MT waves (LDS)
    auto colsum = [&](float x, int stage) {
        x += __shfl_xor(x, 16);
        x += __shfl_xor(x, 32);
        if (kg == 0) red[stage][wave][r16] = x;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < MT; ++w) t += red[stage][w][r16];
        return t;
    };
    const int col = nt * 16 + r16;
    const int row0 = 16 * wave + 4 * kg;
    const float invT = 1.0f / (float)Tp;
    if (EPI == X3_PLAIN) {
        if (!wv) return;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) C[(size_t)(bm + row0 + e) * ldc + col] = row0 + e < Tp ? z[e] + bv : 0.f;
        return;
    }
    if (EPI == X3_FWD) {
        const float bv = bias ? bias[col] : 0.f;
        float sacc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { z[e] += bv; if (wv && row0 + e < Tp) sacc += z[e]; }
        const float mean = colsum(sacc, 0) * invT;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (wv && row0 + e < Tp) { const float d = z[e] - mean; q += d * d; }
        const float rs = 1.0f / sqrtf(colsum(q, 1) * invT + 1e-5f);
        if (!wv) return;
        if (wave == 0 && kg == 0) rstd_io[(size_t)clip * N + col] = rs;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = (z[e] - mean) * rs;
            C[(size_t)(bm + row0 + e) * ldc + col] = row0 + e < Tp ? (u > 0.f ? u : 0.2f * u) : 0.f;
        }
        return;
    }
    {   // X3_BWD
        const float rs = rstd_io[(size_t)clip * N + col];
        float du[4], uv[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int rr = wv ? row0 + e : 0;                                   // idle waves read a valid row, masked below
            const float av = act[(size_t)(bm + rr) * ldc + col];
            const bool valid = wv && row0 + e < Tp;
            uv[e] = valid ? (av > 0.f ? av : av * 5.0f) : 0.f;
            du[e] = valid ? z[e] * (av > 0.f ? 1.f : 0.2f) : 0.f;
            s1 += du[e]; s2 += du[e] * uv[e];
        }
        const float m1 = colsum(s1, 0) * invT, m2 = colsum(s2, 1) * invT;
        if (!wv) return;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            C[(size_t)(bm + row0 + e) * ldc + col] = row0 + e < Tp ? rs * (du[e] - m1 - uv[e] * m2) : 0.f;
    }
}

// fewer than this many workgroups of the throughput kernel: use the latency variant
constexpr int kSmallGrid = 128;

void launch_gemm_clip_x3(const float* A, int lda, const void* Bpk, const float* bias, float* C, int ldc, int B, int nwm,
                         int Tp, int N, int K, int epi, float* rstd_io, const float* act, hipStream_t st,
                         const void* lastpk, float* zpart, int CL, int Mrows) {
    const int tn = N / 128;
    if (Mrows <= 0) Mrows = B * 32 * nwm;
    if (tn * B < kSmallGrid && Mrows == B * 32 * nwm && !(epi == X3_FWD && lastpk && zpart) && K % 32 == 0) {
        const int t16 = N / 16;
#define SK(M_, E_) hipLaunchKernelGGL((gemm_clip_x3_small_kernel<M_, E_>), dim3(t16 * B), dim3(512), 0, st, A, lda,          \
                                      (const u32x4*)Bpk, bias, C, ldc, Tp, N, K, t16, rstd_io, act)
#define SM(E_) switch (nwm) { case 1: SK(1, E_); break; case 2: SK(2, E_); break; case 3: SK(3, E_); break; default: SK(4, E_); break; }
        if (epi == X3_FWD) { SM(X3_FWD) } else if (epi == X3_BWD) { SM(X3_BWD) } else { SM(X3_PLAIN) }
#undef SM
#undef SK
        return;
    }
    if (epi == X3_FWD && lastpk && zpart) epi = X3_FWD_LAST;
#define XK(M_, E_) hipLaunchKernelGGL((gemm_clip_x3_kernel<M_, E_, 8>), dim3(tn * B), dim3(512), 0, st, A, lda,          \
                                      (const u32x4*)Bpk, bias, C, ldc, Tp, N, K, tn, tn * B, rstd_io, act,                \
                                      (const u32x4*)lastpk, zpart, CL, Mrows)
#define XM(E_) switch (nwm) { case 1: XK(1, E_); break; case 2: XK(2, E_); break; case 3: XK(3, E_); break; default: XK(4, E_); break; }
    if (epi == X3_FWD) { XM(X3_FWD) } else if (epi == X3_BWD) { XM(X3_BWD) } else if (epi == X3_FWD_LAST) { XM(X3_FWD_LAST) }
    else { XM(X3_PLAIN) }
#undef XM
#undef XK
}

// ---------------------------------------------------------------------------------------------------
// Mel front end of the detector for a uniform batch of clips of at most 192 frames, ONE launch, one workgroup per clip:
// mel projection (K = 256 band bins, N = 128) on the bf16x3 tile GEMM with all of the clip's rows in one tile, then
// InstanceNorm1d(128) over time, GlobalStandardize over the clip, AvgPool1d(2, 2) in the epilogue (the statistics are
// in-register sums, two shuffles and one exchange through LDS).  Replaces the plain mel GEMM + mel_norm_clip_fwd_kernel
// (24 + 11.5 us at B = 256: both latency-bound) and the 25 MB round trip of the mel tile between them; the raw mel tile
// is still written once, for the backward kernel.
//   reference: detection/modules/mel.py:185-201, multibit_detector_net.py:50,126-131, modules/globalStandardize.py:16-21
// ---------------------------------------------------------------------------------------------------
template <int RG>
__global__ __launch_bounds__(512, 2) void mel_front_x3_kernel(const float* __restrict__ mag, int lda, const u32x4* __restrict__ Bpk,
                                                               const int* __restrict__ frame_off, const int* __restrict__ pool_off,
                                                               float* __restrict__ xm, float* __restrict__ x0,
                                                               float* __restrict__ stats, float* __restrict__ gstat, int K,
                                                               float* __restrict__ amax_out) {
    constexpr int MT = 2 * RG;
    constexpr int FRAG = 1024;
    constexpr int BUF = 2 * 3 * MT * FRAG;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
    __shared__ float red8[8];
    const int b = blockIdx.x;
    const int f0 = frame_off[b], T = frame_off[b + 1] - f0, Tp = T / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    f32x4 acc[MT][1];
    x3_tile_gemm<RG, 8>(mag, lda, Bpk, K, f0, 0, lds, acc, T);         // rows beyond T read row T-1 again; masked below
    const int c = wave * 16 + r16;
    const float fT = (float)T;
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (m * 16 + 4 * kg + e < T) s += acc[m][0][e];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mu = s / fT;
    float q = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (m * 16 + 4 * kg + e < T) { const float d = acc[m][0][e] - mu; q += d * d; }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    const float M2 = q;
    const float rs = 1.0f / sqrtf(M2 / fT + 1e-5f);                    // biased var, eps 1e-5 (InstanceNorm1d)
    // GlobalStandardize: unbiased std of the normalised tile (its mean is 0): sum over channels of rs^2 M2
    float su = (kg == 0) ? rs * rs * M2 : 0.f;
    su = wave_sum(su);
    __syncthreads();                                                   // every wave has left the K loop (LDS is reused below)
    if (lane == 0) red8[wave] = su;
    // the raw mel tile, for the backward kernel: row-major through LDS (pitch 132)
    float (*Tm)[132] = reinterpret_cast<float (*)[132]>(lds);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) Tm[16 * m + 4 * kg + e][c] = acc[m][0][e];
    __syncthreads();
    const float suu = ((red8[0] + red8[1]) + (red8[2] + red8[3])) + ((red8[4] + red8[5]) + (red8[6] + red8[7]));
    const float n = fT * 128.f;
    const float gs = sqrtf(suu / (n - 1.f));
    const float ginv = 1.0f / (gs + 1e-8f);
    if (kg == 0) { float* stp = stats + ((size_t)b * 128 + c) * 4; stp[0] = mu; stp[1] = rs; stp[2] = M2; stp[3] = 0.f; }
    if (tid == 0) { gstat[b * 4 + 0] = ginv; gstat[b * 4 + 1] = gs; gstat[b * 4 + 2] = n; gstat[b * 4 + 3] = fT; }
    const int c4 = (lane & 31) * 4, rr = 2 * wave + (lane >> 5);
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int row = rr + 16 * j;
        if (row < T) *reinterpret_cast<float4*>(xm + (size_t)(f0 + row) * 128 + c4) = *reinterpret_cast<const float4*>(&Tm[row][c4]);
    }
    __syncthreads();                                                   // the tile has been read: the buffer takes the pooled tile
    // pooled rows: the frame pair (2 tp, 2 tp + 1) is the register pair (e, e + 1) of one lane
    float pmax = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int tp = (m * 16 + 4 * kg) / 2 + h;
            const float u0 = (acc[m][0][2 * h] - mu) * rs, u1 = (acc[m][0][2 * h + 1] - mu) * rs;
            const float pv = tp < Tp ? 0.5f * (u0 * ginv + u1 * ginv) : 0.f;           // AvgPool1d(2, 2)
            Tm[tp][c] = pv;
            pmax = fmaxf(pmax, fabsf(pv));
        }
    if (amax_out) {                                                    // partial maxima of the pooled tile (gemm_h2.hip's scale)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, o));
        if (lane == 0) amax_out[(size_t)b * 64 + wave] = pmax;
    }
    __syncthreads();
    const int Tpad = (Tp + 31) & ~31;
    float* o = x0 + (size_t)pool_off[b] * 128;
    for (int row = rr; row < Tpad; row += 16)          // (rows past the tile -- Tpad can exceed 16 RG -- are padding: zeros)
        *reinterpret_cast<float4*>(o + (size_t)row * 128 + c4) =
            row < 8 * MT ? *reinterpret_cast<const float4*>(&Tm[row][c4]) : make_float4(0.f, 0.f, 0.f, 0.f);
}

bool mel_front_x3_supported(int T, int K, int lda) { return T >= 2 && T <= 192 && K % 64 == 0 && lda % 4 == 0; }
void launch_mel_front_x3(const float* mag, int lda, const void* melTpk, const int* frame_off, const int* pool_off, float* xm,
                         float* x0, float* stats, float* gstat, int B, int T, int K, hipStream_t st, float* amax_out) {
#define MF(R_) hipLaunchKernelGGL((mel_front_x3_kernel<R_>), dim3(B), dim3(512), 0, st, mag, lda, (const u32x4*)melTpk, frame_off,  \
                                  pool_off, xm, x0, stats, gstat, K, amax_out)
    const int rg = (T + 31) / 32;
    switch (rg) { case 1: MF(1); break; case 2: MF(2); break; case 3: MF(3); break; case 4: MF(4); break; case 5: MF(5); break;
                  default: MF(6); break; }
#undef MF
}

// The backward counterpart of mel_front_x3_kernel, same batches: the data gradient of the first conv block (K = its output
// channels, N = 128 mel channels; the clip's pooled rows as one tile) with the backward of AvgPool, GlobalStandardize and
// InstanceNorm1d(128) in the epilogue.  xm holds the raw mel tile on entry and dL/d(mel) on exit (as
// mel_norm_clip_bwd_kernel leaves it); stats / gstat as the forward kernel wrote them.  Replaces the plain data-gradient GEMM
// + mel_norm_clip_bwd_kernel and the round trip of the pooled gradient between them.
//   reference: what loss.backward() derives for multibit_detector_net.py:126-131
template <int RG>
__global__ __launch_bounds__(512, 2) void mel_back_x3_kernel(const float* __restrict__ dZ, int lda,
                                                                            const u32x4* __restrict__ Bpk,
                                                                            const int* __restrict__ frame_off,
                                                                            const int* __restrict__ pool_off,
                                                                            float* __restrict__ xm, const float* __restrict__ stats,
                                                                            const float* __restrict__ gstat, int K) {
    constexpr int MT = 2 * RG;
    constexpr int FRAG = 1024;
    constexpr int BUF = 2 * 3 * MT * FRAG;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
    __shared__ float red8[2][8];
    const int b = blockIdx.x;
    const int f0 = frame_off[b], T = frame_off[b + 1] - f0, Tp = T / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int Tpad = (Tp + 31) & ~31;
    f32x4 acc[MT][1];
    // (an odd T has one more frame than pooled pairs: the tile may be taller than the clip's pooled rows; reads are clamped)
    x3_tile_gemm<RG, 8>(dZ, lda, Bpk, K, pool_off[b], 0, lds, acc, Tpad);
    const int c = wave * 16 + r16;
    const float* stp = stats + ((size_t)b * 128 + c) * 4;
    const float mu = stp[0], rs = stp[1], M2 = stp[2];
    const float ginv = gstat[b * 4 + 0], gs = gstat[b * 4 + 1], n = gstat[b * 4 + 2];
    float* x = xm + (size_t)f0 * 128 + c;
    float ua[MT][4], ub[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {                   // loads first, from clamped rows; masked below
            const int t = 2 * (m * 16 + 4 * kg + e);
            ua[m][e] = x[(size_t)min(t, T - 1) * 128];
            ub[m][e] = x[(size_t)min(t + 1, T - 1) * 128];
        }
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool pair = m * 16 + 4 * kg + e < Tp;
            const float dv = pair ? 0.5f * acc[m][0][e] : 0.f;        // d(avg pool): half the pooled gradient to each frame
            acc[m][0][e] = dv;
            ua[m][e] = (ua[m][e] - mu) * rs;                          // u
            ub[m][e] = (ub[m][e] - mu) * rs;
            if (pair) { a1 += 2.f * dv; a2 += dv * ua[m][e] + dv * ub[m][e]; }
        }
    a1 += __shfl_xor(a1, 16); a1 += __shfl_xor(a1, 32);
    a2 += __shfl_xor(a2, 16); a2 += __shfl_xor(a2, 32);
    const float D1 = a1, D2 = a2;
    const float w1 = wave_sum(kg == 0 ? D1 : 0.f), w2 = wave_sum(kg == 0 ? D2 : 0.f);
    if (lane == 0) { red8[0][wave] = w1; red8[1][wave] = w2; }
    __syncthreads();
    const float sa = ((red8[0][0] + red8[0][1]) + (red8[0][2] + red8[0][3])) + ((red8[0][4] + red8[0][5]) + (red8[0][6] + red8[0][7]));
    const float sb = ((red8[1][0] + red8[1][1]) + (red8[1][2] + red8[1][3])) + ((red8[1][4] + red8[1][5]) + (red8[1][6] + red8[1][7]));
    const float fT = (float)T;
    const float mdv = sa / n;
    const float Q = (gs > 0.f) ? sb * ginv * ginv / ((n - 1.f) * gs) : 0.f;
    const float m1 = ginv * (D1 / fT - mdv);
    const float m2 = (ginv * D2 - Q * rs * rs * M2) / fT;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t = 2 * (m * 16 + 4 * kg + e);
            const float dv = acc[m][0][e];
            if (t < T) { const float u = ua[m][e]; x[(size_t)t * 128] = rs * (((dv - mdv) * ginv - u * Q) - m1 - u * m2); }
            if (t + 1 < T) { const float u = ub[m][e]; x[(size_t)(t + 1) * 128] = rs * (((dv - mdv) * ginv - u * Q) - m1 - u * m2); }
        }
}

void launch_mel_back_x3(const float* dZ, int lda, const void* wTpk, const int* frame_off, const int* pool_off, float* xm,
                        const float* stats, const float* gstat, int B, int T, int K, hipStream_t st) {
#define MB(R_) hipLaunchKernelGGL((mel_back_x3_kernel<R_>), dim3(B), dim3(512), 0, st, dZ, lda, (const u32x4*)wTpk, frame_off,     \
                                  pool_off, xm, stats, gstat, K)
    const int rg = ((T + 1) / 2 + 31) / 32;            // rows cover the unpaired last frame of an odd T
    switch (rg) { case 1: MB(1); break; case 2: MB(2); break; default: MB(3); break; }
#undef MB
}

bool gemm_clip_x3_supported(int nwm, int N, int K, int lda) {
    return nwm >= 1 && nwm <= 4 && N % 128 == 0 && K % 64 == 0 && lda % 4 == 0;
}


// ---------------------------------------------------------------------------------------------------
// Read-out block of the embed loop in ONE kernel (uniform batches): last Conv1dBlock (C <= 64 channels)
// -> InstanceNorm -> LeakyReLU -> BRH head -> loss -> backward of all of it -> data gradient of the last
// conv -> backward of the previous block's InstanceNorm + LeakyReLU.  Replaces three launches (split-K
// GEMM, tail kernel, clip-aligned data-gradient GEMM) whose work is tiny and latency-bound.
//   reference: detection/multibit_detector_net.py:58-70,133-140 (last block + BRH), modules/BRH.py:16-27,
//              embedding/losses.py, embedding/multibit_embedder.py:109-122 (loss, best tracking)
// Two launches (round 2; one kernel with G workgroups per clip redid the head G times -- 38 of its 82 us at B = 256):
//   readout_head_x3_kernel  one workgroup per clip: sums the clip's split-K partials, InstanceNorm, LeakyReLU, BRH, loss,
//                           bookkeeping and the backward of all of it; leaves dL/dZ of the last block as bf16x3 A fragments
//                           (k = channel, zero-padded to 64) in a scratch image of IMG bytes per clip;
//   readout_grad_x3_kernel  G = Cin/128 workgroups per clip, one 16-column tile per wave, no LDS, no barrier: the data
//                           gradient of the last conv from the image (fragments straight from L2) with the backward of
//                           the previous block's InstanceNorm + LeakyReLU in its epilogue.
// ---------------------------------------------------------------------------------------------------
template <int RG, int NC>
__global__ __launch_bounds__(512) void readout_head_x3_kernel(const float* __restrict__ zpart, int nslab, size_t slab_stride,
                                                               const float* __restrict__ bias, const float* __restrict__ target,
                                                               float* __restrict__ pred, float* __restrict__ loss_out,
                                                               float* __restrict__ best_loss, int* __restrict__ improved,
                                                               int* __restrict__ step, unsigned char* __restrict__ img_out,
                                                               int Tp, int C, int nbits, int loss_kind,
                                                               const float* __restrict__ loss_add) {
    constexpr int MT = 2 * RG;
    constexpr int FRAG = 1024;
    constexpr int KSC = 2;                           // K32 steps of the data-gradient GEMM (C <= 64)
    constexpr int IMG = KSC * 3 * MT * FRAG;         // dZ_last as A fragments of the data-gradient GEMM
    __shared__ __attribute__((aligned(16))) unsigned char img[IMG];
    __shared__ float red[6][8][64];
    __shared__ float mean_s[64], dm[64];

    const int clip = blockIdx.x, g = 0;
    const int bm = clip * 32 * RG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const float invT = 1.0f / (float)Tp;

    // ---- phase 1: z = sum of the split-K partials written by the previous block's kernel (X3_FWD_LAST);
    //      wave m (< MT) holds the 16 rows of row tile m ----
    f32x4 z[NC];
#pragma unroll
    for (int n = 0; n < NC; ++n) z[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wave < MT) {
        const float* zp = zpart + (size_t)(bm + 16 * wave + 4 * kg) * C;
        for (int sl0 = 0; sl0 < nslab; sl0 += 4) {                  // four slabs' loads in flight per round
            float t[4][NC][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sl = sl0 + j < nslab ? sl0 + j : nslab - 1;
#pragma unroll
                for (int n = 0; n < NC; ++n) {
                    const int c = 16 * n + r16 < C ? 16 * n + r16 : C - 1;   // clamped, not branched: loads stay batched
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[j][n][e] = zp[(size_t)sl * slab_stride + (size_t)e * C + c];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (sl0 + j < nslab) {
#pragma unroll
                    for (int n = 0; n < NC; ++n)
#pragma unroll
                        for (int e = 0; e < 4; ++e) z[n][e] += t[j][n][e];
                }
        }
#pragma unroll
        for (int n = 0; n < NC; ++n)
            if (16 * n + r16 >= C) z[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // column sums over the clip's rows: in-lane -> across the four row groups of the wave -> across waves (LDS)
    auto colsum = [&](const float (&v)[NC], int stage, float (&tot)[NC]) {
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            float x = v[n];
            x += __shfl_xor(x, 16);
            x += __shfl_xor(x, 32);
            if (kg == 0) red[stage][wave][16 * n + r16] = x;
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < MT; ++w) t += red[stage][w][16 * n + r16];
            tot[n] = t;
        }
    };
    const bool wv = wave < MT;
    float v[NC], tot[NC], mu[NC], rs[NC];
    // ---- phase 2: InstanceNorm, LeakyReLU, BRH, loss and their backward ----
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        const int c = 16 * n + r16;
        const float bv = (c < C && bias) ? bias[c] : 0.f;
        float sacc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            z[n][e] += bv;
            if (wv && 16 * wave + 4 * kg + e < Tp) sacc += z[n][e];
        }
        v[n] = sacc;
    }
    colsum(v, 0, tot);
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        mu[n] = tot[n] * invT;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (wv && 16 * wave + 4 * kg + e < Tp) { const float d = z[n][e] - mu[n]; q += d * d; }
        v[n] = q;
    }
    colsum(v, 1, tot);
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        rs[n] = 1.0f / sqrtf(tot[n] * invT + 1e-5f);
        float am = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = (z[n][e] - mu[n]) * rs[n];
            z[n][e] = u;                                   // z now holds the normalised pre-activation
            if (wv && 16 * wave + 4 * kg + e < Tp) am += (u > 0.f ? u : 0.2f * u);
        }
        v[n] = am;
    }
    colsum(v, 2, tot);
    if (wave == 0 && kg == 0) {
#pragma unroll
        for (int n = 0; n < NC; ++n) mean_s[16 * n + r16] = (16 * n + r16 < C) ? tot[n] * invT : 0.f;
    }
    __syncthreads();
    if (wave == 0) {
        const int c = lane;
        float lterm = 0.f, dp = 0.f, p = 0.f;
        if (c < nbits) {
            p = tanhf(mean_s[2 * c] - mean_s[2 * c + 1]);
            const float tg = target[clip * nbits + c];
            const float inv = 1.0f / (float)nbits;
            loss_term(loss_kind, p, tg, inv, lterm, dp);
            const float dpre = dp * (1.f - p * p);      // tanh'
            dm[2 * c] = dpre; dm[2 * c + 1] = -dpre;
            if (g == 0) pred[clip * nbits + c] = p;
        }
        float L = wave_sum(lterm);
        if (loss_add) L += loss_add[clip];                       // per-clip term computed elsewhere (L1 on the coefficients)
        if (g == 0 && lane == 0) {
            loss_out[clip] = L;
            if (best_loss) {                                     // null: gradient-only call, no bookkeeping
                const float bl = best_loss[clip];
                const int imp = L < bl;
                improved[clip] = imp;
                if (imp) best_loss[clip] = L;
            }
            if (step && clip == 0) *step += 1;
        }
    }
    // zero the fragment image (columns C..63 and padding rows stay zero)
    for (int i = tid; i < IMG / 16; i += 512) reinterpret_cast<uint4*>(img)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    float ga[NC], v2[NC], tot2[NC];
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        const int c = 16 * n + r16;
        ga[n] = (c < C) ? dm[c] * invT : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (wv && 16 * wave + 4 * kg + e < Tp) { const float du = ga[n] * (z[n][e] > 0.f ? 1.f : 0.2f); s1 += du; s2 += du * z[n][e]; }
        v[n] = s1; v2[n] = s2;
    }
    {   // both sums through one barrier
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            float x = v[n], y = v2[n];
            x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
            y += __shfl_xor(y, 16); y += __shfl_xor(y, 32);
            if (kg == 0) { red[3][wave][16 * n + r16] = x; red[4][wave][16 * n + r16] = y; }
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < MT; ++w) { t1 += red[3][w][16 * n + r16]; t2 += red[4][w][16 * n + r16]; }
            tot[n] = t1; tot2[n] = t2;
        }
    }
    if (wv) {
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            const int c = 16 * n + r16;
            const float m1 = tot[n] * invT, m2 = tot2[n] * invT;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 16 * wave + 4 * kg + e;
                if (row < Tp && c < C) {
                    const float du = ga[n] * (z[n][e] > 0.f ? 1.f : 0.2f);
                    const float dz = rs[n] * (du - m1 - z[n][e] * m2);
                    unsigned p0, p1, p2;
                    split_pair(dz, 0.f, p0, p1, p2);
                    // A-fragment order of the data-gradient GEMM: k = c
                    unsigned char* d = img + (size_t)((c >> 5) * 3 * MT + wave) * FRAG + ((row & 15) + 16 * ((c & 31) >> 3)) * 16 + (c & 7) * 2;
                    *reinterpret_cast<unsigned short*>(d) = (unsigned short)p0;
                    *reinterpret_cast<unsigned short*>(d + MT * FRAG) = (unsigned short)p1;
                    *reinterpret_cast<unsigned short*>(d + 2 * MT * FRAG) = (unsigned short)p2;
                }
            }
        }
    }
    __syncthreads();
    // the finished image goes to the clip's slot of the scratch buffer, 16 bytes per thread and round
    uint4* dst = reinterpret_cast<uint4*>(img_out + (size_t)clip * IMG);
    for (int i = tid; i < IMG / 16; i += 512) dst[i] = reinterpret_cast<const uint4*>(img)[i];
}

template <int RG>
__global__ __launch_bounds__(512, RG <= 3 ? 4 : 2) void readout_grad_x3_kernel(const float* __restrict__ hin, int ci,
                                                                                const unsigned char* __restrict__ img_in,
                                                                                const u32x4* __restrict__ WTpk,
                                                                                const float* __restrict__ rstd_prev,
                                                                                float* __restrict__ dZ, int Tp, int G, int ntiles,
                                                                                float* __restrict__ amax_out) {
    constexpr int MT = 2 * RG;
    constexpr int FRAG = 1024;
    constexpr int KSC = 2;
    constexpr int IMG = KSC * 3 * MT * FRAG;
    int id = blockIdx.x;
    if ((ntiles & 7) == 0) id = (id & 7) * (ntiles >> 3) + (id >> 3);
    const int clip = id / G, g = id % G;
    const int bm = clip * 32 * RG;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    const float invT = 1.0f / (float)Tp;
    const unsigned char* img = img_in + (size_t)clip * IMG;
    const int ntb = g * 8 + wave;                       // the 16-column tile of this wave
    u32x4 bw[KSC][3];
#pragma unroll
    for (int t = 0; t < KSC; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) bw[t][p] = WTpk[(((size_t)ntb * KSC + t) * 3 + p) * 64 + lane];
    // Global memory is touched in a ROW-MAJOR layout, not in the accumulator's: lane = 4 consecutive columns (16 bytes), half
    // a wave = one 512-byte row segment, rows rr + 16 j.  In the accumulator layout a wave instruction covers 4 rows x 64 bytes
    // and this kernel -- which only streams one activation in and one gradient out -- ran at 3.0 TB/s; the same bytes in
    // this layout move at 6 TB/s (measured, timing-only ablation).  The MFMA result changes layout through LDS.
    // (the fragment image and the transposed tile share LDS: every wave reads all of the clip's A fragments, 36 KB, and
    //  re-reading them per wave from L2 through the texture path cost more than the streaming itself)
    constexpr int TBYTES = 32 * RG * 132 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_buf[TBYTES > IMG ? TBYTES : IMG];
    __shared__ float red[2][8][128];
    float (*T)[132] = reinterpret_cast<float (*)[132]>(lds_buf);
    for (int i = tid; i < IMG / 16; i += 512) reinterpret_cast<uint4*>(lds_buf)[i] = reinterpret_cast<const uint4*>(img)[i];
    const int c4 = (lane & 31) * 4, rr = 2 * wave + (lane >> 5);
    const size_t gcol = (size_t)g * 128 + c4;
    float4 hv[MT];
#pragma unroll
    for (int j2 = 0; j2 < MT; ++j2)         // unconditional (padding rows exist and hold zeros)
        hv[j2] = *reinterpret_cast<const float4*>(hin + (size_t)(bm + rr + 16 * j2) * ci + gcol);
    const float4 rsp = *reinterpret_cast<const float4*>(rstd_prev + (size_t)clip * ci + gcol);
    __syncthreads();

    // ---- dL/dh = dZ_last * W, 16 columns per wave, fused backward of the previous block ----
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KSC; ++t) {
        bf16x8 b[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, bw[t][p]);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {                // row tiles by halves: RG fragments x 3 planes in registers at a time
            bf16x8 a[RG][3];
#pragma unroll
            for (int m = 0; m < RG; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    a[m][p] = *reinterpret_cast<const bf16x8*>(lds_buf + (size_t)((t * 3 + p) * MT + hf * RG + m) * FRAG + lane * 16);
#pragma unroll
            for (int term = 0; term < 6; ++term) {
                const int pa = term == 0 ? 2 : (term == 1 || term == 3) ? 1 : 0;
                const int pb = term == 2 ? 2 : (term == 1 || term == 4) ? 1 : 0;
#pragma unroll
                for (int m = 0; m < RG; ++m)
                    acc[hf * RG + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][pa], b[pb], acc[hf * RG + m], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                   // every wave has read its fragments: the buffer becomes the tile
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) T[16 * m + 4 * kg + e][16 * wave + r16] = acc[m][e];
    __syncthreads();
    float4 du[MT];
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
    for (int j2 = 0; j2 < MT; ++j2) {
        const int row = rr + 16 * j2;
        const float4 gacc = *reinterpret_cast<const float4*>(&T[row][c4]);
        const bool valid = row < Tp;
        auto one = [&](float gv, float& av, float& d, float& a1, float& a2) {
            const float uv = valid ? (av > 0.f ? av : av * 5.0f) : 0.f;                 // invert LeakyReLU(0.2)
            d = valid ? gv * (av > 0.f ? 1.f : 0.2f) : 0.f;
            av = uv;
            a1 += d;
            a2 += d * uv;
        };
        one(gacc.x, hv[j2].x, du[j2].x, s1.x, s2.x);
        one(gacc.y, hv[j2].y, du[j2].y, s1.y, s2.y);
        one(gacc.z, hv[j2].z, du[j2].z, s1.z, s2.z);
        one(gacc.w, hv[j2].w, du[j2].w, s1.w, s2.w);
    }
    // column sums: the two row halves of the wave, then the eight waves
    s1.x += __shfl_xor(s1.x, 32); s1.y += __shfl_xor(s1.y, 32); s1.z += __shfl_xor(s1.z, 32); s1.w += __shfl_xor(s1.w, 32);
    s2.x += __shfl_xor(s2.x, 32); s2.y += __shfl_xor(s2.y, 32); s2.z += __shfl_xor(s2.z, 32); s2.w += __shfl_xor(s2.w, 32);
    if (lane < 32) {
        *reinterpret_cast<float4*>(&red[0][wave][c4]) = s1;
        *reinterpret_cast<float4*>(&red[1][wave][c4]) = s2;
    }
    __syncthreads();
    float4 m1 = make_float4(0.f, 0.f, 0.f, 0.f), m2 = m1;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const float4 p1 = *reinterpret_cast<const float4*>(&red[0][w][c4]), p2 = *reinterpret_cast<const float4*>(&red[1][w][c4]);
        m1.x += p1.x; m1.y += p1.y; m1.z += p1.z; m1.w += p1.w;
        m2.x += p2.x; m2.y += p2.y; m2.z += p2.z; m2.w += p2.w;
    }
    m1.x *= invT; m1.y *= invT; m1.z *= invT; m1.w *= invT;
    m2.x *= invT; m2.y *= invT; m2.z *= invT; m2.w *= invT;
    float omax = 0.f;
#pragma unroll
    for (int j2 = 0; j2 < MT; ++j2) {
        const int row = rr + 16 * j2;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < Tp) {
            o.x = rsp.x * (du[j2].x - m1.x - hv[j2].x * m2.x);
            o.y = rsp.y * (du[j2].y - m1.y - hv[j2].y * m2.y);
            o.z = rsp.z * (du[j2].z - m1.z - hv[j2].z * m2.z);
            o.w = rsp.w * (du[j2].w - m1.w - hv[j2].w * m2.w);
        }
        omax = fmaxf(fmaxf(omax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        *reinterpret_cast<float4*>(dZ + (size_t)(bm + row) * ci + gcol) = o;
    }
    if (amax_out) {                                     // partial maxima of the gradient tile (gemm_h2.hip's scale)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
        if (lane == 0) amax_out[(size_t)clip * 64 + g * 8 + wave] = omax;
    }
}

// The same gradient kernel for RAGGED batches (clips of any length): dZl is dL/dZ of the last block as float32 rows with a
// pitch of 64 (tail_kernel, zero K padding); a workgroup owns 128 columns of one clip and walks the clip's pooled rows in
// chunks of 96.  A clip longer than one chunk is swept twice -- column sums of the InstanceNorm backward first, results
// second -- recomputing the (K = 64) products instead of writing raw values and reading them back: the kernel only streams
// the activation in (twice) and the gradient out (once), row-major.  Replaces the K = 64 launch of the ragged conv kernel
// (raw tile + second pass in the accumulator layout: 230 us on config 5).
__global__ __launch_bounds__(512, 4) void readout_grad_ragged_x3_kernel(const float* __restrict__ hin, int ci,
                                                                         const float* __restrict__ dZl,
                                                                         const u32x4* __restrict__ WTpk,
                                                                         const float* __restrict__ rstd_prev, float* __restrict__ dZ,
                                                                         const int* __restrict__ frame_off,
                                                                         const int* __restrict__ pool_off,
                                                                         const int* __restrict__ order, int G,
                                                                         float* __restrict__ amax_out) {
    constexpr int RG = 3, MT = 2 * RG, FRAG = 1024, KSC = 2;
    constexpr int IMG = KSC * 3 * MT * FRAG;
    constexpr int TBYTES = 32 * RG * 132 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_buf[TBYTES > IMG ? TBYTES : IMG];
    __shared__ float red[2][8][128];
    float (*T)[132] = reinterpret_cast<float (*)[132]>(lds_buf);
    const int id = blockIdx.x;
    const int clip = order ? order[id / G] : id / G, g = id % G;
    const int Tp = (frame_off[clip + 1] - frame_off[clip]) / 2;
    if (Tp < 1) return;
    const int row0 = pool_off[clip], npad = (Tp + 31) & ~31;
    const int nchunk = (npad + 32 * RG - 1) / (32 * RG);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    const float invT = 1.0f / (float)Tp;
    const int ntb = g * 8 + wave;                       // the 16-column tile of this wave
    u32x4 bw[KSC][3];
#pragma unroll
    for (int t = 0; t < KSC; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) bw[t][p] = WTpk[(((size_t)ntb * KSC + t) * 3 + p) * 64 + lane];
    const int c4 = (lane & 31) * 4, rr = 2 * wave + (lane >> 5);
    const size_t gcol = (size_t)g * 128 + c4;
    const float4 rsp = *reinterpret_cast<const float4*>(rstd_prev + (size_t)clip * ci + gcol);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, m1 = s1, m2 = s1;
    float pm = 0.f;                                      // max |gradient| of this lane's four columns (gemm_h2.hip's scale)
    // always two sweeps (statistics, then results): keeping a chunk's values in registers across the reduction for the
    // single-chunk case costs the 24 VGPRs that decide between one and two workgroups per CU; a short clip's second read of
    // its 49 KB of activation rows comes from L2
    constexpr int nsweep = 2;
    for (int sweep = 0; sweep < nsweep; ++sweep) {
        for (int ch = 0; ch < nchunk; ++ch) {
            const int rbase = 32 * RG * ch;                     // first row of the chunk inside the clip
            const int rows = min(32 * RG, npad - rbase);        // allocated rows under this chunk (a multiple of 32)
            // dL/dZ of the last block for these rows -> bf16x3 A fragments (k = channel) in LDS
            if (ch || sweep) __syncthreads();                   // the previous round's tile has been read
            for (int i = tid; i < 32 * RG * 16; i += 512) {     // one float4 = 4 consecutive channels of one row
                const int r = i >> 4, q = i & 15, c = 4 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < rows) v = *reinterpret_cast<const float4*>(dZl + (size_t)(row0 + rbase + r) * 64 + c);
                unsigned a0, a1, a2, b0, b1, b2;
                split_pair(v.x, v.y, a0, a1, a2);
                split_pair(v.z, v.w, b0, b1, b2);
                unsigned char* d = lds_buf + (size_t)((c >> 5) * 3 * MT + (r >> 4)) * FRAG + ((r & 15) + 16 * ((c & 31) >> 3)) * 16 + (c & 7) * 2;
                *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
                *reinterpret_cast<uint2*>(d + MT * FRAG) = make_uint2(a1, b1);
                *reinterpret_cast<uint2*>(d + 2 * MT * FRAG) = make_uint2(a2, b2);
            }
            __syncthreads();
            f32x4 acc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < KSC; ++t) {
                bf16x8 b[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, bw[t][p]);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    bf16x8 a[RG][3];
#pragma unroll
                    for (int m = 0; m < RG; ++m)
#pragma unroll
                        for (int p = 0; p < 3; ++p)
                            a[m][p] = *reinterpret_cast<const bf16x8*>(lds_buf + (size_t)((t * 3 + p) * MT + hf * RG + m) * FRAG + lane * 16);
#pragma unroll
                    for (int term = 0; term < 6; ++term) {
                        const int pa = term == 0 ? 2 : (term == 1 || term == 3) ? 1 : 0;
                        const int pb = term == 2 ? 2 : (term == 1 || term == 4) ? 1 : 0;
#pragma unroll
                        for (int m = 0; m < RG; ++m)
                            acc[hf * RG + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m][pa], b[pb], acc[hf * RG + m], 0, 0, 0);
                    }
                }
            }
            // activation rows of the chunk, row-major (requested here, behind the products: before them they cost the
            // registers that keep two workgroups on a CU; the barriers and the transpose below cover most of the latency)
            float4 hv[MT];
#pragma unroll
            for (int j = 0; j < MT; ++j)
                hv[j] = *reinterpret_cast<const float4*>(hin + (size_t)(row0 + rbase + min(rr + 16 * j, rows - 1)) * ci + gcol);
            __syncthreads();                                    // every wave has read its fragments: the buffer becomes the tile
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) T[16 * m + 4 * kg + e][16 * wave + r16] = acc[m][e];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int rl = rr + 16 * j, row = rbase + rl;   // row inside the chunk / the clip
                const float4 gacc = *reinterpret_cast<const float4*>(&T[rl][c4]);
                const bool valid = row < Tp;
                float4 d, u;
                auto one = [&](float gv, float av, float& dd, float& uu) {
                    uu = valid ? (av > 0.f ? av : av * 5.0f) : 0.f;                         // invert LeakyReLU(0.2)
                    dd = valid ? gv * (av > 0.f ? 1.f : 0.2f) : 0.f;
                };
                one(gacc.x, hv[j].x, d.x, u.x);
                one(gacc.y, hv[j].y, d.y, u.y);
                one(gacc.z, hv[j].z, d.z, u.z);
                one(gacc.w, hv[j].w, d.w, u.w);
                if (sweep == 0) {
                    s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
                    s2.x += d.x * u.x; s2.y += d.y * u.y; s2.z += d.z * u.z; s2.w += d.w * u.w;
                } else if (rl < rows) {
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (valid) {
                        o.x = rsp.x * (d.x - m1.x - u.x * m2.x);
                        o.y = rsp.y * (d.y - m1.y - u.y * m2.y);
                        o.z = rsp.z * (d.z - m1.z - u.z * m2.z);
                        o.w = rsp.w * (d.w - m1.w - u.w * m2.w);
                    }
                    pm = fmaxf(fmaxf(pm, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                    *reinterpret_cast<float4*>(dZ + (size_t)(row0 + row) * ci + gcol) = o;
                }
            }
            if (sweep == 0 && ch == nchunk - 1) {
                // column sums: the two row halves of the wave, then the eight waves
                s1.x += __shfl_xor(s1.x, 32); s1.y += __shfl_xor(s1.y, 32); s1.z += __shfl_xor(s1.z, 32); s1.w += __shfl_xor(s1.w, 32);
                s2.x += __shfl_xor(s2.x, 32); s2.y += __shfl_xor(s2.y, 32); s2.z += __shfl_xor(s2.z, 32); s2.w += __shfl_xor(s2.w, 32);
                if (lane < 32) {
                    *reinterpret_cast<float4*>(&red[0][wave][c4]) = s1;
                    *reinterpret_cast<float4*>(&red[1][wave][c4]) = s2;
                }
                __syncthreads();
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    const float4 p1 = *reinterpret_cast<const float4*>(&red[0][w][c4]), p2 = *reinterpret_cast<const float4*>(&red[1][w][c4]);
                    m1.x += p1.x; m1.y += p1.y; m1.z += p1.z; m1.w += p1.w;
                    m2.x += p2.x; m2.y += p2.y; m2.z += p2.z; m2.w += p2.w;
                }
                m1.x *= invT; m1.y *= invT; m1.z *= invT; m1.w *= invT;
                m2.x *= invT; m2.y *= invT; m2.z *= invT; m2.w *= invT;
            }
        }
    }
    if (amax_out) {
        // partial maxima per 16-column group: the group of a lane is (lane & 31) >> 2 (its four lanes in both half-waves),
        // then the eight waves through LDS
        pm = fmaxf(pm, __shfl_xor(pm, 1));
        pm = fmaxf(pm, __shfl_xor(pm, 2));
        pm = fmaxf(pm, __shfl_xor(pm, 32));
        __syncthreads();                                 // `red` is free: every wave has read the column sums
        float* gm = &red[0][0][0];
        if ((lane & 35) == 0) gm[wave * 8 + ((lane & 31) >> 2)] = pm;
        __syncthreads();
        if (tid < 8) {
            float m = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) m = fmaxf(m, gm[w * 8 + tid]);
            amax_out[(size_t)clip * 64 + g * 8 + tid] = m;
        }
    }
}

void launch_readout_grad_ragged_x3(const float* hin, int ci, const float* dZl, const void* WTpk, const float* rstd_prev, float* dZ,
                                   const int* frame_off, const int* pool_off, const int* order, int B, hipStream_t st,
                                   float* amax_out) {
    const int G = ci / 128;
    hipLaunchKernelGGL(readout_grad_ragged_x3_kernel, dim3(B * G), dim3(512), 0, st, hin, ci, dZl, (const u32x4*)WTpk, rstd_prev, dZ,
                       frame_off, pool_off, order, G, amax_out);
}

bool readout_x3_supported(int nwm, int ci, int C) { return nwm >= 1 && nwm <= 4 && ci % 128 == 0 && C >= 2 && C <= 48 && C % 2 == 0; }

// zpart: [nslab][B*32*nwm][C] split-K partials of the last conv (written by launch_gemm_clip_x3 with lastpk/zpart);
// WTpk: x3_pack of the last conv's transposed weights ([ci][C], k zero-padded to 64);
// img: scratch of readout_x3_image_bytes(B, nwm) bytes (dL/dZ of the last block as A fragments)
size_t readout_x3_image_bytes(int B, int nwm) { return (size_t)B * 2 * 3 * (2 * nwm) * 1024; }
void launch_readout_x3(const float* hin, int ci, const float* zpart, int nslab, const float* bias, const void* WTpk,
                       const float* rstd_prev, const float* target, float* pred, float* loss, float* best_loss, int* improved,
                       int* step, float* dZ, int B, int nwm, int Tp, int C, int nbits, int loss_kind, hipStream_t st,
                       const float* loss_add, void* img, float* amax_out) {
    const int G = ci / 128, nc = (C + 15) / 16;
    const size_t slab_stride = (size_t)B * 32 * nwm * C;
#define RH(M_, N_) hipLaunchKernelGGL((readout_head_x3_kernel<M_, N_>), dim3(B), dim3(512), 0, st, zpart, nslab, slab_stride, bias,    \
                                      target, pred, loss, best_loss, improved, step, (unsigned char*)img, Tp, C, nbits, loss_kind, \
                                      loss_add)
#define RN(M_) switch (nc) { case 1: RH(M_, 1); break; case 2: RH(M_, 2); break; case 3: RH(M_, 3); break; default: RH(M_, 4); break; }
    switch (nwm) { case 1: RN(1) break; case 2: RN(2) break; case 3: RN(3) break; default: RN(4) break; }
#undef RN
#undef RH
#define RGK(M_) hipLaunchKernelGGL((readout_grad_x3_kernel<M_>), dim3(B * G), dim3(512), 0, st, hin, ci, (const unsigned char*)img, \
                                   (const u32x4*)WTpk, rstd_prev, dZ, Tp, G, B * G, amax_out)
    switch (nwm) { case 1: RGK(1); break; case 2: RGK(2); break; case 3: RGK(3); break; default: RGK(4); break; }
#undef RGK
}

}  // namespace aware
