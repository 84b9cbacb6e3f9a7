// Internal launch interface between the C ABI (capi.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "common.hpp"

namespace aware {

// ---- dsp_kernels.hip ------------------------------------------------------------------
constexpr int kStreamWaves = 4;        // waves (= runs) per workgroup of the streaming DSP kernels

struct AnalysisLaunch {
    PlanDev plan;
    const int* frame_off = nullptr;
    int B = 0, max_frames = 0;
    // streaming kernels: frames per run (0: chosen from B and max_frames) and the flat workgroup table of the batch
    // (entry = clip << 12 | workgroup within the clip; null: a (workgroups of the longest clip) x B grid)
    int run_frames = 0, n_wg = 0;
    const int* wg_tab = nullptr;
    const float* sig = nullptr;
    const int* sig_off = nullptr;
    const int* sig_len = nullptr;
    const unsigned long long* pmax = nullptr;
    const int* pcount = nullptr;
    int pstride = 0;
    int double_norm = 0;
    float unit_default = 0.f;
    float* mag = nullptr;
    void* unit = nullptr;
    void* full = nullptr;
    // adjoint mode
    int adjoint = 0;
    const float* yraw = nullptr;
    const double* pdot = nullptr;
    const void* phasor = nullptr;
    float* coef = nullptr; float* mom = nullptr; float* vel = nullptr;
    const float* lo = nullptr; const float* hi = nullptr; float* best = nullptr;
    const int* improved = nullptr;
    const void* sched = nullptr;
    int sched_len = 0;
    const int* step = nullptr;
    float* grad_out = nullptr;
    int do_step = 0;
    float hyp[4] = {0.1f, 0.999f, 0.001f, 1e-8f};
    // stream = 1: barrier-free streaming wave kernels (dsp_stream.hip; band inside bins 1..256 only)
    int stream = 0;
    const float* gpad = nullptr;      // stream + adjoint: reflect-pad parts written by the streaming synthesis adjoint
    int write_pad = 1;                // stream, forward: write the zero tail of the mag / unit rows
    const float* c0 = nullptr;        // stream + adjoint: original coefficients (the box is recomputed from them)
    float box_ratio = 0.f;            // 10^(-tolerance_db / 20)
    float l1_weight = 0.f;            // != 0: L1 term on the coefficients (loss push_extremes + L1)
    float* mel_out = nullptr;         // stream, forward: the mel tile [NF][128] instead of mag (AnalysisArgs)
    const float* melf_w = nullptr;
    const unsigned char* melf_s = nullptr;
};
struct SynthLaunch {
    PlanDev plan;
    const int* frame_off = nullptr;
    int B = 0, max_frames = 0;
    int n_wg = 0;
    const int* wg_tab = nullptr;        // as in AnalysisLaunch, for runs of run_blocks hop blocks
    const float* amp = nullptr;
    const void* ph = nullptr;
    const void* full = nullptr;
    float* out = nullptr;
    const float* add = nullptr;
    unsigned long long* pmax = nullptr;
    int pstride = 0;
    int adjoint = 0;
    const float* yraw = nullptr;
    const unsigned long long* pmax_in = nullptr;
    const int* pcount = nullptr;
    double* pdot = nullptr;
    int stream = 0;
    float* gpad = nullptr;            // stream + adjoint: [B][2][512] reflect-pad parts out
    int run_blocks = kSynthBlocks;    // hop blocks per run (aware_batch::synth_run); the partial counts follow it
    const float* c0 = nullptr;        // stream, forward: per-run sums of |amp - c0| into pl1 (L1 term)
    double* pl1 = nullptr;
    const int* sig_off = nullptr;     // staged adjoint on a full spectrum: general-length output (see SynthArgs)
    const int* sig_len = nullptr;
    const float* dmel = nullptr;      // stream + adjoint: amplitudes from dL/d(mel) [NF][128] through the two-tap table (SynthArgs)
    const void* melw = nullptr;
    const unsigned char* melm = nullptr;
};
void launch_absmax_partials(const float* sig, const int* sig_off, const int* sig_len, unsigned long long* pmax,
                            int pstride, int B, int max_len, hipStream_t st);
void launch_analysis(const AnalysisLaunch& L, hipStream_t st);
void launch_synth(const SynthLaunch& L, hipStream_t st);
// dsp_stream.hip: true when the streaming wave kernels serve this plan (band inside bins 1..256)
bool stream_supported(const PlanDev& plan);
void launch_analysis_stream(const AnalysisLaunch& L, hipStream_t st);
void launch_synth_stream(const SynthLaunch& L, hipStream_t st);
// l1term[b] = weight * (sum of the clip's pl1 partials) / (nband * T_b): the L1 part of the loss push_extremes + L1
void launch_l1_reduce(const double* pl1, const int* pcount, int pstride, const int* frame_off, int nband, float weight,
                      float* l1term, int B, hipStream_t st);
void launch_embed_prepare(const float* c0, float* coef, float* lo, float* hi, float* mom, float* vel, float* best,
                          float* c0_keep, float ratio, size_t n, hipStream_t st);
void launch_oob_residual(const float* audio, const int* in_off, const unsigned long long* pmax, const int* pcount,
                         int pstride, const float* band, const int* frame_off, float* oob, int B, int max_frames,
                         hipStream_t st);
void launch_finish(const float* yraw, const int* frame_off, const unsigned long long* pmax, const int* pcount,
                   int pstride, const float* rescale, float* out, const int* out_off, int B, int max_frames,
                   hipStream_t st);

// ---- detector_kernels.hip ---------------------------------------------------------------
// C[M][N] = A[M][K] * Bt[N][K]^T (+ bias[N]); all row-major fp32, K % 4 == 0, 16-byte aligned rows
void launch_gemm_nt(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C, int ldc, int M,
                    int N, int K, hipStream_t st);
void launch_gemm_nt_variant(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C, int ldc,
                            int M, int N, int K, int variant, hipStream_t st);
int gemm_autotune(const float* A, int lda, const float* Bt, int ldb, float* C, int ldc, int M, int N, int K,
                  hipStream_t st);
// clip-aligned GEMM with fused InstanceNorm+LeakyReLU epilogues (uniform batches, <= 128 pooled rows per clip)
void launch_gemm_clip(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C, int ldc, int B,
                      int nwm, int Tp, int N, int K, int epi, float* rstd_io, const float* act, hipStream_t st);
// ---- gemm_x3.hip: the same clip-aligned GEMM on the bf16 matrix pipe, f32-equivalent (3-way operand split) ----
size_t x3_packed_bytes(int N, int K);
void x3_pack(const float* Wt, int N, int K, uint16_t* out);      // host: [N][K] f32 -> fragment-ordered bf16 planes
void launch_x3_pack_dev(const float* Wt_dev, int ldw, int Nvalid, int Kvalid, int N, int K, void* out, hipStream_t st);   // device -> device
bool gemm_clip_x3_supported(int nwm, int N, int K, int lda);
// mel projection + InstanceNorm + GlobalStandardize + AvgPool of a UNIFORM batch of clips of T <= 192 frames in one launch
// (one workgroup per clip): xm [NF][128] raw mel tile (kept for the backward), x0 pooled tile, stats / gstat as
// launch_mel_norm_fwd leaves them
bool mel_front_x3_supported(int T, int K, int lda);
void launch_mel_front_x3(const float* mag, int lda, const void* melTpk, const int* frame_off, const int* pool_off, float* xm,
                         float* x0, float* stats, float* gstat, int B, int T, int K, hipStream_t st, float* amax_out = nullptr);
// (amax_out: [B][64], 8 partial maxima of |x0| per clip for gemm_h2.hip, or null)
// backward of the same block for the same batches: data gradient of the first conv block (dZ [NP][K], wTpk = x3_pack of its
// transposed weights [128][K]) + AvgPool / GlobalStandardize / InstanceNorm backward; xm: raw mel in, dL/d(mel) out
void launch_mel_back_x3(const float* dZ, int lda, const void* wTpk, const int* frame_off, const int* pool_off, float* xm,
                        const float* stats, const float* gstat, int B, int T, int K, hipStream_t st);
// lastpk/zpart (forward epilogue only): also emit the split-K partials [N/128][B*32*nwm][CL] of the next, last conv
// block (x3_pack of its weights zero-padded to a multiple of 16 rows), consumed by launch_readout_x3
void launch_gemm_clip_x3(const float* A, int lda, const void* Bpk, const float* bias, float* C, int ldc, int B, int nwm,
                         int Tp, int N, int K, int epi, float* rstd_io, const float* act, hipStream_t st,
                         const void* lastpk = nullptr, float* zpart = nullptr, int CL = 0, int Mrows = 0);
// ---- gemm_h2.hip: the same block on the f16 matrix pipe, two-term operand split, three products (f32-level) ----
size_t h2_packed_bytes(int N, int K);
void launch_h2_pack(const float* Wt_dev, int ldw, int N, int K, void* out, hipStream_t st);   // device -> device
void launch_clip_amax(const float* A, int lda, int K, int rows_per_clip, int B, float* amax, hipStream_t st);
bool gemm_clip_h2_supported(int nwm, int N, int K, int lda);
int gemm_clip_h2_slab_width(int nwm, int N, int B);     // columns per workgroup (FWD_LAST: N / width partial slabs in zpart)
// amax_in: [B][64] partial maxima of |A| per clip (K/16 valid); amax_out: [B][64] the same of C (N/16 written) or null
void launch_gemm_clip_h2(const float* A, int lda, const void* Bpk, const float* amax_in, float* amax_out, const float* bias,
                         float* C, int ldc, int B, int nwm, int Tp, int N, int K, int epi, float* rstd_io, const float* act,
                         hipStream_t st, const void* lastpk = nullptr, float* zpart = nullptr, int CL = 0);
void launch_gemm_ragged_h2(const float* A, int lda, const void* Bpk, const float* amax_in, float* amax_out, const float* bias,
                           float* C, int ldc, int B, const int* frame_off, const int* pool_off, const int* order, int N, int K,
                           int epi, float* rstd_io, const float* act, hipStream_t st);
void launch_ragged_amax(const float* A, int lda, int K, const int* frame_off, const int* pool_off, int B, float* amax, hipStream_t st);
// (Mrows > 0, plain epilogue only: the matrices have Mrows < B*32*nwm rows -- the last row block is partial)
// the same block for ragged batches / clips of any length (one launch; clips longer than 96 pooled frames in two passes)
void launch_gemm_ragged_x3(const float* A, int lda, const void* Bpk, const float* bias, float* C, int ldc, int B,
                           const int* frame_off, const int* pool_off, const int* order, int N, int K, int epi, float* rstd_io,
                           const float* act, hipStream_t st);
// (order: [B] clip indices, longest first, or null = as given; speed only)
// fused read-out of the embed loop: last conv block + BRH + loss + their backward + data gradient of the last conv
// + backward of the previous block's norm/activation (uniform batches; see gemm_x3.hip)
bool readout_x3_supported(int nwm, int ci, int C);
void launch_readout_x3(const float* hin, int ci, const float* zpart, int nslab, const float* bias, const void* WTpk,
                       const float* rstd_prev, const float* target, float* pred, float* loss, float* best_loss, int* improved,
                       int* step, float* dZ, int B, int nwm, int Tp, int C, int nbits, int loss_kind, hipStream_t st,
                       const float* loss_add, void* img, float* amax_out = nullptr);
// (amax_out: [B][64], ci/16 partial maxima of |dZ| per clip for gemm_h2.hip, or null)
size_t readout_x3_image_bytes(int B, int nwm);      // scratch `img` of launch_readout_x3 (must not alias dZ)
// ragged batches: data gradient of the last conv from dZl (float32 rows, pitch 64, zero K padding: launch_tail with ldz = 64)
// with the backward of the previous block's InstanceNorm + LeakyReLU; ci % 128 == 0, last conv of at most 64 channels
void launch_readout_grad_ragged_x3(const float* hin, int ci, const float* dZl, const void* WTpk, const float* rstd_prev, float* dZ,
                                   const int* frame_off, const int* pool_off, const int* order, int B, hipStream_t st,
                                   float* amax_out = nullptr);
// mel block: InstanceNorm over time, per-clip GlobalStandardize, AvgPool(2,2)
bool launch_mel_norm_fwd(const float* xm, const int* frame_off, const int* pool_off, float* x0, float* stats,
                         float* gstat, float* part, int pstride, int B, int max_frames, hipStream_t st,
                         float* amax_out = nullptr);   // (amax_out as launch_mel_front_x3; returns whether it was written)
void launch_mel_norm_bwd(const float* dx0, float* xm_inout, const int* frame_off, const int* pool_off, const float* stats,
                         const float* gstat, float* part, int pstride, int B, int max_frames, hipStream_t st);
// conv block tail: InstanceNorm over time + LeakyReLU(0.2), in place; saves rstd
void launch_in_lrelu_fwd(float* z, const int* frame_off, const int* pool_off, float* rstd, int C, int B, int max_pooled,
                         hipStream_t st);
// backward of the same, in place on dA (A is the post-activation output of the forward)
void launch_in_lrelu_bwd(float* dA, const float* A, const int* frame_off, const int* pool_off, const float* rstd, int C, int B,
                         int max_pooled, hipStream_t st);
// BRH + loss + dL/dA3; also best-loss tracking
void launch_head(const float* a3, const int* frame_off, const int* pool_off, const float* target, float* pred, float* loss,
                 float* best_loss, int* improved, float* dA3, int* step, int loss_kind, int nbits, int B,
                 hipStream_t st, const float* loss_add = nullptr);
void launch_gemm_nt_splitk(const float* A, int lda, const float* Bt, int ldb, float* Cpart, int ldc, int M, int N, int K,
                           int ksplit, hipStream_t st);
void launch_tail(const float* zpart, int nsplit, size_t slab, const float* bias, const int* frame_off, const int* pool_off,
                 const float* target, float* pred, float* loss, float* best_loss, int* improved, float* dZ, int* step,
                 int loss_kind, int nbits, int B, int max_pooled, hipStream_t st, const float* loss_add = nullptr, int ldz = 0);
// (ldz: row pitch of dZ; 0 = 2*nbits, 64 = zero-padded to the K of the bf16x3 data-gradient GEMM)

// ---- seam_kernels.hip: element-wise pieces of the differentiable plug-in seam -----------------------------
void launch_polar_decompose(const void* spec, float* mag, float* phase, size_t n, hipStream_t st);
void launch_polar_decompose_bwd(const void* spec, const float* gmag, const float* gphase, void* gspec, size_t n, hipStream_t st);
void launch_polar_assemble(const float* mag, const float* phase, void* spec, size_t n, hipStream_t st);
void launch_polar_assemble_bwd(const float* mag, const float* phase, const void* gspec, float* gmag, float* gphase, size_t n,
                               hipStream_t st);
void launch_transpose(const float* in, float* out, int R, int C, hipStream_t st);
void launch_colsum(const float* in, float* out, int R, int C, hipStream_t st);
void launch_normalize_bwd(const float* x, const float* g, float* dx, const int* off, const int* len, int B, hipStream_t st);
void launch_nadam_clamp(float* p, const float* g, float* m, float* v, const float* lo, const float* hi, size_t n, float c_grad,
                        float c_mom, float bias_corr2, float beta1, float beta2, float eps, hipStream_t st);

void launch_opt_clamp(int kind, float* p, const float* g, float* m, float* v, const float* lo, const float* hi, size_t n,
                      const float* c4, const float* h8, hipStream_t st);
void launch_opt_rows(int kind, float* coef, const float* grad, float* mom, float* vel, const float* c0, float ratio, float* best,
                     const int* improved, const int* frame_off, int B, int NF, const double* tab, int tab_len, const int* step,
                     const double* lr_clip, double wd, const float* h8, int nband, hipStream_t st);
void launch_plateau(const float* loss, double* state, double* lr_clip, int B, double factor, int patience, double threshold,
                    double min_lr, double eps, hipStream_t st);

// ---- attack_kernels.hip -----------------------------------------------------------------
void launch_pcm_quantize(const float* in, float* out, const int* off, const int* len, const unsigned long long* pmax,
                         const int* pcount, int pstride, float q, float lo, float hi, int B, int max_len,
                         hipStream_t st);
void launch_normalize(const float* in, float* out, const int* off, const int* len, const unsigned long long* pmax,
                      const int* pcount, int pstride, int B, int max_len, hipStream_t st);
void launch_upfirdn(const float* in, const int* in_off, const int* in_len, float* out, const int* out_off,
                    const int* out_len, const float* h, int nh, int up, int down, int half_len, int B, int max_out,
                    hipStream_t st);
void launch_iir_full(const float* in, const int* off, const int* len, void* out, int out_f64, const double* b,
                     const double* a, const double* zi, int ncoef, int mode, double* scratch, int sstride, int B,
                     hipStream_t st);
void launch_gaussian_noise_full(const float* in, float* out, const int* off, const int* len, const unsigned* seeds,
                                double* power, float snr_db, int B, int max_len, hipStream_t st);
void launch_snr(const float* a, const int* a_off, const float* b, const int* b_off, const int* len, double* out, int B,
                hipStream_t st);
void launch_phase_vocoder(const void* in, const int* fin, void* out, const int* fout, double rate, int B, hipStream_t st);
void launch_spectral_quantize(void* spec, int nframes, float step_db, float floor_db, hipStream_t st);
void launch_spectral_quantize_bwd(const void* spec, const void* gout, void* gin, int nframes, float step_db, float floor_db,
                                  hipStream_t st);
void launch_decimate_interp(const float* in, const int* off, const int* len, double* out, int k, int B, int max_len,
                            hipStream_t st);
void launch_segment_copy(const float* in, const int* in_off, float* out, const int* out_off, const int* out_len,
                         const int* cut_start, const int* cut_len, int zero_fill, int B, int max_len, hipStream_t st);

}  // namespace aware
