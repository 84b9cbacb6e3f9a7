/* libaware_hip -- C ABI of the MI355X (gfx950) AWARE hot path.
 *
 * The drop-in boundary for deepmarkpy/aware's embed -> attack -> detect path.  The
 * reference has no FFI of its own (it is pure Python); the seam it offers is the
 * plugin call  BaseAudioProcessor.__call__(tensor) -> tensor
 * (src/AWARE/interfaces/audio.py:6-9) plus AWAREEmbedder.embed / AWAREDetector.detect
 * (src/AWARE/interfaces/embedding.py:5-8, interfaces/detection.py:6-14) and
 * Attack.apply(audio, sr) (scripts/attacks.py:16-30).  Each entry point below names
 * the reference code it replaces.  The host side (aware_amd/, Python + ctypes) maps
 * these onto the reference's class / function names; see INTEGRATION.md.
 *
 * Conventions
 *   - every pointer marked "dev" is a device (HBM) pointer owned by the caller
 *     (PyTorch-ROCm tensors' data_ptr()); the library never frees caller memory;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued on it, no entry point synchronises unless documented;
 *   - return value: 0 on success, a negative AWARE_E_* code otherwise; nothing throws;
 *   - all arithmetic is IEEE fp32 unless a parameter says f64;
 *   - a "batch" is a ragged set of B mono clips; clip b has n_b samples,
 *     T_b = 1 + n_b/256 frames, Ny_b = 256*(T_b-1) output samples, T_b/2 pooled frames.
 *   - band-limited spectra are frame-major [total frames][256] (first 225 columns are
 *     bins 32..256 = 500..4000 Hz at 16 kHz, n_fft 1024; the tail is zero).
 */
#ifndef AWARE_HIP_H
#define AWARE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AWARE_OK 0
#define AWARE_E_BADARG (-1)
#define AWARE_E_UNSUPPORTED (-2)   /* e.g. n_fft != 1024, hop != 256, band wider than 256 bins */
#define AWARE_E_HIP (-3)           /* a HIP runtime call failed; see aware_last_hip_error() */
#define AWARE_E_WORKSPACE (-4)     /* caller's workspace too small */

/* loss kinds (embedding/losses.py:95-103 and two additions) */
#define AWARE_LOSS_PUSH_EXTREMES 0
#define AWARE_LOSS_MSE 1
#define AWARE_LOSS_HINGE 2
#define AWARE_LOSS_SIGN 3
#define AWARE_LOSS_PUSH_SIGMOID 4
#define AWARE_LOSS_BER 5            /* no gradient, losses.py:90-92 */
#define AWARE_LOSS_PUSH_L1 6        /* EXTENSION (BASELINE config 3 "BER+L1"): push_extremes + l1_weight * mean|c - c0| */
#define AWARE_LOSS_EXTERNAL 7       /* internal: `target` holds dL/dpred (aware_detector_backward) */

#define AWARE_SPEC_STRIDE 256      /* floats per frame row of a band-limited array */
#define AWARE_FULL_STRIDE 520      /* complex values per frame row of a full one-sided spectrum */

typedef struct aware_plan aware_plan;
typedef struct aware_detector aware_detector;
typedef struct aware_batch aware_batch;
typedef struct aware_embed aware_embed;

/* ABI version (200: no process-global knobs, the kernel choices live in aware_embed_config; 300: conv_pipe 0 = f16 two-term
 * kernels, optimiser / scheduler registries, device-side detector training, aware_stft_bwd for any clip length) */
int aware_version(void);
/* text of the last failed HIP runtime call on the calling thread (thread-local) */
const char* aware_last_hip_error(void);
/* The library keeps no mutable process-global state: handles are independent, every entry point may be called
 * from any host thread on any stream (one thread at a time per handle).  The only shared object is a mutex-guarded
 * memo of measured GEMM tile choices, which never changes a result (all tile configurations are bit-identical). */

/* ---- plan: FFT twiddles, window, band ------------------------------------------------
 * Replaces the constructor state of STFT / ISTFT (src/AWARE/utils/audio/stft.py:14-25,
 * :34-45: n_fft, hop_length, window "hann"|"hamming", win_length) and
 * AWAREEmbedder._get_embedding_frequency_indices (embedding/multibit_embedder.py:43-47).
 * window: 0 = hann (periodic), 1 = hamming (periodic).  band_lo_bin/band_hi_bin inclusive. */
int aware_plan_create(aware_plan** out, int n_fft, int hop, int win_length, int window,
                      int band_lo_bin, int band_hi_bin);
void aware_plan_destroy(aware_plan* plan);

/* ---- batch geometry ----------------------------------------------------------------------
 * n_samples[B] (host): clip lengths.  in_offsets[B] (host, may be NULL = densely packed):
 * float offset of clip b inside the caller's ragged audio array. */
int aware_batch_create(aware_batch** out, int B, const int* n_samples, const int* in_offsets);
void aware_batch_destroy(aware_batch* batch);
int aware_batch_total_frames(const aware_batch* batch);   /* sum T_b          */
int aware_batch_total_pooled(const aware_batch* batch);   /* sum T_b/2        */
int aware_batch_total_out(const aware_batch* batch);      /* sum 256*(T_b-1)  */
int aware_batch_out_offset(const aware_batch* batch, int b); /* float offset of clip b's output */
int aware_batch_out_length(const aware_batch* batch, int b);
int aware_batch_frames(const aware_batch* batch, int b);

/* ---- DSP plug-ins ---------------------------------------------------------------------------
 * aware_stft: WaveformNormalizer (optional) + STFT.__call__  (utils/audio/waveform.py:18-19,
 * utils/audio/stft.py:27-28).  audio: dev ragged f32.  spec: dev complex64
 * [total frames][AWARE_FULL_STRIDE] (bins 0..512 valid).  normalize: 0 none, 1 x/max(|x|+1e-8).
 * scratch: dev, >= aware_batch_scratch_bytes(). */
size_t aware_batch_scratch_bytes(const aware_batch* batch);
int aware_stft(const aware_plan* plan, const aware_batch* batch, const float* audio, int normalize,
               void* spec, void* scratch, void* stream);
/* aware_istft: ISTFT.__call__ (utils/audio/stft.py:47-48; no length argument, output
 * 256*(T-1) samples per clip at aware_batch_out_offset).  normalize as above, applied to the output. */
int aware_istft(const aware_plan* plan, const aware_batch* batch, const void* spec, int normalize,
                float* out, void* scratch, void* stream);
/* aware_stft_band: normalise + STFT + STFTDecomposer restricted to the embedding band:
 * mag [total frames][256] and unit phasor (cos, sin of the phase) [total frames][256] complex64. */
int aware_stft_band(const aware_plan* plan, const aware_batch* batch, const float* audio, int normalize,
                    float* mag, void* phasor, void* scratch, void* stream);

/* Backward passes of the two transforms for the differentiable plug-in seam (BaseAudioProcessor.__call__,
 * interfaces/audio.py:6-9; what torch autograd derives for torch.stft / torch.istft inside
 * embedding/multibit_embedder.py:49-67).  Gradients of a real loss; complex gradients in torch's convention
 * (dL/dRe + i dL/dIm).
 * aware_stft_bwd: grad_spec dev complex64 [total frames][AWARE_FULL_STRIDE] -> grad_audio dev f32, laid out like the audio
 *   aware_stft takes (clip b: n_b samples at in_offsets[b]): any clip length n > 512, ragged batches (the reflect pads fold
 *   about sample 0 and sample n - 1).
 * aware_istft_bwd: grad_audio dev f32 [total out] -> grad_spec dev complex64 [total frames][AWARE_FULL_STRIDE]. */
int aware_stft_bwd(const aware_plan* plan, const aware_batch* batch, const void* grad_spec, float* grad_audio,
                   void* stream);
int aware_istft_bwd(const aware_plan* plan, const aware_batch* batch, const float* grad_audio, void* grad_spec,
                    void* stream);

/* The element-wise plug-ins and the optimiser step of the reference's loop, with their backward passes, for the same
 * seam (n = number of complex / real elements; all pointers dev):
 *   aware_polar_decompose      STFTDecomposer: (|S|, angle S)  utils/audio/stft.py:54-55  (phase may be NULL)
 *   aware_polar_assemble       STFTAssembler: mag * exp(i phase)  utils/audio/stft.py:61-62
 *   aware_waveform_normalize_bwd  backward of x / max(|x| + 1e-8), utils/audio/waveform.py:18-19 (through the max)
 *   aware_nadam_clamp_step     torch.optim.NAdam single-tensor step + torch.clamp(coeffs, lo, hi)
 *                              (embedding/multibit_embedder.py:112-117); coef3 = the three per-step scalars written by
 *                              aware_nadam_coefficients (host; mu_product_io carries torch's mu_product between steps,
 *                              start it at 1.0f; step counts from 1).  Same arithmetic as the fused loop. */
int aware_polar_decompose(const void* spec, float* mag, float* phase, size_t n, void* stream);
int aware_polar_decompose_bwd(const void* spec, const float* grad_mag, const float* grad_phase, void* grad_spec, size_t n,
                              void* stream);
int aware_polar_assemble(const float* mag, const float* phase, void* spec, size_t n, void* stream);
int aware_polar_assemble_bwd(const float* mag, const float* phase, const void* grad_spec, float* grad_mag,
                             float* grad_phase, size_t n, void* stream);
int aware_waveform_normalize_bwd(const float* in, const float* grad_out, float* grad_in, const int* off, const int* len,
                                 int B, void* stream);
int aware_nadam_coefficients(int step, float lr, float beta1, float beta2, float momentum_decay, float* mu_product_io,
                             float* coef3);
int aware_nadam_clamp_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* lo,
                           const float* hi, size_t n, const float* coef3, float beta1, float beta2, float eps, void* stream);

/* ---- detector --------------------------------------------------------------------------------
 * AWAREDetectorNet (detection/multibit_detector_net.py:17-80).  All arrays are host fp32:
 * mel_basis [n_mels][n_fft/2+1] (detection/modules/mel.py:105-149), conv weights
 * [channels[i+1]][channels[i]] and biases [channels[i+1]] for i < n_layers
 * (channels = {128, 512, 1024, 1024, 40}).  The library uploads and pre-transposes them. */
int aware_detector_create(aware_detector** out, const aware_plan* plan, const float* mel_basis, int n_mels,
                          int n_layers, const int* channels, const float* const* weights,
                          const float* const* biases);
void aware_detector_destroy(aware_detector* det);

/* AWAREDetector.detect (detection/multibit_detector.py:28-42), batched: normalise, STFT, |.|,
 * zero the out-of-band bins, network forward.  values: dev f32 [B][n_bits].
 * workspace: dev, >= aware_detect_workspace_bytes(). */
size_t aware_detect_workspace_bytes(const aware_batch* batch, const aware_detector* det);
int aware_detect(const aware_plan* plan, const aware_detector* det, const aware_batch* batch,
                 const float* audio, float* values, void* workspace, size_t workspace_bytes, void* stream);
/* network forward only (AWAREDetectorNet.forward, :109-140) on a band magnitude array */
int aware_detector_forward(const aware_detector* det, const aware_batch* batch, const float* mag,
                           float* values, void* workspace, size_t workspace_bytes, void* stream);

/* forward + backward of the network for the differentiable seam (BaseDetectorNet.forward, interfaces/detection.py:10-14,
 * under autograd): values [B][n_bits] (may be NULL) and grad_mag [total frames][256] = J^T grad_values; data gradients
 * only (the reference freezes the weights, multibit_embedder.py:76-77). */
size_t aware_detector_backward_workspace_bytes(const aware_batch* batch, const aware_detector* det);
int aware_detector_backward(const aware_detector* det, const aware_batch* batch, const float* mag,
                            const float* grad_values, float* values, float* grad_mag, void* workspace,
                            size_t workspace_bytes, void* stream);

/* EXTENSION -- detector training step (BASELINE.json north_star: "RCCL all-reduce over xGMI on the embedder/detector
 * gradients"; the reference freezes the detector, multibit_embedder.py:76-77, and trains nothing: parity unpinned, specified
 * by torch autograd on the oracle's Detector).  aware_detector_weight_gradients = aware_detector_backward plus
 * dL/dW_l [Cout][Cin] and dL/db_l [Cout] of every conv block (grad_weights / grad_biases: host arrays of n_layers device
 * pointers, entries may be NULL); the caller all-reduces them over its ranks, applies its optimiser and writes the new
 * parameters back with aware_detector_update (host arrays as for aware_detector_create, same shapes, synchronous). */
size_t aware_detector_train_workspace_bytes(const aware_batch* batch, const aware_detector* det);
int aware_detector_weight_gradients(const aware_detector* det, const aware_batch* batch, const float* mag,
                                    const float* grad_values, float* values, float* grad_mag,
                                    float* const* grad_weights, float* const* grad_biases, void* workspace,
                                    size_t workspace_bytes, void* stream);
int aware_detector_update(aware_detector* det, const float* mel_basis, const float* const* weights,
                          const float* const* biases);
/* the step without host round trips: aware_detector_train_gradients evaluates the loss inside (ONE forward + backward; target
 * [B][n_bits] bipolar, loss_kind 0..5 as the embed loop, loss_out dev [B]; the gradients are those of the SUM of the per-clip
 * losses; grad_mag may be NULL), aware_detector_update_device rebuilds every device image of the parameters from DEVICE arrays
 * (asynchronous on `stream`; dev_weights / dev_biases: host arrays of device pointers, biases may be NULL). */
int aware_detector_train_gradients(const aware_detector* det, const aware_batch* batch, const float* mag, const float* target,
                                   int loss_kind, float* loss_out, float* values, float* grad_mag,
                                   float* const* grad_weights, float* const* grad_biases, void* workspace,
                                   size_t workspace_bytes, void* stream);
int aware_detector_update_device(aware_detector* det, const float* const* dev_weights, const float* const* dev_biases,
                                 void* stream);

/* ---- embedder -----------------------------------------------------------------------------------
 * AWAREEmbedder.embed / _optimize (embedding/multibit_embedder.py:70-197), batched and ragged:
 * every clip is its own optimisation problem.  loss: AWARE_LOSS_* above -- 0 push_extremes, 1 mse, 2 hinge, 3 sign,
 * 4 push_sigmoid, 5 ber (no gradient) (embedding/losses.py:95-103; bce needs sigmoid outputs), 6 push_extremes + L1
 * (EXTENSION, streaming DSP path only).  optimizer: NAdam (embedding/optimizers.py:5; torch.optim.NAdam
 * single-tensor semantics) with lr, beta1, beta2, eps, momentum_decay; the reference's
 * ReduceLROnPlateau(patience 500) never fires within 400 iterations and is not modelled. */
typedef struct aware_embed_config {
    int num_iterations;      /* cards/config.yaml:16  (400) */
    float tolerance_db;      /* cards/config.yaml:13  (6.0) */
    int loss;                /* 0 = push_extremes */
    float lr, beta1, beta2, eps, momentum_decay;   /* 0.1, 0.9, 0.999, 1e-8, 4e-3 */
    int use_graph;           /* 1: capture one iteration into a hipGraph and replay it */
    /* kernel choices of this session (zero = default).
     * conv_pipe 0: the conv blocks and their data-gradient GEMMs of a uniform batch that fills the chip run on the f16 matrix
     *   pipe: every f32 operand scaled by a power of two (per output channel / per clip) and written as two binary16 terms
     *   (representation error <= one f32 ulp, rms 2^-24.5; l_a l_b dropped), three partial products per multiply-add, f32 accumulation
     *   (csrc/gemm_h2.hip); every other GEMM as conv_pipe 2.
     * conv_pipe 2: the detector's GEMMs on the bf16 matrix pipe with every f32 operand split exactly into three
     *   bf16 terms, six partial products per multiply-add, f32 accumulation (csrc/gemm_x3.hip) wherever K % 64 == 0 and
     *   N % 128 == 0, f32 MFMA otherwise.
     * conv_pipe 1: f32-input MFMA everywhere (the pipe the 16-bit kernels are tested against).
     * All three agree with fp64 to f32 rounding level (tests/test_gpu_kernels.py::test_gemm_clip_x3).
     * readout 0: fused read-out kernel on uniform batches; 1: split-K GEMM + tail kernel + data-gradient GEMM (the
     *   path ragged batches take). */
    int conv_pipe;
    int readout;
    /* dsp_path 0: streaming wave kernels for the framed STFT / iSTFT and their adjoints (csrc/dsp_stream.hip: one wave
     *   streams a run of frames, overlap-add and frame overlap in registers, no barrier) wherever the band lies inside
     *   bins 1..256; 1: workgroup-staged kernels (csrc/dsp_kernels.hip: any band; the form the streaming kernels are
     *   tested against). */
    int dsp_path;
    /* loss AWARE_LOSS_PUSH_L1 only (EXTENSION, BASELINE config 3 "BER + L1"; the reference's imperceptibility device is
     * the box constraint :157-160, which stays in force): weight of mean|c - c0| over a clip's 225*T coefficients */
    float l1_weight;
    /* mel 0: the mel projection's backward runs as two taps per bin inside the streaming synthesis adjoint (a triangular filter
     *   bank -- detection/modules/mel.py:105-149 -- has at most two adjacent non-zero weights per FFT bin; dL/d|S| is never
     *   stored) whenever dsp_path is 0 and the detector's basis has that form; 1: the dense [NF][128] x [128][256] GEMM
     *   (what the fused form is tested against). */
    int mel;
} aware_embed_config;

/* ---- optimiser / scheduler registries (the reference's third seam: embedding/optimizers.py:3-20, schedulers.py:3-16) ------
 * The model card's NAdam + never-firing ReduceLROnPlateau runs fused in the adjoint kernel's epilogue (aware_embed_config).
 * Any other registered optimiser / schedule: the HOST computes the per-step scalars of torch's single-tensor update under
 * the chosen learning-rate schedule, the device applies them in one element-wise launch per iteration (+ clamp, + best
 * snapshot).  kind: AWARE_OPT_*.  table: HOST [num_iterations][5] doubles per step t = 1..: (ux, uy, z, lr_t, h0_t) with
 * c.x = lr*ux, c.y = lr*uy (sgd: uy = 1 on the first step), c.z = z, h0_t >= 0 overrides hyp[0] (CyclicLR cycles beta1 /
 * momentum); hyp: see csrc/dsp_args.hpp::opt_clamp_update.  plateau != 0: torch's ReduceLROnPlateau (mode min, rel
 * threshold, cooldown 0) with one state PER CLIP, stepped with the clip's loss after each optimiser step; lr0 = initial rate.
 * Call after aware_embed_create and before the first aware_embed_iterate. */
#define AWARE_OPT_NADAM 0
#define AWARE_OPT_ADAM 1
#define AWARE_OPT_ADAMW 2
#define AWARE_OPT_SGD 3
#define AWARE_OPT_RMSPROP 4
#define AWARE_OPT_ADAGRAD 5
#define AWARE_OPT_ADAMAX 6
#define AWARE_OPT_ADADELTA 7
typedef struct aware_optimizer_config {
    int kind;
    float hyp[8];
    double weight_decay;      /* adamw only (decoupled); the L2 form of the others travels in hyp[4] */
    const double* table;
    int plateau, patience;
    double factor, threshold, min_lr, eps, lr0;
} aware_optimizer_config;
int aware_embed_set_optimizer(aware_embed* e, const aware_optimizer_config* cfg, void* stream);
/* the same update on the caller's flat tensors, for the plug-in loop (generalises aware_nadam_clamp_step): coef4 = the
 * step's (c.x, c.y, c.z, c.w) as float, hyp8 as above; state1 = exp_avg / momentum buffer / acc_delta, state2 = exp_avg_sq /
 * square_avg / state_sum / exp_inf */
int aware_opt_clamp_step(int kind, float* param, const float* grad, float* state1, float* state2, const float* lo,
                         const float* hi, size_t n, const float* coef4, const float* hyp8, void* stream);

size_t aware_embed_workspace_bytes(const aware_batch* batch, const aware_detector* det);
int aware_embed_create(aware_embed** out, const aware_plan* plan, const aware_detector* det,
                       const aware_batch* batch, const aware_embed_config* cfg, void* workspace,
                       size_t workspace_bytes, void* stream);
void aware_embed_destroy(aware_embed* e);
/* analysis, bounds, optimiser reset.  audio: dev ragged f32 (un-normalised); target: dev f32
 * [B][n_bits] bipolar (+-1), PatternEncoder output (utils/watermark/encoder.py:35-45). */
int aware_embed_begin(aware_embed* e, const float* audio, const float* target, void* stream);
/* n_iters loop bodies (:95-122): synth -> normalise -> analysis -> detector fwd -> loss ->
 * backward -> NAdam -> clamp -> best snapshot.  No host synchronisation.  AWARE_E_BADARG when more than
 * cfg.num_iterations steps would have run since aware_embed_begin (the reference's loop runs exactly that many). */
int aware_embed_iterate(aware_embed* e, int n_iters, void* stream);
/* forward + backward without the optimiser step and without best-loss bookkeeping; grad: dev f32
 * [total frames][256] = dL/dcoef; loss[] and pred[] (aware_embed_buffer 0, 2) are refreshed */
int aware_embed_gradient(aware_embed* e, float* grad, void* stream);
/* Timing aid for the roofline report: runs n_iters loop bodies eagerly with a HIP event recorded on
 * `stream` after every kernel launch; writes the elapsed milliseconds between consecutive events and
 * a kernel kind per launch (0 synth, 1 analysis, 2 generic gemm, 3 mel-norm, 4 in+lrelu, 5 read-out/tail,
 * 6 synth adjoint, 7 analysis adjoint + NAdam, 8 misc, 9 clip-aligned gemm with fused forward
 * epilogue, 10 the same with fused backward epilogue).  Synchronises the stream.  Returns the number of
 * entries written or a negative error.  (These iterations DO step the optimiser and count against
 * cfg.num_iterations like aware_embed_iterate.) */
int aware_embed_profile(aware_embed* e, int n_iters, int max_entries, float* ms_out, int* kind_out, void* stream);
/* final synthesis from the best coefficients (:173-194) and the service-level rescale
 * (service/embed.py:69,73): out[b] = rescale[b] * normalise(istft(...)).  rescale: dev f32 [B] or NULL. */
int aware_embed_finish(aware_embed* e, const float* rescale, float* out, void* stream);
/* device pointers to internal state for inspection: 0 loss[B], 1 best_loss[B], 2 pred[B][n_bits],
 * 3 coef [frames][256], 4 best coef, 5 lo, 6 hi, 7 phasor (complex64), 8 step counter (int32), 9 un-normalised synthesis,
 * 10 band magnitudes of the last analysis, 11 per-clip learning rates (f64 [B]; NULL unless aware_embed_set_optimizer ran) */
void* aware_embed_buffer(aware_embed* e, int which);

/* ---- attacks (scripts/attacks.py) ------------------------------------------------------------------
 * All operate on ragged batches given by dev int32 arrays off[B], len[B]. */
/* PCMBitDepthConversion.apply :44-70. bits in {8,12,16,24}.  scratch >= B*(ceil(max_len/4096)*8 + 4) bytes */
int aware_pcm_quantize(const float* in, float* out, const int* off, const int* len, int B, int max_len,
                       int bits, void* scratch, void* stream);
/* WaveformNormalizer.__call__ (src/AWARE/utils/audio/waveform.py:18-19) as a stand-alone op:
 * out = in / max(|in| + 1e-8) per clip.  scratch as for aware_pcm_quantize (+ 4*B bytes). */
int aware_waveform_normalize(const float* in, float* out, const int* off, const int* len, int B, int max_len,
                             void* scratch, void* stream);
/* scipy.signal.resample_poly's polyphase core (Resample.apply :290-293, scripts/test.py:60-63):
 * out[j] = sum_i in[i] * h[j*down - i*up + half_len], fp32.  h: dev f32 [nh] (already scaled by up). */
int aware_upfirdn(const float* in, const int* in_off, const int* in_len, float* out, const int* out_off,
                  const int* out_len, int B, int max_out, const float* h, int nh, int up, int down,
                  int half_len, void* stream);
/* Resample.apply's branch for sr // target_sr = factor > 1 (scripts/attacks.py:275-288): keep every factor-th sample
 * and linearly interpolate back to the original length with np.interp's float64 arithmetic.  out: dev f64 at the
 * same offsets as `in`. */
int aware_decimate_interp(const float* in, const int* off, const int* len, int B, int max_len, int factor,
                          double* out, void* stream);
/* scipy.signal.lfilter (LowPassFilter / HighPassFilter :400-455) and filtfilt (RandomBandstop
 * :324-356) in f64.  b, a: dev f64 [B][ncoef] (a[0] == 1); zi: dev f64 [B][ncoef-1] (filtfilt only).
 * out is f64 when out_f64 != 0 (the reference returns float64 from lfilter), else f32.
 * scratch (filtfilt): dev f64 [B][max_len + 6*ncoef].  2 <= ncoef <= 12.  A filtfilt clip must be longer than the
 * padding of 3*ncoef samples (scipy raises for such a clip; the host binding does the same, the kernel only stays inside
 * its buffers).  One workgroup per clip, parallel in time (128 chunks chained with a double-double state transition):
 * results agree with scipy's sequential recurrence to the rounding of that recurrence itself. */
int aware_iir(const float* in, const int* off, const int* len, int B, int max_len, void* out, int out_f64,
              const double* b, const double* a, const double* zi, int ncoef, int filtfilt,
              void* scratch, void* stream);
/* DeleteSamples :162-178 / Cropout :192-205 (zero_fill = 0) and SampleSupression :370-385
 * (zero_fill = 1): cut_start[B], cut_len[B] dev int32 chosen by the host RNG. */
int aware_segment_cut(const float* in, const int* in_off, float* out, const int* out_off, const int* out_len,
                      const int* cut_start, const int* cut_len, int zero_fill, int B, int max_len, void* stream);
/* EXTENSION (not in the reference): additive Gaussian noise at snr_db, Philox-4x32-10 keyed by
 * seeds[b].  scratch >= B*8 bytes. */
int aware_gaussian_noise(const float* in, float* out, const int* off, const int* len, int B, int max_len,
                         const uint32_t* seeds, float snr_db, void* scratch, void* stream);

/* EXTENSION (not in the reference): MP3-like quantisation surrogate on a full one-sided spectrum
 * [n_frames][AWARE_FULL_STRIDE] complex64, in place: per frame the magnitudes are quantised on a
 * step_db grid relative to the frame maximum and zeroed below floor_db; the phase is kept.
 * Used as  aware_stft -> aware_spectral_quantize -> aware_istft. */
int aware_spectral_quantize(void* spec, int n_frames, float step_db, float floor_db, void* stream);
/* Its backward, which makes the surrogate a DIFFERENTIABLE op (north_star: "differentiable MP3-like quantisation
 * surrogates"; the reference's MP3Compression, scripts/attacks.py:73-148, shells out to ffmpeg and has no gradient):
 * straight-through on the magnitude (dQ/d|X| := 1 on kept bins, 0 on bins dropped below the floor, frame maximum constant),
 * exact through the phase.  spec_in: the spectrum BEFORE quantisation; grad_out = dL/dRe Y + i dL/dIm Y; grad_in likewise for
 * X; all dev complex64 [n_frames][AWARE_FULL_STRIDE].  Specified by oracle/aware_oracle.py::mp3_surrogate_spectrum under torch
 * autograd (parity unpinned: extension). */
int aware_spectral_quantize_bwd(const void* spec_in, const void* grad_out, void* grad_in, int n_frames, float step_db,
                                float floor_db, void* stream);

/* EXTENSION (stand-in for the reference's rubberband-based TimeStretch / PitchShift, scripts/attacks.py:208-252; the
 * binary is absent, parity with it unpinned): phase vocoder on one-sided spectra [frames][AWARE_FULL_STRIDE] complex64.
 * Output frame t of clip c sits at input position t*rate: linear magnitude interpolation, phase accumulated from the
 * wrapped per-bin phase increments (f64).  frame_off_* are device arrays of B+1 frame offsets; the caller sizes
 * the output as ceil(T_c / rate) frames per clip.  Used as aware_stft -> aware_phase_vocoder -> aware_istft. */
int aware_phase_vocoder(const void* spec_in, const int* frame_off_in, void* spec_out, const int* frame_off_out, int B,
                        double rate, void* stream);

/* ---- quality metric ----------------------------------------------------------------------------------
 * SNR.__call__ (src/AWARE/metrics/audio.py:68-89) per clip: 10 log10(mean(output^2) / mean((output - target)^2))
 * over lengths[c] samples (the caller passes the common length, :82-84), +inf when the clips are identical;
 * f64 accumulation in a fixed order.  Offsets are float offsets of clip c in the two signal arrays. */
int aware_snr(const float* output, const int* out_offsets, const float* target, const int* tgt_offsets,
              const int* lengths, int B, double* snr_db, void* stream);

/* ---- bare GEMM (tests / roofline): C[M][N] = A[M][K] * Bt[N][K]^T + bias ------------------------------ */
int aware_gemm_nt(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C, int ldc,
                  int M, int N, int K, void* stream);
/* same with an explicit tile configuration: variant 0 = automatic, 1..16 = fixed (tuning aid; all
 * configurations produce bit-identical results) */
int aware_gemm_nt_variant(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C, int ldc,
                          int M, int N, int K, int variant, void* stream);

/* ---- the clip-aligned conv block alone (tests / roofline) ----------------------------------------------
 * One Conv1dBlock of the detector (detection/modules/conv1d.py:38-42) on a uniform batch: clip b owns rows
 * [b*32*ceil(Tp/32), +Tp) of A and C (the rest of each 32-row group is padding, written as zero).
 *   epi 0: C = A*Bt^T + bias          epi 1: C = LeakyReLU_0.2(InstanceNorm_t(A*Bt^T + bias)), rstd_io[b][n] written
 *   epi 2: A is dL/d(output of the previous block), act that output; C = dL/d(its pre-norm conv output)
 * mode 0 runs the f32-MFMA kernel on Bt [N][K]; mode 1 the bf16 matrix-pipe kernel on Bpk, the same matrix split
 * exactly into three bf16 planes and re-ordered by aware_x3_pack (host buffers; aware_x3_packed_bytes bytes);
 * six partial products per multiply-add, f32 accumulation: f32-equivalent results (N % 128 == 0, K % 64 == 0). */
size_t aware_x3_packed_bytes(int N, int K);
int aware_x3_pack(const float* host_wt, int N, int K, void* host_out);
int aware_gemm_clip(const float* A, int lda, const float* Bt, int ldb, const void* Bpk, const float* bias, float* C,
                    int ldc, int B, int Tp, int N, int K, int epi, float* rstd_io, const float* act, int mode,
                    void* stream);
/* The same forward block (epi 1, bf16 matrix-pipe kernel) as the embed loop runs it for the block in front of the skinny last
 * conv (detection/multibit_detector_net.py:58-70: 1024 -> 40): besides C and rstd_out the epilogue writes the split-K
 * partials of the NEXT conv, zpart [N/128][B*32*ceil(Tp/32)][CL] with  sum_s zpart[s] = C * Wlast^T  (no bias).
 * lastpk: dev, aware_x3_pack of Wlast [16*ceil(CL/16)][N] (rows beyond CL zero).  2 <= CL <= 48.  Test / roofline entry. */
int aware_gemm_clip_last(const float* A, int lda, const void* Bpk, const float* bias, float* C, int ldc, int B, int Tp,
                         int N, int K, float* rstd_out, const void* lastpk, float* zpart, int CL, void* stream);
/* The same block on the DEFAULT conv pipe of the embed loop (csrc/gemm_h2.hip): the f16 matrix pipe with every f32 operand
 * written as two binary16 terms after a power-of-two scaling (per output channel for the weights, per clip for A), three
 * partial products per multiply-add, f32 accumulation; representation error <= 2^-23 relative per operand (one f32 ulp; rms 2^-24.5), the l_a l_b term (<= 2^-22) dropped.
 * Bt: DEV [N][K] (row pitch ldb); the entry packs it and computes the clips' max |A| into `workspace`
 * (>= aware_gemm_clip_h2_workspace_bytes).  epi 0..2 as aware_gemm_clip; lastpk / zpart / CL: as aware_gemm_clip_last (epi 1,
 * may be NULL / 0); amax_out: dev [B][64] partial maxima of |C| per clip (N/16 written per clip) or NULL.
 * N % 128 == 0, K % 64 == 0, K <= 1024. */
size_t aware_gemm_clip_h2_workspace_bytes(int B, int N, int K);
int aware_gemm_clip_h2(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C, int ldc, int B, int Tp,
                       int N, int K, int epi, float* rstd_io, const float* act, const void* lastpk, float* zpart, int CL,
                       float* amax_out, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AWARE_HIP_H */
