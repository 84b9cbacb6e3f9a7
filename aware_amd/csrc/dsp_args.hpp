// Argument blocks and small device helpers shared by the workgroup-staged DSP kernels (dsp_kernels.hip) and the
// streaming wave kernels (dsp_stream.hip).
#pragma once
#include "common.hpp"

namespace aware {

// 1-ulp hardware reciprocal / square root (v_rcp_f32, v_sqrt_f32) instead of the ~10-instruction IEEE expansions
// hipcc emits for `/` and sqrtf: a quarter of the analysis kernels' vector instructions were division fix-ups.
// Where the reference divides by a per-clip scalar (waveform.py:18-19) the scalar's reciprocal is still an IEEE
// division, taken once per thread; the per-sample operation becomes a multiplication (<= 1 ulp from the quotient).
#ifndef AWARE_EXACT_DIV
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
__device__ __forceinline__ float fast_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ float fast_sqrt(float x) { return sqrtf(x); }
#endif

// torch.optim.NAdam single-tensor step (torch/optim/nadam.py _single_tensor_nadam) + clamp to the tolerance box
// (embedding/multibit_embedder.py:112-117).  c_grad = -lr (1 - mu_t) / (1 - mu_product), c_mom = -lr mu_{t+1} /
// (1 - mu_product mu_{t+1}), inv_bc2 = 1 / (1 - beta2^t); hyp = {1 - beta1, beta2, 1 - beta2, eps}.
__device__ __forceinline__ void nadam_clamp_update(float& p, float& mo, float& ve, float g, float blo, float bhi,
                                                   float c_grad, float c_mom, float inv_bc2, const float4& hyp) {
    mo = mo + hyp.x * (g - mo);                    // exp_avg.lerp_(grad, 1-beta1)
    ve = ve * hyp.y + (hyp.z * g) * g;            // mul_(beta2).addcmul_(g, g, 1-beta2)
    const float rden = fast_rcp(fast_sqrt(ve * inv_bc2) + hyp.w);   // 1 / (sqrt(v / bias_corr2) + eps)
    p = p + (c_grad * g) * rden;
    p = p + (c_mom * mo) * rden;
    p = fminf(fmaxf(p, blo), bhi);
}

// The other optimisers of the reference's registry (embedding/optimizers.py:3-20), as one element-wise update with the
// arithmetic of torch's single-tensor implementations (torch/optim/{adam,sgd,rmsprop,adagrad,adamax,adadelta}.py -- what the
// reference's get_optimizer(name, [coeffs], **params).step() runs on its CPU tensors), followed by the clamp to the box.
// c: per-step scalars computed on the host (learning-rate schedule included), h: hyper-parameters:
//   kind 1 adam / 2 adamw  c.x = -lr/bc1, c.z = sqrt(bc2), c.w = 1 - lr*wd (adamw);  h0 = 1-beta1, h1 = beta2, h2 = 1-beta2, h3 = eps
//   kind 3 sgd             c.x = -lr, c.y = 1 on the first step (buf = grad);  h0 = momentum, h5 = nesterov, h6 = 1-dampening
//   kind 4 rmsprop         c.x = -lr;  h1 = alpha, h2 = 1-alpha, h3 = eps
//   kind 5 adagrad         c.x = -lr/(1+(t-1)*lr_decay);  h3 = eps
//   kind 6 adamax          c.x = -lr/bc1;  h0 = 1-beta1, h1 = beta2, h3 = eps
//   kind 7 adadelta        c.x = -lr;  h1 = rho, h2 = 1-rho, h3 = eps     (mo holds acc_delta, ve square_avg)
//   kind 0 nadam           c.x, c.y, c.z as nadam_clamp_update (c.z = bc2);  h0..h3 as adam
//   all but adamw: h4 = weight_decay (L2: grad += wd * param);  all: h7 = gradient scale (0 = none)
enum { OPT_NADAM = 0, OPT_ADAM = 1, OPT_ADAMW = 2, OPT_SGD = 3, OPT_RMSPROP = 4, OPT_ADAGRAD = 5, OPT_ADAMAX = 6, OPT_ADADELTA = 7 };
struct OptHyp { float h[8]; };
__device__ __forceinline__ void opt_clamp_update(int kind, float& p, float& mo, float& ve, float g, float blo, float bhi,
                                                 const float4& c, const OptHyp& H) {
    const float* h = H.h;
    if (h[7] != 0.f) g = g * h[7];                 // gradient scale (e.g. 1 / clips of a summed batch gradient); 0 = none
    if (kind == OPT_ADAMW) p = p * c.w;
    else if (h[4] != 0.f) g = g + h[4] * p;
    if (kind == OPT_NADAM) {
        mo = mo + h[0] * (g - mo);
        ve = ve * h[1] + (h[2] * g) * g;
        const float den = sqrtf(ve / c.z) + h[3];
        p = p + (c.x * g) / den;
        p = p + (c.y * mo) / den;
    } else if (kind == OPT_ADAM || kind == OPT_ADAMW) {
        mo = mo + h[0] * (g - mo);
        ve = ve * h[1] + (h[2] * g) * g;
        const float den = sqrtf(ve) / c.z + h[3];
        p = p + (c.x * mo) / den;
    } else if (kind == OPT_SGD) {
        float d = g;
        if (h[0] != 0.f) {
            mo = (c.y != 0.f) ? g : mo * h[0] + h[6] * g;
            d = (h[5] != 0.f) ? g + h[0] * mo : mo;
        }
        p = p + c.x * d;
    } else if (kind == OPT_RMSPROP) {
        ve = ve * h[1] + (h[2] * g) * g;
        const float avg = sqrtf(ve) + h[3];
        p = p + (c.x * g) / avg;
    } else if (kind == OPT_ADAGRAD) {
        ve = ve + g * g;
        const float sd = sqrtf(ve) + h[3];
        p = p + (c.x * g) / sd;
    } else if (kind == OPT_ADAMAX) {
        mo = mo + h[0] * (g - mo);
        ve = fmaxf(ve * h[1], fabsf(g) + h[3]);
        p = p + (c.x * mo) / ve;
    } else {      // OPT_ADADELTA
        ve = ve * h[1] + (h[2] * g) * g;
        const float sd = sqrtf(ve + h[3]);
        const float delta = sqrtf(mo + h[3]) / sd * g;
        mo = mo * h[1] + (h[2] * delta) * delta;
        p = p + c.x * delta;
    }
    p = fminf(fmaxf(p, blo), bhi);
}

// The tolerance box of a coefficient (embedding/multibit_embedder.py:157-160): d = c0 * 10^(-tol/20),
// lo = max(c0 - d, 0), hi = c0 + d.  Explicitly rounded operations (no fused multiply-add), so that the kernel that
// stores the box and the kernel that recomputes it from c0 agree bit for bit.
__device__ __forceinline__ void box_bounds(float c0, float ratio, float& lo, float& hi) {
    const float d = __fmul_rn(c0, ratio);
    lo = fmaxf(__fsub_rn(c0, d), 0.f);
    hi = __fadd_rn(c0, d);
}

enum { AN_NORM = 0, AN_ADJ = 1 };
enum { SY_FWD = 0, SY_ADJ = 1 };

struct AnalysisArgs {
    PlanDev plan;
    const int* frame_off;             // [B+1]
    const int* wg_tab;                // streaming kernels: [gridDim.x] clip << 12 | workgroup within the clip (null: 2-D grid)
    const float* sig;                 // signal base
    const int* sig_off;               // [B] float offset of clip b in `sig`
    const int* sig_len;               // [B] samples (reflect padding uses this length)
    const unsigned long long* pmax;   // [B][pstride] partial |y| maxima (or null: no normalisation)
    const int* pcount;                // [B] number of partials
    int pstride;
    int double_norm;                  // AN_NORM: 1 = y/m/m2, 0 = y/m
    float unit_default;               // phasor written where |X| == 0 (x component)
    float* mag;                       // [NF][kFS] or null
    cf* unit;                         // [NF][kFS] or null
    cf* full;                         // [NF][520] full complex spectrum (k = 0..512) or null
    // AN_ADJ only (adjoint of synthesis + fused optimiser step)
    const float* yraw;                // un-normalised synthesis output (same offsets as sig)
    const double* pdot;               // [B][pstride] partial sums of g*y2
    const cf* phasor;                 // [NF][kFS] unit phasor of the original phase
    float* coef;                      // [NF][kFS] variables
    float* mom;                       // exp_avg
    float* vel;                       // exp_avg_sq
    const float* lo;
    const float* hi;
    float* best;
    const int* improved;              // [B]
    const float4* sched;              // per step {c_grad, c_mom, bias_correction2, 0}
    int sched_len;                    // entries in `sched` (the step index is clamped to it)
    const int* step;                  // device step counter
    float* grad_out;                  // optional [NF][kFS] raw gradient (tests)
    int do_step;                      // 0: only write grad_out
    float4 hyp;                       // {1-beta1, beta2, 1-beta2, eps}
    // streaming wave kernels only (dsp_stream.hip)
    const float* gpad;                // AN_ADJ: [B][2][512] reflect-pad parts of the synthesis adjoint, folded in on load
    int write_pad;                    // AN_NORM: also write the zero tail (columns nband..255) of mag / unit rows
    const float* c0;                  // AN_ADJ (streaming): original coefficients [NF][kFS]; the clamp's box is recomputed from
    float box_ratio;                  //   them (one operand instead of lo and hi): box_bounds(c0, box_ratio)
    float l1_weight;                  // loss push_extremes + L1 (EXTENSION): dL/dc += l1_weight * sign(c - c0) / (nband * T)
    // streaming AN_NORM with the mel projection folded in (mag is not written): mel_out[row][m] = sum_j melf_w[m][j] *
    // |X|[melf_s[m] + j], j < kMelTapsA (m < 64) / kMelTapsB (m >= 64): a triangular filter is a short run of adjacent bins
    float* mel_out;                   // [NF][128]
    const float* melf_w;              // [128][kMelTapsB], zero beyond the filter's support
    const unsigned char* melf_s;      // [128] first band column of the support (<= kFS - kMelTapsB)
};



struct SynthArgs {
    PlanDev plan;
    const int* frame_off;
    const int* wg_tab;                // as in AnalysisArgs
    const float* amp;                 // [NF][kFS] real amplitude (coefficients or dL/dmag)
    const cf* ph;                     // [NF][kFS] unit phasor
    const cf* full;                   // [NF][520] full complex spectrum (SY_FWD only) or null
    float* out;                       // per-clip signals at offset 256*(frame_off[b]-b)
    const float* add;                 // SY_FWD: constant out-of-band part added to the output (or null)
    unsigned long long* pmax;         // SY_FWD: [B][pstride] partial maxima out
    int pstride;
    // SY_ADJ
    const float* yraw;                // forward synthesis output
    const unsigned long long* pmax_in;
    const int* pcount;
    double* pdot;                     // [B][pstride] partial sums of g2*y2 out
    float* gpad;                      // streaming SY_ADJ: [B][2][512] reflect-pad parts out (left pads, right pads)
    int run_blocks;                   // hop blocks per run / workgroup segment (<= kSynthBlocks)
    const float* c0;                  // streaming SY_FWD, L1 term: original coefficients; per-run sums of |amp - c0| go to
    double* pl1;                      //   pl1[B][pstride] (null: no L1 term)
    // staged SY_ADJ on a full spectrum, general form (backward of aware_stft for clips of ANY length n > 512): clip b's
    // gradient goes to out + sig_off[b], sig_len[b] samples, the right reflect pad mirrors about sample n - 1
    const int* sig_off;
    const int* sig_len;
    // streaming SY_ADJ with the mel projection's backward folded in: amp is not read; the amplitude of band column f is
    //   melw[f].x * dmel[row][melm[f]] + melw[f].y * dmel[row][melm[f] + 1]
    // (dmel [NF][128] = dL/d(mel); a triangular mel filter bank has at most two adjacent non-zero weights per bin)
    const float* dmel;
    const float2* melw;               // [kFS]
    const unsigned char* melm;        // [kFS], <= 126
};


}  // namespace aware
