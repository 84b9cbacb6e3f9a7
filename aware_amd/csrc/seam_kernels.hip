// Element-wise kernels of the differentiable plug-in seam (gfx950): what the reference's plug-in objects and its
// optimiser do one torch op at a time between the transforms --
//   STFTDecomposer  abs / angle                 utils/audio/stft.py:54-55
//   STFTAssembler   mag * exp(i phase)          utils/audio/stft.py:61-62
//   WaveformNormalizer backward                 utils/audio/waveform.py:18-19 (gradient flows through the max)
//   NAdam step + clamp to the tolerance box     embedding/multibit_embedder.py:112-117 (torch.optim.NAdam)
// with their backward passes, so that the reference-shaped loop (plug-in lists + autograd, aware_amd/utils/audio/
// plugins.py) runs on the same arithmetic as the fused hot loop.  All HBM-bound streaming kernels.
#include "common.hpp"
#include "dsp_args.hpp"
#include "kernels.h"

namespace aware {

__global__ __launch_bounds__(256) void polar_decompose_kernel(const cf* __restrict__ spec, float* __restrict__ mag,
                                                              float* __restrict__ phase, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const cf s = spec[i];
        mag[i] = hypotf(s.x, s.y);
        if (phase) phase[i] = atan2f(s.y, s.x);
    }
}
// d|S| = Re(conj(S) dS)/|S| (0 at S = 0, torch's sgn convention); d angle = Im(conj(S) dS)/|S|^2
__global__ __launch_bounds__(256) void polar_decompose_bwd_kernel(const cf* __restrict__ spec, const float* __restrict__ gmag,
                                                                  const float* __restrict__ gphase, cf* __restrict__ gspec, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const cf s = spec[i];
        const float r = hypotf(s.x, s.y);
        cf g = mk(0.f, 0.f);
        if (r > 0.f) {
            const float ir = 1.0f / r;
            if (gmag) { const float gm = gmag[i]; g.x += gm * s.x * ir; g.y += gm * s.y * ir; }
            if (gphase) { const float gp = gphase[i] * ir * ir; g.x += -gp * s.y; g.y += gp * s.x; }
        }
        gspec[i] = g;
    }
}
__global__ __launch_bounds__(256) void polar_assemble_kernel(const float* __restrict__ mag, const float* __restrict__ phase,
                                                             cf* __restrict__ spec, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float sn, cs;
        sincosf(phase[i], &sn, &cs);
        const float m = mag[i];
        spec[i] = mk(m * cs, m * sn);
    }
}
__global__ __launch_bounds__(256) void polar_assemble_bwd_kernel(const float* __restrict__ mag, const float* __restrict__ phase,
                                                                 const cf* __restrict__ gspec, float* __restrict__ gmag,
                                                                 float* __restrict__ gphase, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float sn, cs;
        sincosf(phase[i], &sn, &cs);
        const cf g = gspec[i];
        if (gmag) gmag[i] = g.x * cs + g.y * sn;
        if (gphase) gphase[i] = mag[i] * (g.y * cs - g.x * sn);
    }
}

// y = x / m, m = max|x| + 1e-8:  dx_j = g_j / m - [j == k] sign(x_k) (sum_i g_i x_i) / m^2, k = first arg max |x|.
// One workgroup per clip: max pass, dot pass (f64, fixed order), apply pass.
__global__ __launch_bounds__(256) void normalize_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            float* __restrict__ dx, const int* __restrict__ off,
                                                            const int* __restrict__ len) {
    __shared__ unsigned long long red[4];
    __shared__ double dred[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = len[b];
    const float* xb = x + off[b];
    const float* gb = g + off[b];
    float* db = dx + off[b];
    unsigned long long v = 0;
    double s = 0.0;
    for (int i = tid; i < n; i += 256) {
        const float xv = xb[i];
        v = umax64(v, pack_max(fabsf(xv), (unsigned)i));
        s += (double)gb[i] * (double)xv;
    }
    v = wave_max64(v);
    s = wave_sum_d(s);
    if ((tid & 63) == 0) { red[tid >> 6] = v; dred[tid >> 6] = s; }
    __syncthreads();
    v = umax64(umax64(red[0], red[1]), umax64(red[2], red[3]));
    const double dot = (dred[0] + dred[1]) + (dred[2] + dred[3]);
    const float m = __uint_as_float((unsigned)(v >> 32)) + 1e-8f;
    const unsigned k = 0xFFFFFFFFu - (unsigned)(v & 0xFFFFFFFFu);
    const float xk = n > 0 ? xb[min(k, (unsigned)(n - 1))] : 0.f;
    const float sk = (xk > 0.f) ? 1.f : ((xk < 0.f) ? -1.f : 0.f);
    const float corr = sk * (float)(dot / ((double)m * (double)m));
    for (int i = tid; i < n; i += 256) {
        float d = gb[i] / m;
        if ((unsigned)i == k) d -= corr;
        db[i] = d;
    }
}

// torch.optim.NAdam single-tensor step followed by the clamp to [lo, hi]; the same arithmetic as the fused epilogue
// of the analysis adjoint (nadam_clamp_update in dsp_args.hpp)
__global__ __launch_bounds__(256) void nadam_clamp_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          const float* __restrict__ lo, const float* __restrict__ hi, size_t n,
                                                          float c_grad, float c_mom, float inv_bc2, float4 hyp) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float mo = m[i], ve = v[i], pv = p[i];
        nadam_clamp_update(pv, mo, ve, g[i], lo ? lo[i] : -INFINITY, hi ? hi[i] : INFINITY, c_grad, c_mom, inv_bc2, hyp);
        m[i] = mo; v[i] = ve; p[i] = pv;
    }
}

// Any optimiser of opt_clamp_update (dsp_args.hpp) + clamp, flat arrays, per-step scalars from the host: the generic form of
// nadam_clamp_kernel for the plug-in seam.
__global__ __launch_bounds__(256) void opt_clamp_kernel(int kind, float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        const float* __restrict__ lo, const float* __restrict__ hi, size_t n,
                                                        float4 c, OptHyp H) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float mo = m[i], ve = v[i], pv = p[i];
        opt_clamp_update(kind, pv, mo, ve, g[i], lo ? lo[i] : -INFINITY, hi ? hi[i] : INFINITY, c, H);
        m[i] = mo; v[i] = ve; p[i] = pv;
    }
}
void launch_opt_clamp(int kind, float* p, const float* g, float* m, float* v, const float* lo, const float* hi, size_t n,
                      const float* c4, const float* h8, hipStream_t st) {
    OptHyp H;
    for (int i = 0; i < 8; ++i) H.h[i] = h8[i];
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(opt_clamp_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, st, kind, p, g, m, v, lo, hi,
                       n, make_float4(c4[0], c4[1], c4[2], c4[3]), H);
}

// The optimiser step of an embed session that does not run in the fused epilogue of the analysis adjoint (any optimiser but
// NAdam, or a learning rate that differs per clip): one workgroup per spectral row [256 columns] of the batch.
//   tab: [n_steps][5] doubles per step t: (ux, uy, z, lr_t, h0_t) -- h0_t replaces h[0] (1 - beta1 or the momentum: CyclicLR cycles
//        it) when >= 0; c.x = (float)(lr * ux), c.y = (float)(lr * uy) for the kinds whose
//        c.x / c.y are proportional to the learning rate (all; sgd's first-step flag travels in uy with ux-only scaling),
//        c.z = (float)z, c.w = (float)(1 - lr * wd) for adamw; lr = lr_clip[clip] when lr_clip != null (ReduceLROnPlateau
//        state, per clip) else lr_t;
//   the box comes from c0 and `ratio` (box_bounds), the best snapshot follows improved[clip] (multibit_embedder.py:116-122).
__global__ __launch_bounds__(256) void opt_rows_kernel(int kind, float* __restrict__ coef, const float* __restrict__ grad,
                                                       float* __restrict__ mom, float* __restrict__ vel,
                                                       const float* __restrict__ c0, float ratio, float* __restrict__ best,
                                                       const int* __restrict__ improved, const int* __restrict__ frame_off, int B,
                                                       const double* __restrict__ tab, int tab_len, const int* __restrict__ step,
                                                       const double* __restrict__ lr_clip, double wd, OptHyp H, int nband) {
    const int row = blockIdx.x, f = threadIdx.x;
    int lo_ = 0, hi_ = B;                               // clip of this row: frame_off[clip] <= row < frame_off[clip + 1]
    while (hi_ - lo_ > 1) {
        const int mid = (lo_ + hi_) >> 1;
        if (frame_off[mid] <= row) lo_ = mid; else hi_ = mid;
    }
    const int clip = lo_;
    if (f >= nband) return;
    const int t = min(max(*step - 1, 0), tab_len - 1);  // the read-out kernel already advanced the counter
    const double* e = tab + (size_t)t * 5;
    const double lr = lr_clip ? lr_clip[clip] : e[3];
    if (e[4] >= 0.0) H.h[0] = (float)e[4];
    float4 c;
    c.x = (float)(lr * e[0]);
    c.y = kind == OPT_SGD ? (float)e[1] : (float)(lr * e[1]);
    c.z = (float)e[2];
    c.w = (float)(1.0 - lr * wd);
    const size_t idx = (size_t)row * kFS + f;
    float p = coef[idx], mo = mom[idx], ve = vel[idx], blo, bhi;
    box_bounds(c0[idx], ratio, blo, bhi);
    opt_clamp_update(kind, p, mo, ve, grad[idx], blo, bhi, c, H);
    coef[idx] = p; mom[idx] = mo; vel[idx] = ve;
    if (improved[clip]) best[idx] = p;
}
void launch_opt_rows(int kind, float* coef, const float* grad, float* mom, float* vel, const float* c0, float ratio, float* best,
                     const int* improved, const int* frame_off, int B, int NF, const double* tab, int tab_len, const int* step,
                     const double* lr_clip, double wd, const float* h8, int nband, hipStream_t st) {
    OptHyp H;
    for (int i = 0; i < 8; ++i) H.h[i] = h8[i];
    hipLaunchKernelGGL(opt_rows_kernel, dim3(NF), dim3(256), 0, st, kind, coef, grad, mom, vel, c0, ratio, best, improved, frame_off,
                       B, tab, tab_len, step, lr_clip, wd, H, nband);
}

// torch.optim.lr_scheduler.ReduceLROnPlateau(mode 'min', threshold_mode 'rel', cooldown 0), one state per clip, stepped with
// the clip's loss AFTER the optimiser step like the reference's loop (embedding/multibit_embedder.py:112-113; schedulers.py:4):
// python-float (double) arithmetic as torch's.  state: [B][3] doubles (best, num_bad_epochs, unused); lr_clip [B].
__global__ __launch_bounds__(256) void plateau_kernel(const float* __restrict__ loss, double* __restrict__ state,
                                                      double* __restrict__ lr_clip, int B, double factor, int patience,
                                                      double threshold, double min_lr, double eps) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const double cur = (double)loss[b];
    double best = state[3 * b], bad = state[3 * b + 1];
    if (cur < best * (1.0 - threshold)) { best = cur; bad = 0.0; }
    else bad += 1.0;
    if (bad > (double)patience) {
        const double old = lr_clip[b], nw = fmax(old * factor, min_lr);
        if (old - nw > eps) lr_clip[b] = nw;
        bad = 0.0;
    }
    state[3 * b] = best; state[3 * b + 1] = bad;
}
void launch_plateau(const float* loss, double* state, double* lr_clip, int B, double factor, int patience, double threshold,
                    double min_lr, double eps, hipStream_t st) {
    hipLaunchKernelGGL(plateau_kernel, dim3((B + 255) / 256), dim3(256), 0, st, loss, state, lr_clip, B, factor, patience, threshold,
                       min_lr, eps);
}

// out[c][r] = in[r][c] for in [R][C] row-major (32 x 32 tiles through LDS, both sides coalesced).  Used to turn the
// weight-gradient contraction over rows, dW = dZ^T X, into the K-contiguous NT form of the GEMM kernels.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) out[(size_t)c * R + r] = tile[tx][i];
    }
}
// out[c] = sum_r in[r][c]  (bias gradient; f64 accumulation, fixed order)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    __shared__ double red[8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
    double s = 0.0;
    if (c < C)
        for (int r = g; r < R; r += 8) s += (double)in[(size_t)r * C + c];
    red[g][threadIdx.x & 31] = s;
    __syncthreads();
    if (g == 0 && c < C) {
        double t = 0.0;
        for (int i = 0; i < 8; ++i) t += red[i][threadIdx.x & 31];
        out[c] = (float)t;
    }
}
void launch_transpose(const float* in, float* out, int R, int C, hipStream_t st) {
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, st, in, out, R, C);
}
void launch_colsum(const float* in, float* out, int R, int C, hipStream_t st) {
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 31) / 32), dim3(256), 0, st, in, out, R, C);
}

static inline unsigned gridn(size_t n) {
    size_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}
void launch_polar_decompose(const void* spec, float* mag, float* phase, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(polar_decompose_kernel, dim3(gridn(n)), dim3(256), 0, st, (const cf*)spec, mag, phase, n);
}
void launch_polar_decompose_bwd(const void* spec, const float* gmag, const float* gphase, void* gspec, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(polar_decompose_bwd_kernel, dim3(gridn(n)), dim3(256), 0, st, (const cf*)spec, gmag, gphase, (cf*)gspec, n);
}
void launch_polar_assemble(const float* mag, const float* phase, void* spec, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(polar_assemble_kernel, dim3(gridn(n)), dim3(256), 0, st, mag, phase, (cf*)spec, n);
}
void launch_polar_assemble_bwd(const float* mag, const float* phase, const void* gspec, float* gmag, float* gphase, size_t n,
                               hipStream_t st) {
    hipLaunchKernelGGL(polar_assemble_bwd_kernel, dim3(gridn(n)), dim3(256), 0, st, mag, phase, (const cf*)gspec, gmag, gphase, n);
}
void launch_normalize_bwd(const float* x, const float* g, float* dx, const int* off, const int* len, int B, hipStream_t st) {
    hipLaunchKernelGGL(normalize_bwd_kernel, dim3(B), dim3(256), 0, st, x, g, dx, off, len);
}
void launch_nadam_clamp(float* p, const float* g, float* m, float* v, const float* lo, const float* hi, size_t n, float c_grad,
                        float c_mom, float bias_corr2, float beta1, float beta2, float eps, hipStream_t st) {
    const float4 hyp = make_float4(1.0f - beta1, beta2, 1.0f - beta2, eps);
    hipLaunchKernelGGL(nadam_clamp_kernel, dim3(gridn(n)), dim3(256), 0, st, p, g, m, v, lo, hi, n, c_grad, c_mom,
                       1.0f / bias_corr2, hyp);
}

}  // namespace aware
