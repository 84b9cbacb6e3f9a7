// extern "C" entry points of libaware_hip (see include/aware_hip.h for the contract and the
// reference code each one replaces).  Host-side orchestration only: geometry tables,
// workspace carving, launch sequences, optional hipGraph capture of one optimiser iteration.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <algorithm>
#include <vector>

#include "../../include/aware_hip.h"
#include "common.hpp"
#include "kernels.h"

using namespace aware;

static thread_local std::string g_last_err;

#define HIPCHK(expr)                                                                  \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            g_last_err = std::string(#expr) + ": " + hipGetErrorString(_e);           \
            return AWARE_E_HIP;                                                       \
        }                                                                             \
    } while (0)
#define LAUNCHCHK() HIPCHK(hipGetLastError())

// Optional per-launch timing of one optimiser iteration (aware_embed_profile): an event is
// recorded on the launch stream after every kernel; consecutive events bracket one kernel.
struct LaunchProfiler {
    hipStream_t st;
    std::vector<hipEvent_t> ev;
    std::vector<int> kind;
};
static thread_local LaunchProfiler* g_prof = nullptr;
static inline void prof_mark(int kind) {
    if (!g_prof) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, g_prof->st);
    g_prof->ev.push_back(e);
    g_prof->kind.push_back(kind);
}
#define PROF(kind) prof_mark(kind)
// kernel kinds reported by aware_embed_profile
enum { K_SYNTH = 0, K_ANALYSIS = 1, K_GEMM = 2, K_MELNORM = 3, K_INLRELU = 4, K_HEAD = 5, K_SYNTH_ADJ = 6,
       K_ANALYSIS_ADJ = 7, K_MISC = 8, K_GEMM_CLIP_FWD = 9, K_GEMM_CLIP_BWD = 10, K_GEMM_X3_FWD = 11, K_GEMM_X3_BWD = 12 };

static int absmax_into_scratch(const float* in, const int* off, const int* len, int B, int max_len, void* scratch,
                               unsigned long long** pmax_out, int** pcount_out, int* ps_out, hipStream_t st);

// ---------------------------------------------------------------------------------------------
struct aware_plan {
    PlanDev dev;
    void* mem = nullptr;
};

struct aware_batch {
    int B = 0;
    std::vector<int> n, in_off, T, frame_off, pool_off, out_off, out_len, pc_in, pc_syn;
    int NF = 0, NP = 0, NS = 0, max_frames = 0, max_len = 0, pstride = 0;
    int uniform_tp = 0;   // pooled frames per clip when all clips agree (else 0)
    int synth_run = kSynthBlocks;   // hop blocks per synthesis run: 16, or 8 / 4 when 16 would leave the chip short of waves
    // device tables (one allocation)
    int* d_mem = nullptr;
    int *d_frame_off = nullptr, *d_pool_off = nullptr, *d_in_off = nullptr, *d_in_len = nullptr, *d_out_off = nullptr,
        *d_out_len = nullptr, *d_pc_in = nullptr, *d_pc_syn = nullptr;
    // streaming DSP kernels: one workgroup = kStreamWaves runs of one clip; flat tables (clip << 12 | workgroup within the
    // clip) so that a ragged batch launches exactly the workgroups that have work
    int an_run = 0, n_an_wg = 0, n_syn_wg = 0;
    int *d_an_wg = nullptr, *d_syn_wg = nullptr;
    std::vector<int> an_wg, syn_wg;
    int* d_order = nullptr;            // clips longest first (dispatch order of the ragged GEMM: short clips fill the tail)
    std::vector<int> order;
};

struct aware_detector {
    int n_mels = 0, n_layers = 0, nbits = 0;
    int band_lo = 0, nband = 0;
    int ch[8] = {0};
    int maxc = 0;
    float* mem = nullptr;
    float* melT = nullptr;   // [n_mels][256]  (Bt of the forward mel GEMM)
    float* melB = nullptr;   // [256][n_mels]  (Bt of its data-gradient)
    float* w[8] = {nullptr};   // [Cout][Cin]
    float* wT[8] = {nullptr};  // [Cin][Cout]
    // the same two operands split into three bf16 planes in MFMA fragment order (gemm_x3.hip); null when the
    // shape is not served by that kernel
    void* wpk[8] = {nullptr};
    void* wTpk[8] = {nullptr};
    void* melTpk = nullptr;
    void* melBpk = nullptr;
    void* lastpk = nullptr;    // last conv, rows zero-padded to a multiple of 16 (read-out kernel)
    void* lastTpk = nullptr;   // its transpose [Cin][64], k zero-padded to 64
    void* pkmem = nullptr;
    // the conv blocks' two operands as f16 two-term images (gemm_h2.hip: planes + per-channel inverse scales)
    void* wh2[8] = {nullptr};
    void* wTh2[8] = {nullptr};
    void* h2mem = nullptr;
    float* bias[8] = {nullptr};
    // the mel filter bank as two taps per band column (a triangular bank has at most two adjacent non-zero weights per bin):
    // melw [kFS] float2, melm [kFS] first tap's mel index (<= 126); null when the basis handed over is not of that form
    void* mel2mem = nullptr;
    float2* melw = nullptr;
    unsigned char* melm = nullptr;
    // ... and per filter as a short run of adjacent band columns (forward): melf_w [n_mels][kMelTapsB], melf_s [n_mels]; null
    // when a filter's support is longer than the folded analysis kernel takes (kMelTapsA / kMelTapsB columns)
    float* melf_w = nullptr;
    unsigned char* melf_s = nullptr;
};

extern "C" int aware_version(void) { return 300; }
extern "C" const char* aware_last_hip_error(void) { return g_last_err.c_str(); }

// ---------------------------------------------------------------------------------------------
extern "C" int aware_plan_create(aware_plan** out, int n_fft, int hop, int win_length, int window, int band_lo_bin,
                                 int band_hi_bin) {
    if (!out) return AWARE_E_BADARG;
    if (n_fft != kNfft || hop != kHop || win_length != kNfft) return AWARE_E_UNSUPPORTED;
    if (window != 0 && window != 1) return AWARE_E_BADARG;
    const int nband = band_hi_bin - band_lo_bin + 1;
    if (band_lo_bin < 1 || band_hi_bin > 511 || nband < 1 || nband > kFS) return AWARE_E_UNSUPPORTED;
    const double PI = 3.14159265358979323846;
    std::vector<float> h(2 * 512 + 2 * 512 + 1024 + 1024 + 3 * 768);
    float* tw512 = h.data();
    float* tw1024 = tw512 + 1024;
    float* win = tw1024 + 1024;
    float* win2 = win + 1024;
    for (int j = 0; j < 512; ++j) {
        tw512[2 * j] = (float)cos(2 * PI * j / 512);
        tw512[2 * j + 1] = (float)-sin(2 * PI * j / 512);
        tw1024[2 * j] = (float)cos(2 * PI * j / 1024);
        tw1024[2 * j + 1] = (float)-sin(2 * PI * j / 1024);
    }
    for (int i = 0; i < 1024; ++i) {
        // torch.hann_window / torch.hamming_window (periodic), utils/audio/stft.py:19-25
        double w = (window == 0) ? 0.5 - 0.5 * cos(2 * PI * i / 1024) : 0.54 - 0.46 * cos(2 * PI * i / 1024);
        win[i] = (float)w;
        win2[i] = win[i] * win[i];
    }
    // overlap-add envelope tables, summed in ascending frame order like ola_envelope_loop
    float* env = win2 + 1024;
    for (int q = 0; q < 768; ++q) {
        float e = 0.f;
        for (int t = 0; t <= (q >> 8); ++t) e += win2[q - 256 * t];
        env[q] = e;                                               // head: p = q < 768
        float ei = 0.f;
        for (int off = (q & 255) + 768; off >= (q & 255); off -= 256) ei += win2[off];
        env[768 + q] = ei;                                        // interior: p mod 256 = q & 255
        float et = 0.f;
        for (int off = (q & 255) + 768; off >= q + 256; off -= 256) et += win2[off];
        env[1536 + q] = et;                                       // tail: p = 256*T + q
    }
    aware_plan* p = new aware_plan();
    HIPCHK(hipMalloc(&p->mem, h.size() * sizeof(float)));
    HIPCHK(hipMemcpy(p->mem, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    float* d = (float*)p->mem;
    p->dev.tw512 = (const cf*)d;
    p->dev.tw1024 = (const cf*)(d + 1024);
    p->dev.window = d + 2048;
    p->dev.window2 = d + 3072;
    p->dev.env_tab = d + 4096;
    p->dev.band_lo = band_lo_bin;
    p->dev.nband = nband;
    *out = p;
    return AWARE_OK;
}
extern "C" void aware_plan_destroy(aware_plan* p) {
    if (!p) return;
    if (p->mem) (void)hipFree(p->mem);
    delete p;
}

// ---------------------------------------------------------------------------------------------
extern "C" int aware_batch_create(aware_batch** out, int B, const int* n_samples, const int* in_offsets) {
    if (!out || B < 1 || !n_samples) return AWARE_E_BADARG;
    aware_batch* b = new aware_batch();
    b->B = B;
    b->n.assign(n_samples, n_samples + B);
    b->in_off.resize(B);
    b->T.resize(B);
    b->frame_off.resize(B + 1);
    b->pool_off.resize(B + 1);
    b->out_off.resize(B);
    b->out_len.resize(B);
    b->pc_in.resize(B);
    b->pc_syn.resize(B);
    int acc = 0, max_pc = 1;
    // run length of the synthesis kernels (one wave of the streaming kernels per run): the longest of 16 / 12 / 8 / 6 / 4
    // hop blocks that still gives ~16 waves per CU; a run of r blocks transforms r + 3 frames, so short runs cost more
    // arithmetic and are only worth it while the chip would otherwise idle
    // (measured, 3 s clips, us synthesis / adjoint: B = 256: 16 66/66, 12 65/64, 8 77/75; B = 128: 12 43/47, 8 40/42, 6 39/42,
    // 4 51/52; B = 64: 8 31/30, 6 28/28, 4 28/28)
    for (int rb : {12, 8, 6, 4}) {          // (16 never measured better than 12)
        long runs = 0;
        for (int i = 0; i < B; ++i) runs += (n_samples[i] / kHop + rb - 1) / rb;
        // the staged adjoint folds the reflect pads inside the first / last segment: keep those at >= 3 blocks
        bool ok = true;
        for (int i = 0; i < B && ok; ++i) {
            const int nb = n_samples[i] / kHop, ns = (nb + rb - 1) / rb;
            if (ns > 1 && nb / ns < 3) ok = false;
        }
        if (!ok) continue;
        b->synth_run = rb;
        if (runs >= 4096) break;
    }
    b->frame_off[0] = 0;
    b->pool_off[0] = 0;
    for (int i = 0; i < B; ++i) {
        const int n = n_samples[i];
        // torch.stft's reflect padding needs n > n_fft/2
        if (n <= kHalf) { delete b; return AWARE_E_BADARG; }
        b->in_off[i] = in_offsets ? in_offsets[i] : acc;
        acc += n;
        const int T = 1 + n / kHop;
        b->T[i] = T;
        b->frame_off[i + 1] = b->frame_off[i] + T;
        b->pool_off[i + 1] = b->pool_off[i] + ((T / 2 + 31) & ~31);   // pooled rows are 32-aligned per clip
        b->out_off[i] = kHop * (b->frame_off[i] - i);
        b->out_len[i] = kHop * (T - 1);
        b->pc_in[i] = (n + 4095) / 4096;
        int nseg = (T - 1 + b->synth_run - 1) / b->synth_run;
        if (nseg < 1) nseg = 1;
        b->pc_syn[i] = nseg;
        if (T > b->max_frames) b->max_frames = T;
        if (n > b->max_len) b->max_len = n;
        if (b->pc_in[i] > max_pc) max_pc = b->pc_in[i];
        if (nseg > max_pc) max_pc = nseg;
    }
    b->NF = b->frame_off[B];
    b->NP = b->pool_off[B];
    b->uniform_tp = b->T[0] / 2;
    for (int i = 1; i < B; ++i)
        if (b->T[i] / 2 != b->uniform_tp) b->uniform_tp = 0;
    b->NS = kHop * (b->NF - B);
    b->pstride = max_pc;
    {
        // (runs longer than 12 frames measured slower even when the chip has waves to spare: B = 256 x 3 s, analysis / adjoint
        //  67 / 118 us at R = 12 against 69 / 136 at R = 16)
        constexpr int kAnalysisMaxRun = 12;
        // frames per analysis run: the longest R in [4, 12] that still gives 4096 waves; for a uniform batch then the nearest
        // shorter R whose run count per clip is a multiple of the waves of a workgroup (measurements: dsp_stream.hip)
        auto total_runs = [&](int r) { long t = 0; for (int i = 0; i < B; ++i) t += (b->T[i] + r - 1) / r; return t; };
        int R = 4;
        for (int cand = kAnalysisMaxRun; cand >= 4; --cand)
            if (total_runs(cand) >= 4096) { R = cand; break; }
        bool same = true;
        for (int i = 1; i < B; ++i) same = same && b->T[i] == b->T[0];
        if (same)
            for (int cand = R; cand >= 4 && cand >= R - 3; --cand)
                if (((b->T[0] + cand - 1) / cand) % kStreamWaves == 0) { R = cand; break; }
        b->an_run = R;
        if (B < (1 << 19) && b->max_frames / 4 < 4096 * kStreamWaves) {
            for (int i = 0; i < B; ++i) {
                const int ra = (b->T[i] + R - 1) / R, rs = std::max(1, (b->T[i] - 1 + b->synth_run - 1) / b->synth_run);
                for (int w = 0; w < (ra + kStreamWaves - 1) / kStreamWaves; ++w) b->an_wg.push_back((i << 12) | w);
                for (int w = 0; w < (rs + kStreamWaves - 1) / kStreamWaves; ++w) b->syn_wg.push_back((i << 12) | w);
            }
        }
        b->n_an_wg = (int)b->an_wg.size();
        b->n_syn_wg = (int)b->syn_wg.size();
    }
    b->order.resize(B);
    for (int i = 0; i < B; ++i) b->order[i] = i;
    std::stable_sort(b->order.begin(), b->order.end(), [&](int x, int y) { return b->T[x] > b->T[y]; });
    const size_t ints = (size_t)(B + 1) * 2 + (size_t)B * 7 + b->an_wg.size() + b->syn_wg.size();
    HIPCHK(hipMalloc((void**)&b->d_mem, ints * sizeof(int)));
    int* d = b->d_mem;
    auto up = [&](int*& dst, const std::vector<int>& v) -> hipError_t {
        dst = d;
        d += v.size();
        return hipMemcpy(dst, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice);
    };
    HIPCHK(up(b->d_frame_off, b->frame_off));
    HIPCHK(up(b->d_pool_off, b->pool_off));
    HIPCHK(up(b->d_in_off, b->in_off));
    HIPCHK(up(b->d_in_len, b->n));
    HIPCHK(up(b->d_out_off, b->out_off));
    HIPCHK(up(b->d_out_len, b->out_len));
    HIPCHK(up(b->d_pc_in, b->pc_in));
    HIPCHK(up(b->d_pc_syn, b->pc_syn));
    HIPCHK(up(b->d_order, b->order));
    if (b->n_an_wg) HIPCHK(up(b->d_an_wg, b->an_wg));
    if (b->n_syn_wg) HIPCHK(up(b->d_syn_wg, b->syn_wg));
    *out = b;
    return AWARE_OK;
}
extern "C" void aware_batch_destroy(aware_batch* b) {
    if (!b) return;
    if (b->d_mem) (void)hipFree(b->d_mem);
    delete b;
}
extern "C" int aware_batch_total_frames(const aware_batch* b) { return b ? b->NF : AWARE_E_BADARG; }
extern "C" int aware_batch_total_pooled(const aware_batch* b) { return b ? b->NP : AWARE_E_BADARG; }
extern "C" int aware_batch_total_out(const aware_batch* b) { return b ? b->NS : AWARE_E_BADARG; }
extern "C" int aware_batch_out_offset(const aware_batch* b, int i) {
    return (b && i >= 0 && i < b->B) ? b->out_off[i] : AWARE_E_BADARG;
}
extern "C" int aware_batch_out_length(const aware_batch* b, int i) {
    return (b && i >= 0 && i < b->B) ? b->out_len[i] : AWARE_E_BADARG;
}
extern "C" int aware_batch_frames(const aware_batch* b, int i) {
    return (b && i >= 0 && i < b->B) ? b->T[i] : AWARE_E_BADARG;
}
extern "C" size_t aware_batch_scratch_bytes(const aware_batch* b) {
    return b ? (size_t)b->B * b->pstride * sizeof(unsigned long long) + 256 : 0;
}

// The DSP kernels exist in two forms: streaming wave kernels (dsp_stream.hip; default wherever the band lies inside bins
// 1..256) and workgroup-staged kernels (dsp_kernels.hip; any band, full-spectrum input/output, and the form the
// streaming kernels are tested against).  dsp_path: 0 = streaming where supported, 1 = staged.
static void run_analysis(AnalysisLaunch& L, int dsp_path, hipStream_t st) {
    if (dsp_path == 0 && !L.full && stream_supported(L.plan)) { L.stream = 1; launch_analysis_stream(L, st); }
    else launch_analysis(L, st);
}
static void run_synth(SynthLaunch& S, int dsp_path, hipStream_t st) {
    if (dsp_path == 0 && !S.full && stream_supported(S.plan)) { S.stream = 1; launch_synth_stream(S, st); }
    else launch_synth(S, st);
}

// ---------------------------------------------------------------------------------------------
// workspace carving
struct Carver {
    char* base;
    size_t off = 0, cap;
    bool ok = true;
    Carver(void* p, size_t c) : base((char*)p), cap(c) {}
    template <typename Tp> Tp* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        Tp* r = (Tp*)(base + off);
        off += count * sizeof(Tp);
        if (off > cap) ok = false;
        return r;
    }
};

// ---------------------------------------------------------------------------------------------
extern "C" int aware_stft(const aware_plan* plan, const aware_batch* b, const float* audio, int normalize, void* spec,
                          void* scratch, void* stream) {
    if (!plan || !b || !audio || !spec || (normalize && !scratch)) return AWARE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* pmax = (unsigned long long*)scratch;
    if (normalize) {
        launch_absmax_partials(audio, b->d_in_off, b->d_in_len, pmax, b->pstride, b->B, b->max_len, st);
        LAUNCHCHK();
    }
    AnalysisLaunch L;
    L.plan = plan->dev; L.frame_off = b->d_frame_off; L.B = b->B; L.max_frames = b->max_frames; L.run_frames = b->an_run; L.wg_tab = b->d_an_wg; L.n_wg = b->n_an_wg;
    L.sig = audio; L.sig_off = b->d_in_off; L.sig_len = b->d_in_len;
    L.pmax = normalize ? pmax : nullptr; L.pcount = b->d_pc_in; L.pstride = b->pstride;
    L.full = spec;
    launch_analysis(L, st);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_stft_band(const aware_plan* plan, const aware_batch* b, const float* audio, int normalize,
                               float* mag, void* phasor, void* scratch, void* stream) {
    if (!plan || !b || !audio || (!mag && !phasor) || (normalize && !scratch)) return AWARE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* pmax = (unsigned long long*)scratch;
    if (normalize) {
        launch_absmax_partials(audio, b->d_in_off, b->d_in_len, pmax, b->pstride, b->B, b->max_len, st);
        LAUNCHCHK();
    }
    AnalysisLaunch L;
    L.plan = plan->dev; L.frame_off = b->d_frame_off; L.B = b->B; L.max_frames = b->max_frames; L.run_frames = b->an_run; L.wg_tab = b->d_an_wg; L.n_wg = b->n_an_wg;
    L.sig = audio; L.sig_off = b->d_in_off; L.sig_len = b->d_in_len;
    L.pmax = normalize ? pmax : nullptr; L.pcount = b->d_pc_in; L.pstride = b->pstride;
    L.mag = mag; L.unit = phasor; L.unit_default = 1.f;
    run_analysis(L, 0, st);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_istft(const aware_plan* plan, const aware_batch* b, const void* spec, int normalize, float* out,
                           void* scratch, void* stream) {
    if (!plan || !b || !spec || !out || (normalize && !scratch)) return AWARE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* pmax = (unsigned long long*)scratch;
    SynthLaunch S;
    S.plan = plan->dev; S.frame_off = b->d_frame_off; S.B = b->B; S.max_frames = b->max_frames; S.run_blocks = b->synth_run; S.wg_tab = b->d_syn_wg; S.n_wg = b->n_syn_wg;
    S.full = spec; S.out = out; S.pmax = normalize ? pmax : nullptr; S.pstride = b->pstride;
    launch_synth(S, st);
    LAUNCHCHK();
    if (normalize) {
        launch_finish(out, b->d_frame_off, pmax, b->d_pc_syn, b->pstride, nullptr, out, b->d_out_off, b->B,
                      b->max_frames, st);
        LAUNCHCHK();
    }
    return AWARE_OK;
}

// ---- backward of the two transforms for the differentiable plug-in seam (interfaces/audio.py:6-9) ----
extern "C" int aware_stft_bwd(const aware_plan* plan, const aware_batch* b, const void* grad_spec, float* grad_audio,
                              void* stream) {
    if (!plan || !b || !grad_spec || !grad_audio) return AWARE_E_BADARG;
    // the staged synthesis kernel in adjoint mode: overlap-add of the windowed rfft adjoints, then the fold of the two reflect
    // pads for a clip of any length n > 512 at the clip's offset in the caller's ragged array
    if (b->synth_run > 0)
        for (int i = 0; i < b->B; ++i) {
            const int nb = b->T[i] - 1, ns = (nb + b->synth_run - 1) / b->synth_run;
            if (ns > 1 && nb / ns < 3) return AWARE_E_UNSUPPORTED;          // (aware_batch_create never builds such a batch)
        }
    SynthLaunch S;
    S.plan = plan->dev; S.frame_off = b->d_frame_off; S.B = b->B; S.max_frames = b->max_frames; S.run_blocks = b->synth_run; S.wg_tab = b->d_syn_wg; S.n_wg = b->n_syn_wg;
    S.full = grad_spec; S.out = grad_audio; S.adjoint = 1; S.pstride = b->pstride;
    S.sig_off = b->d_in_off; S.sig_len = b->d_in_len;
    launch_synth(S, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_istft_bwd(const aware_plan* plan, const aware_batch* b, const float* grad_audio, void* grad_spec,
                               void* stream) {
    if (!plan || !b || !grad_audio || !grad_spec) return AWARE_E_BADARG;
    AnalysisLaunch L;
    L.plan = plan->dev; L.frame_off = b->d_frame_off; L.B = b->B; L.max_frames = b->max_frames; L.run_frames = b->an_run; L.wg_tab = b->d_an_wg; L.n_wg = b->n_an_wg;
    L.sig = grad_audio; L.sig_off = b->d_out_off; L.sig_len = b->d_out_len;
    L.pcount = b->d_pc_syn; L.pstride = b->pstride;
    L.full = grad_spec; L.adjoint = 1;
    launch_analysis(L, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" void aware_detector_destroy(aware_detector* d);
// host staging of the detector's parameters (plain + transposed f32 copies, bf16x3 fragment images) and upload; `alloc`
// = false re-uses the device buffers of a detector created with the same shapes (aware_detector_update)
static int detector_upload(aware_detector* d, const float* mel_basis, const float* const* weights,
                           const float* const* biases, bool alloc) {
    const int n_mels = d->n_mels, n_layers = d->n_layers;
    const int* channels = d->ch;
    size_t total = (size_t)n_mels * kFS * 2;
    for (int i = 0; i < n_layers; ++i) total += (size_t)channels[i] * channels[i + 1] * 2 + channels[i + 1];
    std::vector<float> h(total, 0.f);
    size_t o = 0;
    const int nbins = kNfft / 2 + 1;
    const int lo = d->band_lo, nb = d->nband;
    size_t o_melT = o; o += (size_t)n_mels * kFS;
    size_t o_melB = o; o += (size_t)kFS * n_mels;
    // only the in-band columns of the mel basis ever multiply non-zero magnitudes
    // (multibit_embedder.py:104, multibit_detector.py:34-37 zero the rest)
    for (int j = 0; j < n_mels; ++j)
        for (int f = 0; f < nb; ++f) {
            float v = mel_basis[(size_t)j * nbins + lo + f];
            h[o_melT + (size_t)j * kFS + f] = v;
            h[o_melB + (size_t)f * n_mels + j] = v;
        }
    {
        // two-tap form of the band's columns, if the basis has it (any triangular filter bank does)
        std::vector<float2> tw(kFS, make_float2(0.f, 0.f));
        std::vector<unsigned char> tm(kFS, 0);
        bool sparse = n_mels == 128;
        for (int f = 0; f < nb && sparse; ++f) {
            int first = -1, count = 0, lastnz = -1;
            for (int j = 0; j < n_mels; ++j)
                if (h[o_melB + (size_t)f * n_mels + j] != 0.f) { if (first < 0) first = j; lastnz = j; ++count; }
            if (count == 0) continue;
            if (count > 2 || lastnz - first > 1) { sparse = false; break; }
            const int m1 = first < n_mels - 1 ? first : n_mels - 2;
            tm[f] = (unsigned char)m1;
            tw[f] = make_float2(h[o_melB + (size_t)f * n_mels + m1], h[o_melB + (size_t)f * n_mels + m1 + 1]);
        }
        // forward: filter m as kMelTapsA (m < 64) / kMelTapsB adjacent columns starting at fs[m]
        std::vector<float> fw((size_t)128 * kMelTapsB, 0.f);
        std::vector<unsigned char> fs(128, 0);
        bool runs = sparse;
        for (int m = 0; m < n_mels && runs; ++m) {
            int first = -1, lastnz = -1;
            for (int f = 0; f < nb; ++f)
                if (h[o_melT + (size_t)m * kFS + f] != 0.f) { if (first < 0) first = f; lastnz = f; }
            if (first < 0) continue;
            const int taps = m < 64 ? kMelTapsA : kMelTapsB;
            const int start = first <= kFS - kMelTapsB ? first : kFS - kMelTapsB;
            if (lastnz - start >= taps) { runs = false; break; }
            fs[m] = (unsigned char)start;
            for (int j = 0; j < taps; ++j) fw[(size_t)m * kMelTapsB + j] = h[o_melT + (size_t)m * kFS + start + j];
        }
        if (sparse) {
            const size_t bytes = kFS * (sizeof(float2) + 1) + 128 * (kMelTapsB * sizeof(float) + 1) + 64;
            if (!d->mel2mem) HIPCHK(hipMalloc(&d->mel2mem, bytes));
            char* base = (char*)d->mel2mem;
            d->melw = (float2*)base;
            d->melf_w = (float*)(base + kFS * sizeof(float2));
            d->melm = (unsigned char*)(base + kFS * sizeof(float2) + 128 * kMelTapsB * sizeof(float));
            d->melf_s = d->melm + kFS;
            HIPCHK(hipMemcpy(d->melw, tw.data(), kFS * sizeof(float2), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(d->melm, tm.data(), kFS, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(d->melf_w, fw.data(), fw.size() * sizeof(float), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(d->melf_s, fs.data(), 128, hipMemcpyHostToDevice));
            if (!runs) { d->melf_w = nullptr; d->melf_s = nullptr; }
        } else {
            d->melw = nullptr;
            d->melm = nullptr;
            d->melf_w = nullptr;
            d->melf_s = nullptr;
        }
    }
    size_t o_w[8], o_wT[8], o_b[8];
    for (int l = 0; l < n_layers; ++l) {
        const int ci = channels[l], co = channels[l + 1];
        o_w[l] = o; o += (size_t)ci * co;
        o_wT[l] = o; o += (size_t)ci * co;
        o_b[l] = o; o += co;
        for (int r = 0; r < co; ++r)
            for (int c = 0; c < ci; ++c) {
                float v = weights[l][(size_t)r * ci + c];
                h[o_w[l] + (size_t)r * ci + c] = v;
                h[o_wT[l] + (size_t)c * co + r] = v;
            }
        for (int r = 0; r < co; ++r) h[o_b[l] + r] = biases && biases[l] ? biases[l][r] : 0.f;
    }
    if (alloc) HIPCHK(hipMalloc((void**)&d->mem, total * sizeof(float)));
    HIPCHK(hipMemcpy(d->mem, h.data(), total * sizeof(float), hipMemcpyHostToDevice));
    d->melT = d->mem + o_melT;
    d->melB = d->mem + o_melB;
    for (int l = 0; l < n_layers; ++l) { d->w[l] = d->mem + o_w[l]; d->wT[l] = d->mem + o_wT[l]; d->bias[l] = d->mem + o_b[l]; }
    {
        size_t pk_total = 0, o_pk[8], o_pkT[8];
        for (int l = 0; l < n_layers; ++l) {
            const int ci = channels[l], co = channels[l + 1];
            o_pk[l] = o_pkT[l] = (size_t)-1;
            if (co % 128 == 0 && ci % 64 == 0) { o_pk[l] = pk_total; pk_total += x3_packed_bytes(co, ci); }
            if (ci % 128 == 0 && co % 64 == 0) { o_pkT[l] = pk_total; pk_total += x3_packed_bytes(ci, co); }
        }
        const size_t o_mT = pk_total; pk_total += x3_packed_bytes(n_mels, kFS);
        const size_t o_mB = pk_total; pk_total += x3_packed_bytes(kFS, n_mels);
        // read-out kernel operands (last conv block, C <= 64 channels)
        const int cil = channels[n_layers - 1], col = channels[n_layers], colp = 16 * ((col + 15) / 16);
        const bool ro = n_layers >= 2 && readout_x3_supported(1, cil, col);
        size_t o_lp = 0, o_lT = 0;
        if (ro) {
            o_lp = pk_total; pk_total += x3_packed_bytes(colp, cil);
            o_lT = pk_total; pk_total += x3_packed_bytes(cil, 64);
        }
        std::vector<uint16_t> hp(pk_total / 2 + 8, 0);
        if (ro) {
            std::vector<float> wp((size_t)colp * cil, 0.f), wt((size_t)cil * 64, 0.f);
            for (int r = 0; r < col; ++r)
                for (int c = 0; c < cil; ++c) {
                    const float v = weights[n_layers - 1][(size_t)r * cil + c];
                    wp[(size_t)r * cil + c] = v;
                    wt[(size_t)c * 64 + r] = v;
                }
            x3_pack(wp.data(), colp, cil, hp.data() + o_lp / 2);
            x3_pack(wt.data(), cil, 64, hp.data() + o_lT / 2);
        }
        x3_pack(h.data() + o_melT, n_mels, kFS, hp.data() + o_mT / 2);
        x3_pack(h.data() + o_melB, kFS, n_mels, hp.data() + o_mB / 2);
        for (int l = 0; l < n_layers; ++l) {
            const int ci = channels[l], co = channels[l + 1];
            if (o_pk[l] != (size_t)-1) x3_pack(h.data() + o_w[l], co, ci, hp.data() + o_pk[l] / 2);
            if (o_pkT[l] != (size_t)-1) x3_pack(h.data() + o_wT[l], ci, co, hp.data() + o_pkT[l] / 2);
        }
        if (alloc) HIPCHK(hipMalloc(&d->pkmem, pk_total + 16));
        HIPCHK(hipMemcpy(d->pkmem, hp.data(), pk_total, hipMemcpyHostToDevice));
        if (ro) { d->lastpk = (char*)d->pkmem + o_lp; d->lastTpk = (char*)d->pkmem + o_lT; }
        d->melTpk = (char*)d->pkmem + o_mT;
        d->melBpk = (char*)d->pkmem + o_mB;
        for (int l = 0; l < n_layers; ++l) {
            if (o_pk[l] != (size_t)-1) d->wpk[l] = (char*)d->pkmem + o_pk[l];
            if (o_pkT[l] != (size_t)-1) d->wTpk[l] = (char*)d->pkmem + o_pkT[l];
        }
    }
    {
        // f16 two-term images, packed on the device from the f32 copies just uploaded
        size_t total2 = 0, o_h[8], o_hT[8];
        for (int l = 0; l < n_layers; ++l) {
            const int ci = channels[l], co = channels[l + 1];
            o_h[l] = o_hT[l] = (size_t)-1;
            if (gemm_clip_h2_supported(1, co, ci, ci)) { o_h[l] = total2; total2 += (h2_packed_bytes(co, ci) + 255) & ~(size_t)255; }
            if (gemm_clip_h2_supported(1, ci, co, co)) { o_hT[l] = total2; total2 += (h2_packed_bytes(ci, co) + 255) & ~(size_t)255; }
        }
        if (total2) {
            if (alloc) HIPCHK(hipMalloc(&d->h2mem, total2));
            for (int l = 0; l < n_layers; ++l) {
                const int ci = channels[l], co = channels[l + 1];
                if (o_h[l] != (size_t)-1) { d->wh2[l] = (char*)d->h2mem + o_h[l]; launch_h2_pack(d->w[l], ci, co, ci, d->wh2[l], 0); }
                if (o_hT[l] != (size_t)-1) { d->wTh2[l] = (char*)d->h2mem + o_hT[l]; launch_h2_pack(d->wT[l], co, ci, co, d->wTh2[l], 0); }
            }
            LAUNCHCHK();
            HIPCHK(hipStreamSynchronize(0));
        }
    }
    return AWARE_OK;
}

extern "C" int aware_detector_create(aware_detector** out, const aware_plan* plan, const float* mel_basis, int n_mels,
                                     int n_layers, const int* channels, const float* const* weights,
                                     const float* const* biases) {
    if (!out || !plan || !mel_basis || !channels || !weights) return AWARE_E_BADARG;
    if (n_mels != 128 || n_layers < 1 || n_layers > 7 || channels[0] != n_mels) return AWARE_E_UNSUPPORTED;
    const int cl = channels[n_layers];
    if (cl % 2 || cl > 64) return AWARE_E_UNSUPPORTED;
    for (int i = 0; i <= n_layers; ++i)
        if (channels[i] % 4) return AWARE_E_UNSUPPORTED;
    aware_detector* d = new aware_detector();
    d->n_mels = n_mels; d->n_layers = n_layers; d->nbits = cl / 2;
    d->band_lo = plan->dev.band_lo; d->nband = plan->dev.nband;
    for (int i = 0; i <= n_layers; ++i) { d->ch[i] = channels[i]; if (channels[i] > d->maxc) d->maxc = channels[i]; }
    int rc = detector_upload(d, mel_basis, weights, biases, true);
    if (rc) { aware_detector_destroy(d); return rc; }
    *out = d;
    return AWARE_OK;
}
// EXTENSION (detector training, BASELINE north_star; the reference never changes the weights): replace the parameters of a
// detector in place (same layer shapes).  Host arrays as for aware_detector_create.  Synchronous (blocking copies); the
// caller makes sure no work using the detector is in flight.
extern "C" int aware_detector_update(aware_detector* d, const float* mel_basis, const float* const* weights,
                                     const float* const* biases) {
    if (!d || !mel_basis || !weights) return AWARE_E_BADARG;
    return detector_upload(d, mel_basis, weights, biases, false);
}
// EXTENSION (detector training): the same refresh from DEVICE arrays, asynchronous on `stream` -- no host round trip of the
// 1.68 M parameters: copies, transposes, bf16 three-term and f16 two-term images all rebuilt by kernels.  dev_weights[l]:
// [Cout][Cin] f32, dev_biases[l]: [Cout] f32 (device pointers in host arrays; biases may be NULL = unchanged).
extern "C" int aware_detector_update_device(aware_detector* d, const float* const* dev_weights, const float* const* dev_biases,
                                            void* stream) {
    if (!d || !dev_weights) return AWARE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int nl = d->n_layers;
    for (int l = 0; l < nl; ++l) {
        const int ci = d->ch[l], co = d->ch[l + 1];
        if (!dev_weights[l]) return AWARE_E_BADARG;
        HIPCHK(hipMemcpyAsync(d->w[l], dev_weights[l], (size_t)ci * co * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (dev_biases && dev_biases[l])
            HIPCHK(hipMemcpyAsync(d->bias[l], dev_biases[l], (size_t)co * sizeof(float), hipMemcpyDeviceToDevice, st));
        launch_transpose(d->w[l], d->wT[l], co, ci, st);
        if (d->wpk[l]) launch_x3_pack_dev(d->w[l], ci, co, ci, co, ci, d->wpk[l], st);
        if (d->wTpk[l]) launch_x3_pack_dev(d->wT[l], co, ci, co, ci, co, d->wTpk[l], st);
        if (d->wh2[l]) launch_h2_pack(d->w[l], ci, co, ci, d->wh2[l], st);
        if (d->wTh2[l]) launch_h2_pack(d->wT[l], co, ci, co, d->wTh2[l], st);
    }
    if (d->lastpk) {
        const int cil = d->ch[nl - 1], col = d->ch[nl], colp = 16 * ((col + 15) / 16);
        launch_x3_pack_dev(d->w[nl - 1], cil, col, cil, colp, cil, d->lastpk, st);        // rows zero-padded to a multiple of 16
        launch_x3_pack_dev(d->wT[nl - 1], col, cil, col, cil, 64, d->lastTpk, st);        // k zero-padded to 64
    }
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" void aware_detector_destroy(aware_detector* d) {
    if (!d) return;
    if (d->mem) (void)hipFree(d->mem);
    if (d->pkmem) (void)hipFree(d->pkmem);
    if (d->h2mem) (void)hipFree(d->h2mem);
    if (d->mel2mem) (void)hipFree(d->mel2mem);
    delete d;
}

// Plain GEMM C[M][N] = A[M][K] * Bt^T on the bf16x3 kernel when its packed operand exists and the shape fits
// (32-row blocks play the role of clips; plain epilogue), else on the f32-MFMA kernel.
static void gemm_plain(int pipe, const float* A, int lda, const float* Bt, int ldb, const void* Bpk, const float* bias, float* C,
                       int ldc, int M, int N, int K, hipStream_t st) {
    if (pipe != 1 && Bpk && gemm_clip_x3_supported(1, N, K, lda)) {
        // tile height: a workgroup streams its whole weight slab (6 K N/tn bytes) for its rows, so taller tiles cut the L2
        // traffic of these short-K GEMMs; keep at least two workgroups per CU's worth of tiles.  M need not be a multiple
        // of the tile height (ragged batches): the last row block is partial
        const long blocks32 = (M + 31) / 32;
        const int g = (blocks32 * (N / 128) / 2 >= 512) ? 2 : 1;
        const int nb = (int)((M + 32 * g - 1) / (32 * g));
        launch_gemm_clip_x3(A, lda, Bpk, bias, C, ldc, nb, g, 32 * g, N, K, 0, nullptr, nullptr, st, nullptr, nullptr, 0, M);
    } else
        launch_gemm_nt(A, lda, Bt, ldb, bias, C, ldc, M, N, K, st);
}

// detector activations carved from a workspace
struct DetBufs {
    float* xm;        // [NF][128]
    float* x0;        // [NP][128]
    float* act[8];    // [NP][C_l+1]
    float* rstd[8];   // [B][C_l+1]
    float *mstats, *gstat, *mpart;   // mel statistics [B][128][4], [B][4], chunk partials
    int mstride;
    float* pred;      // [B][nbits]
    float* zpart;     // split-K partial slabs of the last conv [kTailSplit][NP][C_last]
    int tail;         // 1: the last conv was left as partials for the fused tail kernel
    int zslabs;       // partial slabs the forward epilogue left in zpart (fused read-out)
    // per-clip partial maxima [B][64] for the f16 two-term GEMM's scales: of x0 (index 0), of act[l] (index l + 1), and of
    // the two gradient ping-pong buffers (gmax)
    float* amax[9];
    float* gmax[2];
};
constexpr int kTailSplit = 4;
// slabs of split-K partials of the last conv: 4 from the split-K GEMM, Cin/128 from the fused forward epilogue
static int zpart_slabs(const aware_detector* d) {
    const int s = d->n_layers >= 1 ? d->ch[d->n_layers - 1] / 128 : 0;
    return s > kTailSplit ? s : kTailSplit;
}
static void carve_det(Carver& c, const aware_batch* b, const aware_detector* d, DetBufs& o) {
    o.xm = c.take<float>((size_t)b->NF * 128);
    o.x0 = c.take<float>((size_t)b->NP * 128);
    for (int l = 0; l < d->n_layers; ++l) {
        o.act[l] = c.take<float>((size_t)b->NP * d->ch[l + 1]);
        o.rstd[l] = c.take<float>((size_t)b->B * d->ch[l + 1]);
    }
    o.mstats = c.take<float>((size_t)b->B * 128 * 4);
    o.gstat = c.take<float>((size_t)b->B * 4);
    o.mstride = (b->max_frames + 31) / 32;
    o.mpart = c.take<float>((size_t)b->B * o.mstride * 256);
    o.pred = c.take<float>((size_t)b->B * d->nbits);
    o.zpart = c.take<float>((size_t)zpart_slabs(d) * b->NP * d->ch[d->n_layers]);
    o.tail = 0;
    for (int l = 0; l <= d->n_layers; ++l) o.amax[l] = c.take<float>((size_t)b->B * 64);
    o.gmax[0] = c.take<float>((size_t)b->B * 64);
    o.gmax[1] = c.take<float>((size_t)b->B * 64);
}
static size_t det_bytes(const aware_batch* b, const aware_detector* d) {
    size_t f = (size_t)b->NF * 128 + (size_t)b->NP * 128 + (size_t)b->B * (128 * 4 + 4 + d->nbits) +
               (size_t)b->B * ((b->max_frames + 31) / 32) * 256;
    for (int l = 0; l < d->n_layers; ++l) f += (size_t)(b->NP + b->B) * d->ch[l + 1];
    f += (size_t)zpart_slabs(d) * b->NP * d->ch[d->n_layers];
    f += (size_t)b->B * 64 * (d->n_layers + 3);
    return f * sizeof(float) + 256 * (12 + 3 * d->n_layers);
}

// number of 32-row groups per clip when the fused clip-aligned GEMM applies (uniform batch,
// at most 128 pooled rows per clip), else 0
static int clip_tile_groups(const aware_batch* b) {
    if (!b->uniform_tp) return 0;
    const int g = (b->uniform_tp + 31) / 32;
    return (g >= 1 && g <= 4) ? g : 0;
}

// fewer clips than this: the per-clip mel front kernel would leave most CUs idle; the two-launch form wins
constexpr int kMelFrontMinClips = 192;
// fewer workgroups than this: the latency variant of the bf16x3 kernel serves the conv block (gemm_x3.hip, kSmallGrid)
constexpr int kH2MinGrid = 128;

static bool mel_front_applies(const aware_detector* d, const aware_batch* b, int pipe) {
    bool same_T = true;
    for (int i = 1; i < b->B; ++i) same_T = same_T && b->T[i] == b->T[0];
    return pipe != 1 && d->melTpk && same_T && b->B >= kMelFrontMinClips && mel_front_x3_supported(b->T[0], kFS, kFS);
}

// forward through the network; mag [NF][256] -> act[last], pred
// xm_ready: o.xm already holds the raw mel tile (the analysis kernel wrote it: mel projection folded in); mag is not read
static int det_forward(const aware_detector* d, const aware_batch* b, const float* mag, DetBufs& o, hipStream_t st,
                       int pipe = 0, bool skip_last = false, bool xm_ready = false) {
    // conv blocks of a uniform batch that fills the chip: the f16 two-term kernel (default pipe); its per-clip scale comes
    // from partial maxima that the producer of each operand leaves behind (x0_max: whether o.amax[0] is current)
    const int nwm0 = clip_tile_groups(b);
    auto h2_fwd = [&](int l) {
        return pipe == 0 && nwm0 && d->wh2[l] && d->ch[l + 1] >= 128 && (d->ch[l + 1] / 128) * b->B >= kH2MinGrid &&
               gemm_clip_h2_supported(nwm0, d->ch[l + 1], d->ch[l], d->ch[l]);
    };
    auto h2_rag = [&](int l) {
        return pipe == 0 && !nwm0 && d->wh2[l] && d->ch[l + 1] >= 128 && gemm_clip_h2_supported(1, d->ch[l + 1], d->ch[l], d->ch[l]) &&
               !(l == d->n_layers - 1 && d->ch[l + 1] <= 64);
    };
    bool cur_max = false;            // o.amax[l] holds the maxima of the current layer input
    if (xm_ready) {
        cur_max = launch_mel_norm_fwd(o.xm, b->d_frame_off, b->d_pool_off, o.x0, o.mstats, o.gstat, o.mpart, o.mstride, b->B,
                                      b->max_frames, st, (pipe == 0 && clip_tile_groups(b)) ? o.amax[0] : nullptr);
        LAUNCHCHK(); PROF(K_MELNORM);
    } else if (mel_front_applies(d, b, pipe)) {
        // uniform batch that fills the chip with one workgroup per clip: the whole mel block in one launch
        launch_mel_front_x3(mag, kFS, d->melTpk, b->d_frame_off, b->d_pool_off, o.xm, o.x0, o.mstats, o.gstat, b->B, b->T[0],
                            kFS, st, o.amax[0]);
        cur_max = true;
        LAUNCHCHK(); PROF(K_MELNORM);
    } else {
        gemm_plain(pipe, mag, kFS, d->melT, kFS, d->melTpk, nullptr, o.xm, 128, b->NF, 128, kFS, st);
        LAUNCHCHK(); PROF(K_GEMM);
        launch_mel_norm_fwd(o.xm, b->d_frame_off, b->d_pool_off, o.x0, o.mstats, o.gstat, o.mpart, o.mstride, b->B,
                            b->max_frames, st);
        LAUNCHCHK(); PROF(K_MELNORM);
    }
    const float* x = o.x0;
    const int nwm = clip_tile_groups(b);
    o.tail = 0;
    for (int l = 0; l < d->n_layers - (skip_last ? 1 : 0); ++l) {
        const int ci = d->ch[l], co = d->ch[l + 1];
        if (l == d->n_layers - 1 && co <= 64 && b->max_frames / 2 <= 320) {
            // skinny last conv: split-K partial slabs, summed inside the fused tail kernel
            launch_gemm_nt_splitk(x, ci, d->w[l], ci, o.zpart, co, b->NP, co, ci, kTailSplit, st);
            LAUNCHCHK(); PROF(K_GEMM);
            o.tail = 1;
        } else if (nwm && co >= 128) {
            // conv + InstanceNorm + LeakyReLU in one kernel (clip-aligned tiles)
            if (h2_fwd(l)) {
                const bool emit = skip_last && l == d->n_layers - 2;   // + split-K partials of the last conv
                if (!cur_max) { launch_clip_amax(x, ci, ci, 32 * nwm, b->B, o.amax[l], st); LAUNCHCHK(); PROF(K_MISC); }
                const bool next_h2 = l + 1 < d->n_layers && h2_fwd(l + 1);
                launch_gemm_clip_h2(x, ci, d->wh2[l], o.amax[l], next_h2 ? o.amax[l + 1] : nullptr, d->bias[l], o.act[l], co, b->B,
                                    nwm, b->uniform_tp, co, ci, 1, o.rstd[l], nullptr, st, emit ? d->lastpk : nullptr,
                                    emit ? o.zpart : nullptr, d->ch[d->n_layers]);
                cur_max = next_h2;
                if (emit) o.zslabs = co / gemm_clip_h2_slab_width(nwm, co, b->B);
                LAUNCHCHK(); PROF(K_GEMM_X3_FWD);
                x = o.act[l];
                continue;
            }
            cur_max = false;
            if (pipe != 1 && d->wpk[l] && gemm_clip_x3_supported(nwm, co, ci, ci)) {
                const bool emit = skip_last && l == d->n_layers - 2;   // + split-K partials of the last conv
                if (emit) o.zslabs = co / 128;
                launch_gemm_clip_x3(x, ci, d->wpk[l], d->bias[l], o.act[l], co, b->B, nwm, b->uniform_tp, co, ci, 1, o.rstd[l],
                                    nullptr, st, emit ? d->lastpk : nullptr, emit ? o.zpart : nullptr, d->ch[d->n_layers]);
                LAUNCHCHK(); PROF(K_GEMM_X3_FWD);
            } else {
                launch_gemm_clip(x, ci, d->w[l], ci, d->bias[l], o.act[l], co, b->B, nwm, b->uniform_tp, co, ci, 1, o.rstd[l],
                                 nullptr, st);
                LAUNCHCHK(); PROF(K_GEMM_CLIP_FWD);
            }
        } else if (h2_rag(l)) {
            // ragged batch / long clips on the default pipe: the f16 two-term kernel, clips walked in chunks of rows
            if (!cur_max) { launch_ragged_amax(x, ci, ci, b->d_frame_off, b->d_pool_off, b->B, o.amax[l], st); LAUNCHCHK(); PROF(K_MISC); }
            const bool next_h2 = l + 1 < d->n_layers && h2_rag(l + 1);
            launch_gemm_ragged_h2(x, ci, d->wh2[l], o.amax[l], next_h2 ? o.amax[l + 1] : nullptr, d->bias[l], o.act[l], co, b->B,
                                  b->d_frame_off, b->d_pool_off, b->d_order, co, ci, 1, o.rstd[l], nullptr, st);
            cur_max = next_h2;
            LAUNCHCHK(); PROF(K_GEMM_X3_FWD);
        } else if (!nwm && pipe != 1 && co >= 128 && d->wpk[l] && gemm_clip_x3_supported(1, co, ci, ci)) {
            // ragged batch / clips longer than the uniform kernel's tile: conv + InstanceNorm + LeakyReLU in one launch,
            // clips walked in chunks of rows (gemm_ragged_x3_kernel)
            launch_gemm_ragged_x3(x, ci, d->wpk[l], d->bias[l], o.act[l], co, b->B, b->d_frame_off, b->d_pool_off, b->d_order, co, ci, 1,
                                  o.rstd[l], nullptr, st);
            LAUNCHCHK(); PROF(K_GEMM_X3_FWD);
        } else {
            gemm_plain(pipe, x, ci, d->w[l], ci, d->wpk[l], d->bias[l], o.act[l], co, b->NP, co, ci, st);
            LAUNCHCHK(); PROF(K_GEMM);
            launch_in_lrelu_fwd(o.act[l], b->d_frame_off, b->d_pool_off, o.rstd[l], co, b->B, b->max_frames / 2, st);
            LAUNCHCHK(); PROF(K_INLRELU);
        }
        x = o.act[l];
    }
    return AWARE_OK;
}

extern "C" size_t aware_detect_workspace_bytes(const aware_batch* b, const aware_detector* d) {
    if (!b || !d) return 0;
    return det_bytes(b, d) + (size_t)b->NF * kFS * sizeof(float) + aware_batch_scratch_bytes(b) + 1024;
}

extern "C" int aware_detector_forward(const aware_detector* d, const aware_batch* b, const float* mag, float* values,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !b || !mag || !values || !workspace) return AWARE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    Carver c(workspace, workspace_bytes);
    DetBufs o;
    carve_det(c, b, d, o);
    if (!c.ok) return AWARE_E_WORKSPACE;
    int rc = det_forward(d, b, mag, o, st);
    if (rc) return rc;
    if (o.tail)
        launch_tail(o.zpart, kTailSplit, (size_t)b->NP * d->ch[d->n_layers], d->bias[d->n_layers - 1], b->d_frame_off,
                    b->d_pool_off, nullptr, values, nullptr, nullptr, nullptr, nullptr, nullptr, 0, d->nbits, b->B, b->max_frames / 2, st);
    else
        launch_head(o.act[d->n_layers - 1], b->d_frame_off, b->d_pool_off, nullptr, values, nullptr, nullptr, nullptr,
                    nullptr, nullptr, 0, d->nbits, b->B, st);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_detect(const aware_plan* plan, const aware_detector* d, const aware_batch* b, const float* audio,
                            float* values, void* workspace, size_t workspace_bytes, void* stream) {
    if (!plan || !d || !b || !audio || !values || !workspace) return AWARE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    Carver c(workspace, workspace_bytes);
    DetBufs o;
    carve_det(c, b, d, o);
    float* mag = c.take<float>((size_t)b->NF * kFS);
    unsigned long long* pmax = c.take<unsigned long long>((size_t)b->B * b->pstride);
    if (!c.ok) return AWARE_E_WORKSPACE;
    int rc = aware_stft_band(plan, b, audio, 1, mag, nullptr, pmax, stream);
    if (rc) return rc;
    rc = det_forward(d, b, mag, o, st);
    if (rc) return rc;
    if (o.tail)
        launch_tail(o.zpart, kTailSplit, (size_t)b->NP * d->ch[d->n_layers], d->bias[d->n_layers - 1], b->d_frame_off,
                    b->d_pool_off, nullptr, values, nullptr, nullptr, nullptr, nullptr, nullptr, 0, d->nbits, b->B, b->max_frames / 2, st);
    else
        launch_head(o.act[d->n_layers - 1], b->d_frame_off, b->d_pool_off, nullptr, values, nullptr, nullptr, nullptr,
                    nullptr, nullptr, 0, d->nbits, b->B, st);
    LAUNCHCHK();
    return AWARE_OK;
}

// Detector forward (multibit_detector_net.py:109-140), loss / gradient seed at the read-out, and the backward pass
// down to dL/d(band magnitudes) (data gradients only: the weights are frozen, multibit_embedder.py:76-77).
struct DetGradCtx {
    int pipe = 0, readout = 0;
    const float* target = nullptr;    // [B][n_bits]: bipolar watermark, or dL/dpred when loss_kind == AWARE_LOSS_EXTERNAL
    int loss_kind = 0;
    float* loss = nullptr;            // [B]
    float* best_loss = nullptr;       // [B] or null (no bookkeeping)
    int* improved = nullptr;
    int* step = nullptr;              // device step counter to advance, or null
    float *d1 = nullptr, *d2 = nullptr;   // gradient ping-pong [NP][maxc]
    float* gmag = nullptr;            // out: [NF][256]
    const float* loss_add = nullptr;  // [B] per-clip term added to the loss before the best-loss bookkeeping (L1 part), or null
    // EXTENSION (detector training): also the parameter gradients dL/dW_l [Cout][Cin], dL/db_l [Cout]; tr1 / tr2 are
    // scratch for the transposed operands, [maxc][NP] each.  Forces the three-kernel read-out (it writes dL/dZ of the
    // last block to memory).
    float* const* wgrad = nullptr;
    float* const* bgrad = nullptr;
    float *tr1 = nullptr, *tr2 = nullptr;
    // the caller expands dL/d(mel) (left in DetBufs::xm) to dL/d|S| itself (streaming synthesis adjoint, two taps per bin):
    // the K = 128 GEMM into gmag is skipped
    bool mel_grad_only = false;
    bool xm_ready = false;            // DetBufs::xm holds the raw mel tile on entry (det_forward)
};
static int det_forward_backward(const aware_detector* d, const aware_batch* b, const float* mag, DetBufs& db,
                                const DetGradCtx& G, hipStream_t st) {
    const int nl = d->n_layers;
    const int nwm = clip_tile_groups(b);
    // one kernel for the last conv block, the BRH head, the loss, their backward and the data gradient of the last
    // conv (uniform batches, bf16x3 configuration); otherwise split-K GEMM + tail kernel + data-gradient GEMM
    const int pipe = G.pipe;
    const bool fused_readout = G.readout == 0 && !G.wgrad && pipe != 1 && nwm && nl >= 2 && d->lastpk && G.target &&
                               readout_x3_supported(nwm, d->ch[nl - 1], d->ch[nl]) && d->wpk[nl - 2] &&
                               gemm_clip_x3_supported(nwm, d->ch[nl - 1], d->ch[nl - 2], d->ch[nl - 2]);
    int rc = det_forward(d, b, mag, db, st, pipe, fused_readout, G.xm_ready);
    if (rc) return rc;
    float* dA = G.d1;
    float* dB = G.d2;
    bool dz_ready = false;      // dA already holds dL/dZ of layer l (fused into the producing kernel)
    int l_top = nl - 1;         // first layer the backward loop below still has to differentiate
    bool last_k64 = false;      // dL/dZ of the last block was written with a row pitch of 64
    bool mel_bwd_done = false;
    const bool mel_fused = mel_front_applies(d, b, pipe);
    // data-gradient GEMMs of a uniform batch that fills the chip: the f16 two-term kernel; gA / gB travel with dA / dB and hold
    // the partial maxima of the gradient in them (g_cur: whether gA is current)
    float* gA = db.gmax[0];
    float* gB = db.gmax[1];
    bool g_cur = false;
    auto h2_bwd = [&](int l) {
        return pipe == 0 && nwm && l > 0 && d->wTh2[l] && d->ch[l] >= 128 && (d->ch[l] / 128) * b->B >= kH2MinGrid &&
               gemm_clip_h2_supported(nwm, d->ch[l], d->ch[l + 1], d->ch[l + 1]);
    };
    auto h2_rag_bwd = [&](int l) {
        return pipe == 0 && !nwm && l > 0 && l < nl && d->wTh2[l] && d->ch[l] >= 128 && d->ch[l + 1] >= 128 &&
               gemm_clip_h2_supported(1, d->ch[l], d->ch[l + 1], d->ch[l + 1]);
    };
    if (fused_readout) {
        launch_readout_x3(db.act[nl - 2], d->ch[nl - 1], db.zpart, db.zslabs, d->bias[nl - 1], d->lastTpk,
                          db.rstd[nl - 2], G.target, db.pred, G.loss, G.best_loss, G.improved, G.step, dA, b->B,
                          nwm, b->uniform_tp, d->ch[nl], d->nbits, G.loss_kind, st, G.loss_add, dB,
                          h2_bwd(nl - 2) ? gA : nullptr);
        g_cur = h2_bwd(nl - 2);
        // (dB, the other half of the gradient ping-pong, is free until the next data-gradient GEMM writes it: it holds the
        //  read-out's fragment image, readout_x3_image_bytes = 36 KB per 3 s clip, far below NP * maxc floats)
        dz_ready = true;
        l_top = nl - 2;
    } else if (db.tail) {
        // ragged batch on the bf16x3 pipe: dL/dZ of the last block with a pitch of 64 (zero K padding), so that its data
        // gradient runs on the ragged conv kernel with the previous block's InstanceNorm + LeakyReLU backward fused
        last_k64 = !nwm && pipe != 1 && !G.wgrad && nl >= 2 && d->lastTpk && d->ch[nl] <= 64 && d->ch[nl - 1] % 128 == 0;
        launch_tail(db.zpart, kTailSplit, (size_t)b->NP * d->ch[nl], d->bias[nl - 1], b->d_frame_off, b->d_pool_off,
                    G.target, db.pred, G.loss, G.best_loss, G.improved, dA, G.step, G.loss_kind, d->nbits, b->B,
                    b->max_frames / 2, st, G.loss_add, last_k64 ? 64 : 0);
        dz_ready = true;
    } else {
        launch_head(db.act[nl - 1], b->d_frame_off, b->d_pool_off, G.target, db.pred, G.loss, G.best_loss,
                    G.improved, dA, G.step, G.loss_kind, d->nbits, b->B, st, G.loss_add);
    }
    LAUNCHCHK(); PROF(K_HEAD);
    for (int l = l_top; l >= 0; --l) {
        const int ci = d->ch[l], co = d->ch[l + 1];
        if (!dz_ready) {
            launch_in_lrelu_bwd(dA, db.act[l], b->d_frame_off, b->d_pool_off, db.rstd[l], co, b->B, b->max_frames / 2, st);
            LAUNCHCHK(); PROF(K_INLRELU);
        }
        if (G.wgrad && G.wgrad[l]) {
            // dA = dL/dZ_l [NP][co] (zero in padding rows), X = input of the block [NP][ci]:  dW = dZ^T X as an NT GEMM over
            // the transposed operands (K = NP contiguous); db = column sums of dZ (zero up to rounding: the InstanceNorm
            // behind the convolution removes any per-channel constant)
            const float* X = l > 0 ? db.act[l - 1] : db.x0;
            launch_transpose(dA, G.tr1, b->NP, co, st);
            launch_transpose(X, G.tr2, b->NP, ci, st);
            launch_gemm_nt(G.tr1, b->NP, G.tr2, b->NP, nullptr, G.wgrad[l], ci, co, ci, b->NP, st);
            if (G.bgrad && G.bgrad[l]) launch_colsum(dA, G.bgrad[l], b->NP, co, st);
            LAUNCHCHK(); PROF(K_MISC);
        }
        if (nwm && l > 0 && ci >= 128) {
            // data-gradient GEMM whose epilogue is the backward of block l-1's InstanceNorm+LeakyReLU
            dz_ready = true;
            if (h2_bwd(l)) {
                if (!g_cur) { launch_clip_amax(dA, co, co, 32 * nwm, b->B, gA, st); LAUNCHCHK(); PROF(K_MISC); }
                const bool next_h2 = h2_bwd(l - 1);
                launch_gemm_clip_h2(dA, co, d->wTh2[l], gA, next_h2 ? gB : nullptr, nullptr, dB, ci, b->B, nwm, b->uniform_tp, ci, co, 2,
                                    db.rstd[l - 1], db.act[l - 1], st);
                g_cur = next_h2;
                LAUNCHCHK(); PROF(K_GEMM_X3_BWD);
                float* t = dA; dA = dB; dB = t;
                t = gA; gA = gB; gB = t;
                continue;
            }
            g_cur = false;
            if (pipe != 1 && d->wTpk[l] && gemm_clip_x3_supported(nwm, ci, co, co)) {
                launch_gemm_clip_x3(dA, co, d->wTpk[l], nullptr, dB, ci, b->B, nwm, b->uniform_tp, ci, co, 2,
                                    db.rstd[l - 1], db.act[l - 1], st);
                LAUNCHCHK(); PROF(K_GEMM_X3_BWD);
            } else {
                launch_gemm_clip(dA, co, d->wT[l], co, nullptr, dB, ci, b->B, nwm, b->uniform_tp, ci, co, 2, db.rstd[l - 1],
                                 db.act[l - 1], st);
                LAUNCHCHK(); PROF(K_GEMM_CLIP_BWD);
            }
        } else if (l == nl - 1 && last_k64) {
            dz_ready = true;
            const bool next_h2 = h2_rag_bwd(l - 1);
            launch_readout_grad_ragged_x3(db.act[l - 1], ci, dA, d->lastTpk, db.rstd[l - 1], dB, b->d_frame_off, b->d_pool_off,
                                          b->d_order, b->B, st, next_h2 ? gB : nullptr);
            g_cur = next_h2;
            LAUNCHCHK(); PROF(K_GEMM_X3_BWD);
            float* t2 = gA; gA = gB; gB = t2;
        } else if (h2_rag_bwd(l)) {
            // ragged batch: data-gradient GEMM + backward of block l-1's InstanceNorm + LeakyReLU in one launch (f16 two-term)
            dz_ready = true;
            if (!g_cur) { launch_ragged_amax(dA, co, co, b->d_frame_off, b->d_pool_off, b->B, gA, st); LAUNCHCHK(); PROF(K_MISC); }
            const bool next_h2 = h2_rag_bwd(l - 1);
            launch_gemm_ragged_h2(dA, co, d->wTh2[l], gA, next_h2 ? gB : nullptr, nullptr, dB, ci, b->B, b->d_frame_off, b->d_pool_off,
                                  b->d_order, ci, co, 2, db.rstd[l - 1], db.act[l - 1], st);
            g_cur = next_h2;
            LAUNCHCHK(); PROF(K_GEMM_X3_BWD);
            float* t2 = gA; gA = gB; gB = t2;
        } else if (!nwm && pipe != 1 && l > 0 && ci >= 128 && d->wTpk[l] && gemm_clip_x3_supported(1, ci, co, co)) {
            // ragged batch: data-gradient GEMM + backward of block l-1's InstanceNorm + LeakyReLU in one launch
            dz_ready = true;
            launch_gemm_ragged_x3(dA, co, d->wTpk[l], nullptr, dB, ci, b->B, b->d_frame_off, b->d_pool_off, b->d_order, ci, co, 2,
                                  db.rstd[l - 1], db.act[l - 1], st);
            LAUNCHCHK(); PROF(K_GEMM_X3_BWD);
        } else if (l == 0 && mel_fused && dz_ready && !G.wgrad && d->wTpk[0] && co % 64 == 0 && ci == 128) {
            // large uniform batch: block 0's data gradient and the backward of the mel block's normalisations in one launch
            launch_mel_back_x3(dA, co, d->wTpk[0], b->d_frame_off, b->d_pool_off, db.xm, db.mstats, db.gstat, b->B, b->T[0], co, st);
            LAUNCHCHK(); PROF(K_MELNORM);
            mel_bwd_done = true;
        } else {
            gemm_plain(pipe, dA, co, d->wT[l], co, d->wTpk[l], nullptr, dB, ci, b->NP, ci, co, st);
            dz_ready = false;
            LAUNCHCHK(); PROF(K_GEMM);
        }
        float* t = dA; dA = dB; dB = t;
    }
    if (!mel_bwd_done) {
        launch_mel_norm_bwd(dA, db.xm, b->d_frame_off, b->d_pool_off, db.mstats, db.gstat, db.mpart, db.mstride, b->B,
                            b->max_frames, st);
        LAUNCHCHK(); PROF(K_MELNORM);
    }
    if (!G.mel_grad_only) {
        gemm_plain(pipe, db.xm, 128, d->melB, 128, d->melBpk, nullptr, G.gmag, kFS, b->NF, kFS, 128, st);
        LAUNCHCHK(); PROF(K_GEMM);
    }
    return AWARE_OK;
}

// detector forward + backward for the differentiable plug-in seam: values = net(mag), grad_mag = (d values / d mag)^T grad_values
extern "C" size_t aware_detector_backward_workspace_bytes(const aware_batch* b, const aware_detector* d) {
    if (!b || !d) return 0;
    return det_bytes(b, d) + (size_t)b->NP * d->maxc * sizeof(float) * 2 + (size_t)b->B * 8 * sizeof(float) + 4096;
}
extern "C" int aware_detector_backward(const aware_detector* d, const aware_batch* b, const float* mag,
                                       const float* grad_values, float* values, float* grad_mag, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    if (!d || !b || !mag || !grad_values || !grad_mag || !workspace) return AWARE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    Carver c(workspace, workspace_bytes);
    DetBufs o;
    carve_det(c, b, d, o);
    DetGradCtx G;
    G.d1 = c.take<float>((size_t)b->NP * d->maxc);
    G.d2 = c.take<float>((size_t)b->NP * d->maxc);
    G.loss = c.take<float>(b->B);
    if (!c.ok) return AWARE_E_WORKSPACE;
    G.target = grad_values; G.loss_kind = AWARE_LOSS_EXTERNAL; G.gmag = grad_mag;
    int rc = det_forward_backward(d, b, mag, o, G, st);
    if (rc) return rc;
    if (values) HIPCHK(hipMemcpyAsync(values, o.pred, (size_t)b->B * d->nbits * sizeof(float), hipMemcpyDeviceToDevice, st));
    return AWARE_OK;
}

// EXTENSION (BASELINE north_star "gradients ... all-reduce"; the reference freezes the detector, multibit_embedder.py:76-77,
// and trains nothing): forward + backward of the network INCLUDING the parameter gradients, for a data-parallel detector
// training step -- the caller all-reduces grad_weights / grad_biases across ranks (RCCL) and applies its optimiser, then
// refreshes the device copy with aware_detector_update.  grad_weights / grad_biases: host arrays of n_layers device
// pointers ([Cout][Cin] and [Cout] f32; entries may be NULL).
extern "C" size_t aware_detector_train_workspace_bytes(const aware_batch* b, const aware_detector* d) {
    if (!b || !d) return 0;
    return aware_detector_backward_workspace_bytes(b, d) + (size_t)b->NP * d->maxc * sizeof(float) * 2 +
           (size_t)b->NF * kFS * sizeof(float) + 2048;
}
static int detector_train_core(const aware_detector* d, const aware_batch* b, const float* mag, const float* target, int loss_kind,
                               float* loss_out, float* values, float* grad_mag, float* const* grad_weights,
                               float* const* grad_biases, void* workspace, size_t workspace_bytes, void* stream);

// The same with the loss evaluated inside (ONE forward + backward per step): target [B][n_bits] bipolar, loss_kind an
// AWARE_LOSS_* of the embed loop (no best-loss bookkeeping), loss_out dev [B] per-clip losses.  The gradients are those of the
// SUM of the per-clip losses (scale by 1 / clips in the optimiser step for their mean).  grad_mag may be NULL.
extern "C" int aware_detector_train_gradients(const aware_detector* d, const aware_batch* b, const float* mag, const float* target,
                                              int loss_kind, float* loss_out, float* values, float* grad_mag,
                                              float* const* grad_weights, float* const* grad_biases, void* workspace,
                                              size_t workspace_bytes, void* stream) {
    if (!d || !b || !mag || !target || !loss_out || !grad_weights || !workspace) return AWARE_E_BADARG;
    if (loss_kind < 0 || loss_kind > AWARE_LOSS_BER) return AWARE_E_BADARG;
    return detector_train_core(d, b, mag, target, loss_kind, loss_out, values, grad_mag, grad_weights, grad_biases, workspace,
                               workspace_bytes, stream);
}

extern "C" int aware_detector_weight_gradients(const aware_detector* d, const aware_batch* b, const float* mag,
                                               const float* grad_values, float* values, float* grad_mag,
                                               float* const* grad_weights, float* const* grad_biases, void* workspace,
                                               size_t workspace_bytes, void* stream) {
    if (!d || !b || !mag || !grad_values || !grad_mag || !grad_weights || !workspace) return AWARE_E_BADARG;
    return detector_train_core(d, b, mag, grad_values, AWARE_LOSS_EXTERNAL, nullptr, values, grad_mag, grad_weights, grad_biases,
                               workspace, workspace_bytes, stream);
}

static int detector_train_core(const aware_detector* d, const aware_batch* b, const float* mag, const float* target, int loss_kind,
                               float* loss_out, float* values, float* grad_mag, float* const* grad_weights,
                               float* const* grad_biases, void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    Carver c(workspace, workspace_bytes);
    DetBufs o;
    carve_det(c, b, d, o);
    DetGradCtx G;
    G.d1 = c.take<float>((size_t)b->NP * d->maxc);
    G.d2 = c.take<float>((size_t)b->NP * d->maxc);
    G.tr1 = c.take<float>((size_t)b->NP * d->maxc);
    G.tr2 = c.take<float>((size_t)b->NP * d->maxc);
    G.loss = loss_out ? loss_out : c.take<float>(b->B);
    if (!grad_mag) grad_mag = c.take<float>((size_t)b->NF * kFS);
    if (!c.ok) return AWARE_E_WORKSPACE;
    // padding rows of the gradient ping-pong buffers take part in the row contraction: keep them finite
    HIPCHK(hipMemsetAsync(G.d1, 0, (size_t)b->NP * d->maxc * sizeof(float) * 2, st));
    G.target = target; G.loss_kind = loss_kind; G.gmag = grad_mag;
    G.wgrad = grad_weights; G.bgrad = grad_biases; G.readout = 1; G.pipe = 1;      // exact-f32 pipe for the training step
    int rc = det_forward_backward(d, b, mag, o, G, st);
    if (rc) return rc;
    if (values) HIPCHK(hipMemcpyAsync(values, o.pred, (size_t)b->B * d->nbits * sizeof(float), hipMemcpyDeviceToDevice, st));
    return AWARE_OK;
}

// ---------------------------------------------------------------------------------------------
struct aware_embed {
    const aware_plan* plan;
    const aware_detector* det;
    const aware_batch* b;
    aware_embed_config cfg;
    DetBufs db;
    // spectral state [NF][256]
    float *coef, *lo, *hi, *mom, *vel, *best, *mag, *gmag;
    cf *P, *U;
    // signals [NS]
    float *yraw, *oob, *gy;
    float* gpad;          // [B][2][512] reflect-pad parts of the synthesis adjoint (streaming DSP kernels)
    // original coefficients (the streaming analysis adjoint recomputes the box from them; also the L1 term's reference);
    // loss push_extremes + L1 (EXTENSION): per-run partial sums of |c - c0|, per-clip loss term
    float* c0 = nullptr;
    double* pl1 = nullptr;
    float* l1term = nullptr;
    // gradient ping-pong [NP][maxc]
    float *d1, *d2;
    // scalars
    float *loss, *best_loss, *target;
    int *improved, *step;
    float4* sched;
    unsigned long long *pmaxA, *pmaxY;
    double* pdot;
    float hyp[4];
    hipGraph_t graph = nullptr, graphN = nullptr;
    hipGraphExec_t gexec = nullptr, gexecN = nullptr;
    hipStream_t cap = nullptr;        // private stream used only to record the graph
    int steps_done = 0;               // optimiser steps since aware_embed_begin (host mirror of *step)
    // optimiser / scheduler other than the model card's (aware_embed_set_optimizer): the step runs as its own launch
    struct {
        bool active = false;
        int kind = 0, plateau = 0, patience = 0;
        float hyp[8] = {0};
        double wd = 0, factor = 0, threshold = 0, min_lr = 0, eps = 0, lr0 = 0;
        double* d_tab = nullptr;      // [num_iterations][5]
        double* d_lr = nullptr;       // [B] per-clip learning rate (plateau)
        double* d_state = nullptr;    // [B][3] plateau state
        std::vector<double> lr_init, state_init;
    } opt;
};

static size_t embed_bytes(const aware_batch* b, const aware_detector* d, int iters) {
    size_t bytes = det_bytes(b, d);
    bytes += (size_t)b->NF * kFS * sizeof(float) * 8;
    bytes += (size_t)b->NF * kFS * sizeof(cf) * 2;
    bytes += (size_t)b->NS * sizeof(float) * 3;
    bytes += (size_t)b->B * 1024 * sizeof(float);
    bytes += (size_t)b->NF * kFS * sizeof(float) + (size_t)b->B * b->pstride * 8 + (size_t)b->B * sizeof(float) + 1024;   // L1 term
    bytes += (size_t)b->NP * d->maxc * sizeof(float) * 2;
    bytes += (size_t)b->B * (3 * d->nbits + 8) * sizeof(float);
    bytes += (size_t)(iters + 1) * sizeof(float4) + (size_t)iters * 5 * sizeof(double) + (size_t)b->B * 4 * sizeof(double) + 1024;
    bytes += (size_t)b->B * b->pstride * 8 * 3;
    return bytes + 256 * 40;
}
extern "C" size_t aware_embed_workspace_bytes(const aware_batch* b, const aware_detector* d) {
    if (!b || !d) return 0;
    return embed_bytes(b, d, 4096);
}

// torch.optim.NAdam's per-step scalars (torch/optim/nadam.py _single_tensor_nadam): mu_product lives in a float32
// tensor and is read back with .item()
static void nadam_coefficients(int s, double lr, double b1, double b2, double md, float& mu_product, float out[3]) {
    const double bc2 = 1.0 - pow(b2, (double)s);
    const double mu = b1 * (1.0 - 0.5 * pow(0.96, s * md));
    const double mu_next = b1 * (1.0 - 0.5 * pow(0.96, (s + 1) * md));
    mu_product = mu_product * (float)mu;
    const double mp = (double)mu_product;
    out[0] = (float)(-lr * (1.0 - mu) / (1.0 - mp));
    out[1] = (float)((-lr * mu_next) / (1.0 - mp * mu_next));
    out[2] = (float)bc2;
}
extern "C" int aware_nadam_coefficients(int step, float lr, float beta1, float beta2, float momentum_decay,
                                        float* mu_product_io, float* coef3) {
    if (step < 1 || !mu_product_io || !coef3) return AWARE_E_BADARG;
    nadam_coefficients(step, lr, beta1, beta2, momentum_decay, *mu_product_io, coef3);
    return AWARE_OK;
}
extern "C" int aware_nadam_clamp_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* lo,
                                      const float* hi, size_t n, const float* coef3, float beta1, float beta2, float eps,
                                      void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !coef3 || n < 1) return AWARE_E_BADARG;
    launch_nadam_clamp(param, grad, exp_avg, exp_avg_sq, lo, hi, n, coef3[0], coef3[1], coef3[2], beta1, beta2, eps,
                       (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}
extern "C" int aware_opt_clamp_step(int kind, float* param, const float* grad, float* state1, float* state2, const float* lo,
                                    const float* hi, size_t n, const float* coef4, const float* hyp8, void* stream) {
    if (kind < 0 || kind > 7 || !param || !grad || !state1 || !state2 || !coef4 || !hyp8 || n < 1) return AWARE_E_BADARG;
    launch_opt_clamp(kind, param, grad, state1, state2, lo, hi, n, coef4, hyp8, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}
extern "C" int aware_waveform_normalize_bwd(const float* in, const float* grad_out, float* grad_in, const int* off,
                                            const int* len, int B, void* stream) {
    if (!in || !grad_out || !grad_in || !off || !len || B < 1) return AWARE_E_BADARG;
    launch_normalize_bwd(in, grad_out, grad_in, off, len, B, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}
extern "C" int aware_polar_decompose(const void* spec, float* mag, float* phase, size_t n, void* stream) {
    if (!spec || !mag || n < 1) return AWARE_E_BADARG;
    launch_polar_decompose(spec, mag, phase, n, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}
extern "C" int aware_polar_decompose_bwd(const void* spec, const float* grad_mag, const float* grad_phase, void* grad_spec,
                                         size_t n, void* stream) {
    if (!spec || !grad_spec || (!grad_mag && !grad_phase) || n < 1) return AWARE_E_BADARG;
    launch_polar_decompose_bwd(spec, grad_mag, grad_phase, grad_spec, n, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}
extern "C" int aware_polar_assemble(const float* mag, const float* phase, void* spec, size_t n, void* stream) {
    if (!mag || !phase || !spec || n < 1) return AWARE_E_BADARG;
    launch_polar_assemble(mag, phase, spec, n, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}
extern "C" int aware_polar_assemble_bwd(const float* mag, const float* phase, const void* grad_spec, float* grad_mag,
                                        float* grad_phase, size_t n, void* stream) {
    if (!mag || !phase || !grad_spec || (!grad_mag && !grad_phase) || n < 1) return AWARE_E_BADARG;
    launch_polar_assemble_bwd(mag, phase, grad_spec, grad_mag, grad_phase, n, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_embed_create(aware_embed** out, const aware_plan* plan, const aware_detector* det,
                                  const aware_batch* b, const aware_embed_config* cfg, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    if (!out || !plan || !det || !b || !cfg || !workspace) return AWARE_E_BADARG;
    if (cfg->num_iterations < 1 || cfg->num_iterations > 4096 || cfg->loss < 0 || cfg->loss > AWARE_LOSS_PUSH_L1) return AWARE_E_BADARG;
    if (cfg->conv_pipe < 0 || cfg->conv_pipe > 2 || cfg->readout < 0 || cfg->readout > 1 || cfg->mel < 0 || cfg->mel > 1)
        return AWARE_E_BADARG;
    if (cfg->dsp_path < 0 || cfg->dsp_path > 1) return AWARE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    aware_embed* e = new aware_embed();
    e->plan = plan; e->det = det; e->b = b; e->cfg = *cfg;
    Carver c(workspace, workspace_bytes);
    carve_det(c, b, det, e->db);
    const size_t nsp = (size_t)b->NF * kFS;
    e->coef = c.take<float>(nsp); e->lo = c.take<float>(nsp); e->hi = c.take<float>(nsp);
    e->mom = c.take<float>(nsp); e->vel = c.take<float>(nsp); e->best = c.take<float>(nsp);
    e->mag = c.take<float>(nsp); e->gmag = c.take<float>(nsp);
    e->P = c.take<cf>(nsp); e->U = c.take<cf>(nsp);
    e->yraw = c.take<float>(b->NS); e->oob = c.take<float>(b->NS); e->gy = c.take<float>(b->NS);
    e->gpad = c.take<float>((size_t)b->B * 1024);
    e->d1 = c.take<float>((size_t)b->NP * det->maxc); e->d2 = c.take<float>((size_t)b->NP * det->maxc);
    e->loss = c.take<float>(b->B); e->best_loss = c.take<float>(b->B);
    e->target = c.take<float>((size_t)b->B * det->nbits);
    e->improved = c.take<int>(b->B); e->step = c.take<int>(4);
    e->sched = c.take<float4>(cfg->num_iterations + 1);
    e->pmaxA = c.take<unsigned long long>((size_t)b->B * b->pstride);
    e->pmaxY = c.take<unsigned long long>((size_t)b->B * b->pstride);
    e->pdot = c.take<double>((size_t)b->B * b->pstride);
    e->c0 = c.take<float>(nsp);
    e->opt.d_tab = c.take<double>((size_t)cfg->num_iterations * 5);
    e->opt.d_lr = c.take<double>(b->B);
    e->opt.d_state = c.take<double>((size_t)b->B * 3);
    if (cfg->loss == AWARE_LOSS_PUSH_L1) {
        // the L1 term lives in the streaming DSP kernels only
        if (cfg->dsp_path != 0 || !stream_supported(plan->dev)) { delete e; return AWARE_E_UNSUPPORTED; }
        e->pl1 = c.take<double>((size_t)b->B * b->pstride);
        e->l1term = c.take<float>(b->B);
    }
    if (!c.ok) { delete e; return AWARE_E_WORKSPACE; }
    // columns nband..255 of every spectral row are padding: zeroed once here, never written by the loop kernels
    HIPCHK(hipMemsetAsync(e->mag, 0, nsp * sizeof(float), st));
    HIPCHK(hipMemsetAsync(e->U, 0, nsp * sizeof(cf), st));
    HIPCHK(hipMemsetAsync(e->gpad, 0, (size_t)b->B * 1024 * sizeof(float), st));
    std::vector<float4> sc(cfg->num_iterations + 1);
    float mu_product = 1.0f;
    const double b1 = cfg->beta1, b2 = cfg->beta2;
    for (int s = 1; s <= cfg->num_iterations; ++s) {
        float c3[3];
        nadam_coefficients(s, cfg->lr, b1, b2, cfg->momentum_decay, mu_product, c3);
        sc[s - 1] = make_float4(c3[0], c3[1], c3[2], 0.f);
    }
    sc[cfg->num_iterations] = sc[cfg->num_iterations - 1];
    HIPCHK(hipMemcpyAsync(e->sched, sc.data(), sc.size() * sizeof(float4), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    e->hyp[0] = (float)(1.0 - b1); e->hyp[1] = (float)b2; e->hyp[2] = (float)(1.0 - b2); e->hyp[3] = cfg->eps;
    // measured tile choice for the ten GEMM shapes of this batch (a few ms, once per geometry;
    // every configuration gives bit-identical results)
    {
        HIPCHK(hipMemsetAsync(e->d1, 0, (size_t)b->NP * det->maxc * sizeof(float), st));
        HIPCHK(hipMemsetAsync(e->mag, 0, nsp * sizeof(float), st));
        // only the shapes the f32 kernel will actually serve (the bf16x3 kernel takes M % 32 == 0, N % 128 == 0, K % 64 == 0)
        auto f32_shape = [&](int M, int N, int K, int lda) {
            return !(cfg->conv_pipe != 1 && M % 32 == 0 && gemm_clip_x3_supported(1, N, K, lda));
        };
        if (f32_shape(b->NF, 128, kFS, kFS)) gemm_autotune(e->mag, kFS, det->melT, kFS, e->db.xm, 128, b->NF, 128, kFS, st);
        if (f32_shape(b->NF, kFS, 128, 128)) gemm_autotune(e->db.xm, 128, det->melB, 128, e->gmag, kFS, b->NF, kFS, 128, st);
        for (int l = 0; l < det->n_layers; ++l) {
            const int ci = det->ch[l], co = det->ch[l + 1];
            if (f32_shape(b->NP, co, ci, ci)) gemm_autotune(e->d1, ci, det->w[l], ci, e->d2, co, b->NP, co, ci, st);
            if (f32_shape(b->NP, ci, co, co)) gemm_autotune(e->d1, co, det->wT[l], co, e->d2, ci, b->NP, ci, co, st);
        }
        HIPCHK(hipStreamSynchronize(st));
    }
    *out = e;
    return AWARE_OK;
}
// The third seam of the reference: optimiser / scheduler registries selected by YAML strings (embedding/optimizers.py:3-20,
// schedulers.py:3-16).  The host computes the per-step scalars of the chosen torch optimiser under the chosen learning-rate
// schedule (aware_amd/embedding/optimizers.py); the device applies them (opt_clamp_update, dsp_args.hpp).
extern "C" int aware_embed_set_optimizer(aware_embed* e, const aware_optimizer_config* oc, void* stream) {
    if (!e || !oc || oc->kind < 0 || oc->kind > 7 || !oc->table) return AWARE_E_BADARG;
    if (e->gexec) return AWARE_E_BADARG;                 // before the first aware_embed_iterate (the graphs are recorded then)
    if (oc->plateau && (!(oc->factor < 1.0) || oc->patience < 0)) return AWARE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int n = e->cfg.num_iterations, B = e->b->B;
    auto& o = e->opt;
    HIPCHK(hipMemcpyAsync(o.d_tab, oc->table, (size_t)n * 5 * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    o.kind = oc->kind; o.plateau = oc->plateau; o.patience = oc->patience;
    for (int i = 0; i < 8; ++i) o.hyp[i] = oc->hyp[i];
    o.wd = oc->weight_decay; o.factor = oc->factor; o.threshold = oc->threshold; o.min_lr = oc->min_lr; o.eps = oc->eps;
    o.lr0 = oc->lr0;
    o.lr_init.assign(B, oc->lr0);
    o.state_init.assign((size_t)B * 3, 0.0);
    for (int b = 0; b < B; ++b) o.state_init[3 * b] = (double)INFINITY;
    o.active = true;
    return AWARE_OK;
}

extern "C" void aware_embed_destroy(aware_embed* e) {
    if (!e) return;

    if (e->gexec) (void)hipGraphExecDestroy(e->gexec);
    if (e->graph) (void)hipGraphDestroy(e->graph);
    if (e->gexecN) (void)hipGraphExecDestroy(e->gexecN);
    if (e->graphN) (void)hipGraphDestroy(e->graphN);
    if (e->cap) (void)hipStreamDestroy(e->cap);
    delete e;
}

extern "C" void* aware_embed_buffer(aware_embed* e, int which) {
    if (!e) return nullptr;
    switch (which) {
        case 0: return e->loss;
        case 1: return e->best_loss;
        case 2: return e->db.pred;
        case 3: return e->coef;
        case 4: return e->best;
        case 5: return e->lo;
        case 6: return e->hi;
        case 7: return e->P;
        case 8: return e->step;
        case 9: return e->yraw;
        case 10: return e->mag;
        case 11: return e->opt.active ? e->opt.d_lr : nullptr;      // per-clip learning rate (double; ReduceLROnPlateau state)
        default: return nullptr;
    }
}

extern "C" int aware_embed_begin(aware_embed* e, const float* audio, const float* target, void* stream) {
    if (!e || !audio || !target) return AWARE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const aware_batch* b = e->b;
    // [WaveformNormalizer, STFT, STFTDecomposer] (multibit_embedder.py:143-147), band only
    launch_absmax_partials(audio, b->d_in_off, b->d_in_len, e->pmaxA, b->pstride, b->B, b->max_len, st);
    LAUNCHCHK();
    AnalysisLaunch L;
    L.plan = e->plan->dev; L.frame_off = b->d_frame_off; L.B = b->B; L.max_frames = b->max_frames; L.run_frames = b->an_run; L.wg_tab = b->d_an_wg; L.n_wg = b->n_an_wg;
    L.sig = audio; L.sig_off = b->d_in_off; L.sig_len = b->d_in_len;
    L.pmax = e->pmaxA; L.pcount = b->d_pc_in; L.pstride = b->pstride;
    L.mag = e->mag; L.unit = e->P; L.unit_default = 1.f;
    run_analysis(L, e->cfg.dsp_path, st);
    LAUNCHCHK();
    // constant out-of-band part of every synthesis: x/m - istft(band of the original)
    SynthLaunch S;
    S.plan = e->plan->dev; S.frame_off = b->d_frame_off; S.B = b->B; S.max_frames = b->max_frames; S.run_blocks = b->synth_run; S.wg_tab = b->d_syn_wg; S.n_wg = b->n_syn_wg;
    S.amp = e->mag; S.ph = e->P; S.out = e->gy; S.pstride = b->pstride;
    run_synth(S, e->cfg.dsp_path, st);
    LAUNCHCHK();
    launch_oob_residual(audio, b->d_in_off, e->pmaxA, b->d_pc_in, b->pstride, e->gy, b->d_frame_off, e->oob, b->B,
                        b->max_frames, st);
    LAUNCHCHK();
    const float ratio = (float)pow(10.0, -(double)e->cfg.tolerance_db / 20.0);
    launch_embed_prepare(e->mag, e->coef, e->lo, e->hi, e->mom, e->vel, e->best, e->c0, ratio, (size_t)b->NF * kFS, st);
    LAUNCHCHK();
    HIPCHK(hipMemcpyAsync(e->target, target, (size_t)b->B * e->det->nbits * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemsetAsync(e->step, 0, 4 * sizeof(int), st));
    // best_loss = +inf (0x7F800000)
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)e->best_loss, 0x7F800000, b->B, st));
    if (e->opt.active) {
        HIPCHK(hipMemcpyAsync(e->opt.d_lr, e->opt.lr_init.data(), (size_t)b->B * sizeof(double), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(e->opt.d_state, e->opt.state_init.data(), (size_t)b->B * 3 * sizeof(double), hipMemcpyHostToDevice, st));
    }
    e->steps_done = 0;
    return AWARE_OK;
}

// one loop body of AWAREEmbedder._optimize (multibit_embedder.py:95-122)
static int embed_iteration(aware_embed* e, hipStream_t st, int do_step, float* grad_out) {
    const aware_batch* b = e->b;
    const aware_detector* d = e->det;
    // :99-103  scatter + Assembler + ISTFT  (out-of-band part is the constant `oob`)
    SynthLaunch S;
    S.plan = e->plan->dev; S.frame_off = b->d_frame_off; S.B = b->B; S.max_frames = b->max_frames; S.run_blocks = b->synth_run; S.wg_tab = b->d_syn_wg; S.n_wg = b->n_syn_wg;
    S.amp = e->coef; S.ph = e->P; S.out = e->yraw; S.add = e->oob; S.pmax = e->pmaxY; S.pstride = b->pstride;
    S.c0 = e->pl1 ? e->c0 : nullptr; S.pl1 = e->pl1;
    const int dsp = e->cfg.dsp_path;
    run_synth(S, dsp, st);
    LAUNCHCHK(); PROF(K_SYNTH);
    if (e->pl1) {
        launch_l1_reduce(e->pl1, b->d_pc_syn, b->pstride, b->d_frame_off, e->plan->dev.nband, e->cfg.l1_weight, e->l1term, b->B, st);
        LAUNCHCHK(); PROF(K_MISC);
    }
    // normalise x2 + STFT + |.| on the band (:104 zeroes the rest, so it is never computed)
    AnalysisLaunch L;
    L.plan = e->plan->dev; L.frame_off = b->d_frame_off; L.B = b->B; L.max_frames = b->max_frames; L.run_frames = b->an_run; L.wg_tab = b->d_an_wg; L.n_wg = b->n_an_wg;
    L.sig = e->yraw; L.sig_off = b->d_out_off; L.sig_len = b->d_out_len;
    L.pmax = e->pmaxY; L.pcount = b->d_pc_syn; L.pstride = b->pstride; L.double_norm = 1;
    L.mag = e->mag; L.unit = e->U; L.unit_default = 0.f; L.write_pad = 0;
    // the mel projection as short runs of adjacent bins inside the streaming analysis kernel (no magnitude array), and its
    // backward as two taps per bin inside the synthesis adjoint (below): streaming DSP path and a filter bank of that form
    const bool mel_taps = dsp == 0 && d->melw && d->melm && stream_supported(e->plan->dev) && e->cfg.mel == 0;
    const bool mel_fold = mel_taps && d->melf_w && d->melf_s && d->n_mels == 128;
    if (mel_fold) { L.mel_out = e->db.xm; L.melf_w = d->melf_w; L.melf_s = d->melf_s; }
    run_analysis(L, dsp, st);
    LAUNCHCHK(); PROF(K_ANALYSIS);
    // :107 detector forward, :109 loss, :120-122 best tracking, :111 backward through the detector
    DetGradCtx G;
    G.pipe = e->cfg.conv_pipe; G.readout = e->cfg.readout; G.target = e->target; G.loss_kind = e->cfg.loss;
    G.loss = e->loss; G.d1 = e->d1; G.d2 = e->d2; G.gmag = e->gmag; G.loss_add = e->l1term;
    G.step = do_step ? e->step : nullptr;               // the read-out kernel advances the step counter
    // aware_embed_gradient (do_step == 0) leaves the best-loss bookkeeping alone: the reference snapshots only
    // inside the optimiser loop (multibit_embedder.py:120-122)
    G.best_loss = do_step ? e->best_loss : nullptr;
    G.improved = do_step ? e->improved : nullptr;
    G.mel_grad_only = mel_taps;
    G.xm_ready = mel_fold;
    int rc = det_forward_backward(d, b, e->mag, e->db, G, st);
    if (rc) return rc;
    // backward through |.|, STFT, reflect padding
    SynthLaunch SA;
    SA.plan = e->plan->dev; SA.frame_off = b->d_frame_off; SA.B = b->B; SA.max_frames = b->max_frames; SA.run_blocks = b->synth_run; SA.wg_tab = b->d_syn_wg; SA.n_wg = b->n_syn_wg;
    SA.amp = e->gmag; SA.ph = e->U; SA.out = e->gy; SA.adjoint = 1; SA.yraw = e->yraw; SA.pmax_in = e->pmaxY;
    SA.pcount = b->d_pc_syn; SA.pdot = e->pdot; SA.pstride = b->pstride; SA.gpad = e->gpad;
    if (mel_taps) { SA.dmel = e->db.xm; SA.melw = d->melw; SA.melm = d->melm; }
    run_synth(SA, dsp, st);
    LAUNCHCHK(); PROF(K_SYNTH_ADJ);
    // backward through the normalisers, ISTFT and the assembler; :112-117 NAdam + clamp
    AnalysisLaunch LA;
    LA.plan = e->plan->dev; LA.frame_off = b->d_frame_off; LA.B = b->B; LA.max_frames = b->max_frames; LA.run_frames = b->an_run; LA.wg_tab = b->d_an_wg; LA.n_wg = b->n_an_wg;
    LA.sig = e->gy; LA.sig_off = b->d_out_off; LA.sig_len = b->d_out_len;
    LA.pmax = e->pmaxY; LA.pcount = b->d_pc_syn; LA.pstride = b->pstride;
    LA.adjoint = 1; LA.yraw = e->yraw; LA.pdot = e->pdot; LA.phasor = e->P;
    LA.coef = e->coef; LA.mom = e->mom; LA.vel = e->vel; LA.lo = e->lo; LA.hi = e->hi; LA.best = e->best;
    LA.improved = e->improved; LA.sched = e->sched; LA.sched_len = e->cfg.num_iterations + 1; LA.step = e->step;
    // (an optimiser / schedule set by aware_embed_set_optimizer steps in its own launch below: the adjoint then only
    //  delivers the gradient, into gmag -- consumed by the synthesis adjoint by now)
    const bool own_step = do_step && e->opt.active;
    LA.grad_out = own_step ? e->gmag : grad_out; LA.do_step = own_step ? 0 : do_step;
    memcpy(LA.hyp, e->hyp, sizeof(LA.hyp));
    LA.gpad = e->gpad; LA.c0 = e->c0; LA.box_ratio = (float)pow(10.0, -(double)e->cfg.tolerance_db / 20.0);
    LA.l1_weight = e->pl1 ? e->cfg.l1_weight : 0.f;
    run_analysis(LA, dsp, st);
    LAUNCHCHK(); PROF(K_ANALYSIS_ADJ);
    if (own_step) {
        const auto& o = e->opt;
        launch_opt_rows(o.kind, e->coef, e->gmag, e->mom, e->vel, e->c0, LA.box_ratio, e->best, e->improved, b->d_frame_off, b->B,
                        b->NF, o.d_tab, e->cfg.num_iterations, e->step, o.plateau ? o.d_lr : nullptr, o.wd, o.hyp,
                        e->plan->dev.nband, st);
        if (o.plateau)
            launch_plateau(e->loss, o.d_state, o.d_lr, b->B, o.factor, o.patience, o.threshold, o.min_lr, o.eps, st);
        LAUNCHCHK(); PROF(K_MISC);
    }
    return AWARE_OK;
}

extern "C" int aware_embed_iterate(aware_embed* e, int n_iters, void* stream) {
    if (!e || n_iters < 0) return AWARE_E_BADARG;
    // the NAdam schedule table holds cfg.num_iterations steps (the reference's loop runs exactly that many,
    // multibit_embedder.py:95): more than that since aware_embed_begin is a caller error
    if (e->steps_done + n_iters > e->cfg.num_iterations) return AWARE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (!e->cfg.use_graph) {
        for (int i = 0; i < n_iters; ++i) {
            int rc = embed_iteration(e, st, 1, nullptr);
            if (rc) return rc;
            ++e->steps_done;
        }
        return AWARE_OK;
    }
    // Two graphs: one loop body, and kGraphIters loop bodies (amortises the replay floor).  The
    // caller's stream may be the legacy default stream, which cannot be captured: record on a
    // private stream, replay on the caller's.  All kernel arguments are iteration-invariant
    // (the optimiser step index lives in device memory), so one recording serves every replay.
    constexpr int kGraphIters = 16;
    if (!e->gexec) {
        if (!e->cap) HIPCHK(hipStreamCreateWithFlags(&e->cap, hipStreamNonBlocking));
        for (int which = 0; which < 2; ++which) {
            HIPCHK(hipStreamBeginCapture(e->cap, hipStreamCaptureModeThreadLocal));
            int rc = AWARE_OK;
            for (int i = 0; i < (which ? kGraphIters : 1) && rc == AWARE_OK; ++i) rc = embed_iteration(e, e->cap, 1, nullptr);
            hipGraph_t g = nullptr;
            hipError_t ce = hipStreamEndCapture(e->cap, &g);
            hipGraphExec_t ge = nullptr;
            if (rc == AWARE_OK && ce == hipSuccess) ce = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            if (rc != AWARE_OK || ce != hipSuccess) {
                // no half-recorded state: drop whatever exists (both graphs) so that a later call records again
                if (ge) (void)hipGraphExecDestroy(ge);
                if (g) (void)hipGraphDestroy(g);
                if (e->gexec) { (void)hipGraphExecDestroy(e->gexec); e->gexec = nullptr; }
                if (e->graph) { (void)hipGraphDestroy(e->graph); e->graph = nullptr; }
                if (rc != AWARE_OK) return rc;
                HIPCHK(ce);
            }
            if (which) { e->graphN = g; e->gexecN = ge; } else { e->graph = g; e->gexec = ge; }
        }
    }
    int left = n_iters;
    for (; left >= kGraphIters; left -= kGraphIters) { HIPCHK(hipGraphLaunch(e->gexecN, st)); e->steps_done += kGraphIters; }
    for (; left > 0; --left) { HIPCHK(hipGraphLaunch(e->gexec, st)); ++e->steps_done; }
    return AWARE_OK;
}

extern "C" int aware_embed_profile(aware_embed* e, int n_iters, int max_entries, float* ms_out, int* kind_out,
                                   void* stream) {
    if (!e || !ms_out || !kind_out || n_iters < 1) return AWARE_E_BADARG;
    if (e->steps_done + n_iters > e->cfg.num_iterations) return AWARE_E_BADARG;      // as aware_embed_iterate
    hipStream_t st = (hipStream_t)stream;
    LaunchProfiler p;
    p.st = st;
    g_prof = &p;
    prof_mark(-1);
    int rc = AWARE_OK;
    for (int i = 0; i < n_iters && rc == AWARE_OK; ++i) { rc = embed_iteration(e, st, 1, nullptr); if (rc == AWARE_OK) ++e->steps_done; }
    g_prof = nullptr;
    hipError_t se = hipStreamSynchronize(st);
    int n = 0;
    for (size_t i = 1; i < p.ev.size(); ++i) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, p.ev[i - 1], p.ev[i]);
        if (n < max_entries) { ms_out[n] = ms; kind_out[n] = p.kind[i]; ++n; }
    }
    for (auto& ev : p.ev) (void)hipEventDestroy(ev);
    if (rc) return rc;
    HIPCHK(se);
    return n;      // number of entries written (>= 0)
}

extern "C" int aware_embed_gradient(aware_embed* e, float* grad, void* stream) {
    if (!e || !grad) return AWARE_E_BADARG;
    return embed_iteration(e, (hipStream_t)stream, 0, grad);
}

extern "C" int aware_embed_finish(aware_embed* e, const float* rescale, float* out, void* stream) {
    if (!e || !out) return AWARE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const aware_batch* b = e->b;
    SynthLaunch S;
    S.plan = e->plan->dev; S.frame_off = b->d_frame_off; S.B = b->B; S.max_frames = b->max_frames; S.run_blocks = b->synth_run; S.wg_tab = b->d_syn_wg; S.n_wg = b->n_syn_wg;
    S.amp = e->best; S.ph = e->P; S.out = e->yraw; S.add = e->oob; S.pmax = e->pmaxY; S.pstride = b->pstride;
    run_synth(S, e->cfg.dsp_path, st);
    LAUNCHCHK();
    launch_finish(e->yraw, b->d_frame_off, e->pmaxY, b->d_pc_syn, b->pstride, rescale, out, b->d_out_off, b->B,
                  b->max_frames, st);
    LAUNCHCHK();
    return AWARE_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" int aware_pcm_quantize(const float* in, float* out, const int* off, const int* len, int B, int max_len,
                                  int bits, void* scratch, void* stream) {
    if (!in || !out || !off || !len || !scratch || B < 1) return AWARE_E_BADARG;
    float q, lo, hi;
    switch (bits) {   // scripts/attacks.py:52-67 (the "12-bit" branch really is 13-bit)
        case 8: q = 127.f; lo = -128.f; hi = 127.f; break;
        case 12: q = 4095.f; lo = -4096.f; hi = 4095.f; break;
        case 16: q = 32767.f; lo = -32768.f; hi = 32767.f; break;
        case 24: q = 8388607.f; lo = -8388608.f; hi = 8388607.f; break;
        default: return AWARE_E_BADARG;   // reference raises ValueError
    }
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* pmax; int* pcount; int ps;
    int rc = absmax_into_scratch(in, off, len, B, max_len, scratch, &pmax, &pcount, &ps, st);
    if (rc) return rc;
    launch_pcm_quantize(in, out, off, len, pmax, pcount, ps, q, lo, hi, B, max_len, st);
    LAUNCHCHK();
    return AWARE_OK;
}

// shared by the max-abs normalising entry points: partial maxima + counts in `scratch`
static int absmax_into_scratch(const float* in, const int* off, const int* len, int B, int max_len, void* scratch,
                               unsigned long long** pmax_out, int** pcount_out, int* ps_out, hipStream_t st) {
    const int ps = (max_len + 4095) / 4096;
    unsigned long long* pmax = (unsigned long long*)scratch;
    int* pcount = (int*)(pmax + (size_t)B * ps);
    HIPCHK(hipMemsetAsync(pmax, 0, (size_t)B * ps * sizeof(unsigned long long), st));
    launch_absmax_partials(in, off, len, pmax, ps, B, max_len, st);
    LAUNCHCHK();
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)pcount, ps, B, st));
    *pmax_out = pmax; *pcount_out = pcount; *ps_out = ps;
    return AWARE_OK;
}

extern "C" int aware_waveform_normalize(const float* in, float* out, const int* off, const int* len, int B,
                                        int max_len, void* scratch, void* stream) {
    if (!in || !out || !off || !len || !scratch || B < 1) return AWARE_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* pmax; int* pcount; int ps;
    int rc = absmax_into_scratch(in, off, len, B, max_len, scratch, &pmax, &pcount, &ps, st);
    if (rc) return rc;
    launch_normalize(in, out, off, len, pmax, pcount, ps, B, max_len, st);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_upfirdn(const float* in, const int* in_off, const int* in_len, float* out, const int* out_off,
                             const int* out_len, int B, int max_out, const float* h, int nh, int up, int down,
                             int half_len, void* stream) {
    if (!in || !out || !h || B < 1 || up < 1 || down < 1) return AWARE_E_BADARG;
    launch_upfirdn(in, in_off, in_len, out, out_off, out_len, h, nh, up, down, half_len, B, max_out,
                   (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_iir(const float* in, const int* off, const int* len, int B, int max_len, void* out, int out_f64,
                         const double* b, const double* a, const double* zi, int ncoef, int filtfilt, void* scratch,
                         void* stream) {
    if (!in || !out || !b || !a || B < 1 || ncoef < 2 || ncoef > 12) return AWARE_E_BADARG;
    if (filtfilt && (!zi || !scratch)) return AWARE_E_BADARG;
    launch_iir_full(in, off, len, out, out_f64, b, a, zi, ncoef, filtfilt ? 1 : 0, (double*)scratch,
                    max_len + 6 * ncoef, B, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_decimate_interp(const float* in, const int* off, const int* len, int B, int max_len, int factor,
                                     double* out, void* stream) {
    if (!in || !off || !len || !out || B < 1 || factor < 2) return AWARE_E_BADARG;
    launch_decimate_interp(in, off, len, out, factor, B, max_len, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_segment_cut(const float* in, const int* in_off, float* out, const int* out_off,
                                 const int* out_len, const int* cut_start, const int* cut_len, int zero_fill, int B,
                                 int max_len, void* stream) {
    if (!in || !out || B < 1) return AWARE_E_BADARG;
    launch_segment_copy(in, in_off, out, out_off, out_len, cut_start, cut_len, zero_fill, B, max_len,
                        (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_gaussian_noise(const float* in, float* out, const int* off, const int* len, int B, int max_len,
                                    const uint32_t* seeds, float snr_db, void* scratch, void* stream) {
    if (!in || !out || !seeds || !scratch || B < 1) return AWARE_E_BADARG;
    launch_gaussian_noise_full(in, out, off, len, seeds, (double*)scratch, snr_db, B, max_len, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_snr(const float* output, const int* out_offsets, const float* target, const int* tgt_offsets,
                         const int* lengths, int B, double* snr_db, void* stream) {
    if (!output || !out_offsets || !target || !tgt_offsets || !lengths || !snr_db || B < 1) return AWARE_E_BADARG;
    launch_snr(output, out_offsets, target, tgt_offsets, lengths, snr_db, B, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_phase_vocoder(const void* spec_in, const int* frame_off_in, void* spec_out, const int* frame_off_out,
                                   int B, double rate, void* stream) {
    if (!spec_in || !frame_off_in || !spec_out || !frame_off_out || B < 1 || !(rate > 0.0)) return AWARE_E_BADARG;
    launch_phase_vocoder(spec_in, frame_off_in, spec_out, frame_off_out, rate, B, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_spectral_quantize(void* spec, int n_frames, float step_db, float floor_db, void* stream) {
    if (!spec || n_frames < 1 || !(step_db > 0.f)) return AWARE_E_BADARG;
    launch_spectral_quantize(spec, n_frames, step_db, floor_db, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_spectral_quantize_bwd(const void* spec_in, const void* grad_out, void* grad_in, int n_frames, float step_db,
                                           float floor_db, void* stream) {
    if (!spec_in || !grad_out || !grad_in || n_frames < 1 || !(step_db > 0.f)) return AWARE_E_BADARG;
    launch_spectral_quantize_bwd(spec_in, grad_out, grad_in, n_frames, step_db, floor_db, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_gemm_nt_variant(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C,
                                     int ldc, int M, int N, int K, int variant, void* stream) {
    if (!A || !Bt || !C || M < 1 || N < 1 || K < 4 || (K & 3) || (lda & 3) || (ldb & 3) || variant < 0 || variant > 16)
        return AWARE_E_BADARG;
    launch_gemm_nt_variant(A, lda, Bt, ldb, bias, C, ldc, M, N, K, variant, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}

// clip-aligned GEMM alone (tests / roofline): mode 0 = f32 MFMA kernel on Bt, mode 1 = bf16 three-way split kernel on Bpk
extern "C" size_t aware_x3_packed_bytes(int N, int K) { return (N > 0 && K > 0 && N % 16 == 0) ? x3_packed_bytes(N, K) : 0; }
extern "C" int aware_x3_pack(const float* host_wt, int N, int K, void* host_out) {
    if (!host_wt || !host_out || N < 16 || N % 16 || K < 1) return AWARE_E_BADARG;
    x3_pack(host_wt, N, K, (uint16_t*)host_out);
    return AWARE_OK;
}
extern "C" int aware_gemm_clip(const float* A, int lda, const float* Bt, int ldb, const void* Bpk, const float* bias, float* C,
                               int ldc, int B, int Tp, int N, int K, int epi, float* rstd_io, const float* act, int mode,
                               void* stream) {
    if (!A || !C || B < 1 || Tp < 1 || Tp > 128 || N < 1 || K < 4 || (K & 3) || (lda & 3) || epi < 0 || epi > 2) return AWARE_E_BADARG;
    if (epi != 0 && !rstd_io) return AWARE_E_BADARG;
    if (epi == 2 && !act) return AWARE_E_BADARG;
    const int nwm = (Tp + 31) / 32;
    if (mode == 1) {
        if (!Bpk || !gemm_clip_x3_supported(nwm, N, K, lda)) return AWARE_E_BADARG;
        launch_gemm_clip_x3(A, lda, Bpk, bias, C, ldc, B, nwm, Tp, N, K, epi, rstd_io, act, (hipStream_t)stream);
    } else {
        if (!Bt || (ldb & 3)) return AWARE_E_BADARG;
        launch_gemm_clip(A, lda, Bt, ldb, bias, C, ldc, B, nwm, Tp, N, K, epi, rstd_io, act, (hipStream_t)stream);
    }
    LAUNCHCHK();
    return AWARE_OK;
}

// the forward conv block whose epilogue also emits the split-K partials of the NEXT (skinny, CL <= 48 channels) conv:
// the kernel the embed loop runs for block 2 (X3_FWD_LAST).  Test / roofline entry; always the throughput kernel.
extern "C" int aware_gemm_clip_last(const float* A, int lda, const void* Bpk, const float* bias, float* C, int ldc, int B, int Tp,
                                    int N, int K, float* rstd_out, const void* lastpk, float* zpart, int CL, void* stream) {
    if (!A || !Bpk || !C || !rstd_out || !lastpk || !zpart || B < 1 || Tp < 1 || Tp > 128 || CL < 2 || CL > 48) return AWARE_E_BADARG;
    const int nwm = (Tp + 31) / 32;
    if (!gemm_clip_x3_supported(nwm, N, K, lda)) return AWARE_E_BADARG;
    launch_gemm_clip_x3(A, lda, Bpk, bias, C, ldc, B, nwm, Tp, N, K, 1, rstd_out, nullptr, (hipStream_t)stream, lastpk, zpart, CL);
    LAUNCHCHK();
    return AWARE_OK;
}

// the same block(s) on the f16 two-term kernel (gemm_h2.hip), test / roofline entry: packs Bt (device, [N][K]) and computes
// the per-clip maxima of A into `workspace` first.  epi 0..2 as aware_gemm_clip; lastpk / zpart / CL (epi 1 only, may be
// null / 0) as aware_gemm_clip_last.
extern "C" size_t aware_gemm_clip_h2_workspace_bytes(int B, int N, int K) {
    return (B > 0 && N > 0 && K > 0) ? h2_packed_bytes(N, K) + (size_t)B * 64 * sizeof(float) * 2 + 1024 : 0;
}
extern "C" int aware_gemm_clip_h2(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C, int ldc, int B,
                                  int Tp, int N, int K, int epi, float* rstd_io, const float* act, const void* lastpk, float* zpart,
                                  int CL, float* amax_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !Bt || !C || !workspace || B < 1 || Tp < 1 || Tp > 128 || epi < 0 || epi > 2 || (ldb & 3)) return AWARE_E_BADARG;
    if (epi != 0 && !rstd_io) return AWARE_E_BADARG;
    if (epi == 2 && !act) return AWARE_E_BADARG;
    const int nwm = (Tp + 31) / 32;
    if (!gemm_clip_h2_supported(nwm, N, K, lda) || N % 16) return AWARE_E_BADARG;
    if (workspace_bytes < aware_gemm_clip_h2_workspace_bytes(B, N, K)) return AWARE_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Carver c(workspace, workspace_bytes);
    void* pk = c.take<char>(h2_packed_bytes(N, K));
    float* amax = c.take<float>((size_t)B * 64);
    launch_h2_pack(Bt, ldb, N, K, pk, st);
    launch_clip_amax(A, lda, K, 32 * nwm, B, amax, st);
    launch_gemm_clip_h2(A, lda, pk, amax, amax_out, bias, C, ldc, B, nwm, Tp, N, K, epi, rstd_io, act, st,
                        (epi == 1 && lastpk && zpart) ? lastpk : nullptr, zpart, CL);
    LAUNCHCHK();
    return AWARE_OK;
}

extern "C" int aware_gemm_nt(const float* A, int lda, const float* Bt, int ldb, const float* bias, float* C, int ldc,
                             int M, int N, int K, void* stream) {
    if (!A || !Bt || !C || M < 1 || N < 1 || K < 4 || (K & 3) || (lda & 3) || (ldb & 3)) return AWARE_E_BADARG;
    launch_gemm_nt(A, lda, Bt, ldb, bias, C, ldc, M, N, K, (hipStream_t)stream);
    LAUNCHCHK();
    return AWARE_OK;
}
