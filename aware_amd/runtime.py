"""Object wrappers over the C ABI: plan, batch geometry, detector weights, embed session.

PyTorch-ROCm is used for device memory and streams only; every computation below is a
call into libaware_hip.so on `torch.cuda.current_stream()`.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np
import torch

from . import _lib
from ._lib import AwareHipError, EmbedConfig, check, load_library, require_gpu

SPEC_STRIDE = 256
FULL_STRIDE = 520
CONV_PIPES = {"f16x2": 0, "f32": 1, "bf16x3": 2}
LOSS_KINDS = {"push_extremes": 0, "mse": 1, "hinge": 2, "sign": 3, "push_sigmoid": 4, "ber": 5, "push_extremes_l1": 6}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


class Plan:
    """FFT tables + window + embedding band (aware_plan)."""

    def __init__(self, n_fft=1024, hop=256, win_length=1024, window="hann", band_bins=(32, 256)):
        require_gpu()
        self.lib = load_library()
        wid = {"hann": 0, "hamming": 1}.get(window)
        if wid is None:
            raise ValueError(f"Invalid window type: {window}")       # utils/audio/stft.py:25
        h = C.c_void_p()
        check(self.lib.aware_plan_create(C.byref(h), n_fft, hop, win_length, wid, int(band_bins[0]), int(band_bins[1])),
              "aware_plan_create")
        self.h = h
        self.n_fft, self.hop, self.band_bins = n_fft, hop, (int(band_bins[0]), int(band_bins[1]))
        self.nband = self.band_bins[1] - self.band_bins[0] + 1

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.aware_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass


class Batch:
    """Ragged batch geometry (aware_batch)."""

    def __init__(self, lengths: Sequence[int], in_offsets: Sequence[int] | None = None):
        require_gpu()
        self.lib = load_library()
        self.lengths = [int(x) for x in lengths]
        self.B = len(self.lengths)
        arr = (C.c_int * self.B)(*self.lengths)
        if in_offsets is None:
            self.in_offsets = np.concatenate([[0], np.cumsum(self.lengths)[:-1]]).astype(np.int64).tolist()
            off = None
        else:
            self.in_offsets = [int(x) for x in in_offsets]
            off = (C.c_int * self.B)(*self.in_offsets)
        h = C.c_void_p()
        rc = self.lib.aware_batch_create(C.byref(h), self.B, arr, off)
        if rc == -1:
            raise ValueError("every clip needs more than n_fft/2 = 512 samples")
        check(rc, "aware_batch_create")
        self.h = h
        self.total_frames = self.lib.aware_batch_total_frames(h)
        self.total_pooled = self.lib.aware_batch_total_pooled(h)
        self.total_out = self.lib.aware_batch_total_out(h)
        self.frames = [self.lib.aware_batch_frames(h, i) for i in range(self.B)]
        self.out_offsets = [self.lib.aware_batch_out_offset(h, i) for i in range(self.B)]
        self.out_lengths = [self.lib.aware_batch_out_length(h, i) for i in range(self.B)]
        self.frame_offsets = np.concatenate([[0], np.cumsum(self.frames)]).tolist()
        self.total_in = self.in_offsets[-1] + self.lengths[-1]
        self.scratch_bytes = self.lib.aware_batch_scratch_bytes(h)
        self.uniform = len(set(self.lengths)) == 1 and in_offsets is None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.aware_batch_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def scratch(self):
        return torch.empty(self.scratch_bytes, dtype=torch.uint8, device=_dev())

    # ---- ragged <-> list helpers (plumbing) ----
    def pack(self, clips) -> torch.Tensor:
        """list of 1-D float arrays/tensors -> one device tensor laid out at in_offsets."""
        out = torch.zeros(self.total_in, dtype=torch.float32, device=_dev())
        for c, o, n in zip(clips, self.in_offsets, self.lengths):
            out[o:o + n] = torch.as_tensor(c, dtype=torch.float32)
        return out

    def unpack_out(self, flat: torch.Tensor):
        return [flat[o:o + n] for o, n in zip(self.out_offsets, self.out_lengths)]


def stft(plan: Plan, batch: Batch, audio: torch.Tensor, normalize=False) -> torch.Tensor:
    """Full one-sided spectrum, frame-major [total_frames, 520] complex64 (bins 0..512 valid)."""
    spec = torch.empty((batch.total_frames, FULL_STRIDE), dtype=torch.complex64, device=audio.device)
    scr = batch.scratch()
    check(plan.lib.aware_stft(plan.h, batch.h, _ptr(audio), int(normalize), _ptr(spec), _ptr(scr), _stream()), "aware_stft")
    return spec


def istft(plan: Plan, batch: Batch, spec: torch.Tensor, normalize=False) -> torch.Tensor:
    out = torch.empty(batch.total_out, dtype=torch.float32, device=spec.device)
    scr = batch.scratch()
    check(plan.lib.aware_istft(plan.h, batch.h, _ptr(spec), int(normalize), _ptr(out), _ptr(scr), _stream()), "aware_istft")
    return out


def stft_band(plan: Plan, batch: Batch, audio: torch.Tensor, normalize=True):
    mag = torch.empty((batch.total_frames, SPEC_STRIDE), dtype=torch.float32, device=audio.device)
    ph = torch.empty((batch.total_frames, SPEC_STRIDE), dtype=torch.complex64, device=audio.device)
    scr = batch.scratch()
    check(plan.lib.aware_stft_band(plan.h, batch.h, _ptr(audio), int(normalize), _ptr(mag), _ptr(ph), _ptr(scr), _stream()),
          "aware_stft_band")
    return mag, ph


def stft_bwd(plan: Plan, batch: Batch, grad_spec: torch.Tensor) -> torch.Tensor:
    """Backward of `stft` (normalize=False): grad_spec [total_frames, 520] complex64 -> grad_audio f32 laid out like the
    audio `stft` takes (clip b: lengths[b] samples at in_offsets[b]); any clip length > 512."""
    out = torch.zeros(batch.total_in, dtype=torch.float32, device=grad_spec.device)
    check(plan.lib.aware_stft_bwd(plan.h, batch.h, _ptr(grad_spec), _ptr(out), _stream()), "aware_stft_bwd")
    return out


def istft_bwd(plan: Plan, batch: Batch, grad_audio: torch.Tensor) -> torch.Tensor:
    """Backward of `istft` (normalize=False): grad_audio [total_out] -> grad_spec [total_frames, 520] complex64."""
    gs = torch.empty((batch.total_frames, FULL_STRIDE), dtype=torch.complex64, device=grad_audio.device)
    check(plan.lib.aware_istft_bwd(plan.h, batch.h, _ptr(grad_audio), _ptr(gs), _stream()), "aware_istft_bwd")
    return gs


def polar_decompose(spec: torch.Tensor):
    lib = load_library()
    mag = torch.empty(spec.shape, dtype=torch.float32, device=spec.device)
    ph = torch.empty(spec.shape, dtype=torch.float32, device=spec.device)
    check(lib.aware_polar_decompose(_ptr(spec), _ptr(mag), _ptr(ph), spec.numel(), _stream()), "aware_polar_decompose")
    return mag, ph


def polar_decompose_bwd(spec, grad_mag, grad_phase):
    lib = load_library()
    gs = torch.empty_like(spec)
    check(lib.aware_polar_decompose_bwd(_ptr(spec), _ptr(grad_mag), _ptr(grad_phase), _ptr(gs), spec.numel(), _stream()),
          "aware_polar_decompose_bwd")
    return gs


def polar_assemble(mag: torch.Tensor, phase: torch.Tensor) -> torch.Tensor:
    lib = load_library()
    spec = torch.empty(mag.shape, dtype=torch.complex64, device=mag.device)
    check(lib.aware_polar_assemble(_ptr(mag), _ptr(phase), _ptr(spec), mag.numel(), _stream()), "aware_polar_assemble")
    return spec


def polar_assemble_bwd(mag, phase, grad_spec, need_mag=True, need_phase=True):
    lib = load_library()
    gm = torch.empty_like(mag) if need_mag else None
    gp = torch.empty_like(mag) if need_phase else None
    check(lib.aware_polar_assemble_bwd(_ptr(mag), _ptr(phase), _ptr(grad_spec), _ptr(gm), _ptr(gp), mag.numel(), _stream()),
          "aware_polar_assemble_bwd")
    return gm, gp


def waveform_normalize_bwd(x: "Ragged", grad_out: torch.Tensor) -> torch.Tensor:
    lib = load_library()
    gi = torch.empty_like(x.data)
    check(lib.aware_waveform_normalize_bwd(_ptr(x.data), _ptr(grad_out), _ptr(gi), _ptr(x.d_off), _ptr(x.d_len), x.B, _stream()),
          "aware_waveform_normalize_bwd")
    return gi


class OptClamp:
    """Any optimiser of the reference's registry (embedding/optimizers.py:3-20) as ONE kernel on the caller's tensors, followed by
    torch.clamp(param, lo, hi): the generic form of NAdamClamp for the plug-in seam (aware_opt_clamp_step).  `scheduler`: a
    table scheduler dict of embedding.schedulers.get_scheduler (or None)."""

    def __init__(self, param: torch.Tensor, name: str, num_steps: int, scheduler_name: str | None = None, scheduler_params=None,
                 **params):
        from .embedding.optimizers import get_optimizer, step_table
        from .embedding.schedulers import get_scheduler
        self.lib = load_library()
        self.param = param
        self.opt = get_optimizer(name, None, **params)
        sched = get_scheduler(scheduler_name, self.opt, num_steps, **(scheduler_params or {})) if scheduler_name else None
        if sched is not None and sched["plateau"] is not None:
            raise NotImplementedError("OptClamp: a firing ReduceLROnPlateau needs the per-clip state of an embed session")
        self.table = step_table(self.opt, num_steps, sched["torch"] if sched else None)
        self.s1 = torch.zeros_like(param)
        self.s2 = torch.zeros_like(param)
        self.t = 0

    def step(self, grad: torch.Tensor, lo: torch.Tensor | None = None, hi: torch.Tensor | None = None):
        from .embedding.optimizers import step_scalars
        self.t += 1
        c4, h8 = step_scalars(self.opt, self.table, self.t)
        check(self.lib.aware_opt_clamp_step(self.opt["kind"], _ptr(self.param), _ptr(grad), _ptr(self.s1), _ptr(self.s2), _ptr(lo),
                                            _ptr(hi), self.param.numel(), c4.ctypes.data_as(C.POINTER(C.c_float)),
                                            h8.ctypes.data_as(C.POINTER(C.c_float)), _stream()), "aware_opt_clamp_step")


class NAdamClamp:
    """torch.optim.NAdam (single tensor, defaults of cards/config.yaml) followed by torch.clamp(param, lo, hi), as ONE
    kernel on the caller's tensors (aware_nadam_clamp_step): the optimiser step of the reference's loop
    (embedding/multibit_embedder.py:112-117) for the plug-in seam."""

    def __init__(self, param: torch.Tensor, lr=0.1, betas=(0.9, 0.999), eps=1e-8, momentum_decay=4e-3):
        self.lib = load_library()
        self.param = param
        self.lr, self.betas, self.eps, self.md = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(momentum_decay)
        self.exp_avg = torch.zeros_like(param)
        self.exp_avg_sq = torch.zeros_like(param)
        self.mu_product = C.c_float(1.0)
        self.t = 0

    def step(self, grad: torch.Tensor, lo: torch.Tensor | None = None, hi: torch.Tensor | None = None):
        self.t += 1
        c3 = (C.c_float * 3)()
        check(self.lib.aware_nadam_coefficients(self.t, self.lr, self.betas[0], self.betas[1], self.md, C.byref(self.mu_product), c3),
              "aware_nadam_coefficients")
        check(self.lib.aware_nadam_clamp_step(_ptr(self.param), _ptr(grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq), _ptr(lo),
                                              _ptr(hi), self.param.numel(), c3, self.betas[0], self.betas[1], self.eps, _stream()),
              "aware_nadam_clamp_step")


def detector_backward(plan: Plan, det: "DetectorWeights", batch: Batch, mag: torch.Tensor, grad_values: torch.Tensor):
    """(values [B, n_bits], grad_mag [total_frames, 256]) = forward and J^T grad_values of the frozen network."""
    nbytes = plan.lib.aware_detector_backward_workspace_bytes(batch.h, det.h)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=mag.device)
    vals = torch.empty((batch.B, det.n_bits), dtype=torch.float32, device=mag.device)
    gmag = torch.empty((batch.total_frames, SPEC_STRIDE), dtype=torch.float32, device=mag.device)
    gv = grad_values.contiguous().float()
    check(plan.lib.aware_detector_backward(det.h, batch.h, _ptr(mag), _ptr(gv), _ptr(vals), _ptr(gmag), _ptr(ws), nbytes, _stream()),
          "aware_detector_backward")
    return vals, gmag


def detector_weight_gradients(plan: Plan, det: "DetectorWeights", batch: Batch, mag: torch.Tensor, grad_values: torch.Tensor):
    """EXTENSION (detector training): (values, grad_mag, [dL/dW_l], [dL/db_l]) of the network for the upstream gradient
    grad_values [B, n_bits] (aware_detector_weight_gradients)."""
    lib = plan.lib
    nbytes = lib.aware_detector_train_workspace_bytes(batch.h, det.h)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=mag.device)
    vals = torch.empty((batch.B, det.n_bits), dtype=torch.float32, device=mag.device)
    gmag = torch.empty((batch.total_frames, SPEC_STRIDE), dtype=torch.float32, device=mag.device)
    ch = det.channels
    gw = [torch.empty((ch[i + 1], ch[i]), dtype=torch.float32, device=mag.device) for i in range(len(ch) - 1)]
    gb = [torch.empty((ch[i + 1],), dtype=torch.float32, device=mag.device) for i in range(len(ch) - 1)]
    pw = (C.c_void_p * len(gw))(*[t.data_ptr() for t in gw])
    pb = (C.c_void_p * len(gb))(*[t.data_ptr() for t in gb])
    gv = grad_values.contiguous().float()
    check(lib.aware_detector_weight_gradients(det.h, batch.h, _ptr(mag), _ptr(gv), _ptr(vals), _ptr(gmag), pw, pb, _ptr(ws), nbytes,
                                              _stream()), "aware_detector_weight_gradients")
    return vals, gmag, gw, gb


def detector_train_gradients(plan: Plan, det: "DetectorWeights", batch: Batch, mag: torch.Tensor, target: torch.Tensor,
                             loss: str = "push_extremes", grad_weights=None, grad_biases=None):
    """EXTENSION (detector training): ONE forward + backward of the network with the loss evaluated on the device
    (aware_detector_train_gradients): (values [B, n_bits], per-clip losses [B], [dL/dW_l], [dL/db_l]) -- gradients of the SUM
    of the per-clip losses.  grad_weights / grad_biases: tensors to write into (e.g. views of one flat bucket)."""
    lib = plan.lib
    nbytes = lib.aware_detector_train_workspace_bytes(batch.h, det.h)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=mag.device)
    vals = torch.empty((batch.B, det.n_bits), dtype=torch.float32, device=mag.device)
    losses = torch.empty((batch.B,), dtype=torch.float32, device=mag.device)
    ch = det.channels
    gw = grad_weights or [torch.empty((ch[i + 1], ch[i]), dtype=torch.float32, device=mag.device) for i in range(len(ch) - 1)]
    gb = grad_biases or [torch.empty((ch[i + 1],), dtype=torch.float32, device=mag.device) for i in range(len(ch) - 1)]
    pw = (C.c_void_p * len(gw))(*[t.data_ptr() for t in gw])
    pb = (C.c_void_p * len(gb))(*[t.data_ptr() for t in gb])
    tg = target.contiguous().float()
    check(lib.aware_detector_train_gradients(det.h, batch.h, _ptr(mag), _ptr(tg), LOSS_KINDS[loss], _ptr(losses), _ptr(vals), None,
                                             pw, pb, _ptr(ws), nbytes, _stream()), "aware_detector_train_gradients")
    return vals, losses, gw, gb


class DetectorWeights:
    """Device copy of the frozen detector (aware_detector)."""

    def __init__(self, plan: Plan, mel_basis: np.ndarray, weights, biases):
        require_gpu()
        self.lib = load_library()
        self.plan = plan
        mel = np.ascontiguousarray(mel_basis, dtype=np.float32)
        self._mel = mel
        ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
        chans = [ws[0].shape[1]] + [w.shape[0] for w in ws]
        self.channels = chans
        self.n_bits = chans[-1] // 2
        nl = len(ws)
        wp = (C.c_void_p * nl)(*[w.ctypes.data for w in ws])
        bp = (C.c_void_p * nl)(*[b.ctypes.data for b in bs])
        h = C.c_void_p()
        rc = self.lib.aware_detector_create(C.byref(h), plan.h, C.c_void_p(mel.ctypes.data), mel.shape[0], nl,
                                            (C.c_int * (nl + 1))(*chans), wp, bp)
        check(rc, "aware_detector_create")
        self.h = h

    def update(self, weights, biases):
        """EXTENSION (detector training): replace the parameters in place (same shapes), aware_detector_update."""
        torch.cuda.synchronize()
        ws = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        bs = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
        wp = (C.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
        bp = (C.c_void_p * len(bs))(*[b.ctypes.data for b in bs])
        check(self.lib.aware_detector_update(self.h, C.c_void_p(self._mel.ctypes.data), wp, bp), "aware_detector_update")

    def update_device(self, weights, biases):
        """EXTENSION (detector training): the same from DEVICE tensors, asynchronous on the current stream -- no host round trip
        (aware_detector_update_device: copies, transposes and both packed images rebuilt by kernels)."""
        ws = [w if w.is_contiguous() else w.contiguous() for w in weights]
        bs = [b if b.is_contiguous() else b.contiguous() for b in biases]
        wp = (C.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
        bp = (C.c_void_p * len(bs))(*[b.data_ptr() for b in bs])
        check(self.lib.aware_detector_update_device(self.h, wp, bp, _stream()), "aware_detector_update_device")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.aware_detector_destroy(self.h)
                self.h = None
        except Exception:
            pass


def detect(plan: Plan, det: DetectorWeights, batch: Batch, audio: torch.Tensor) -> torch.Tensor:
    """AWAREDetector.detect for a ragged batch -> [B, n_bits] raw values (device)."""
    nbytes = plan.lib.aware_detect_workspace_bytes(batch.h, det.h)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=audio.device)
    out = torch.empty((batch.B, det.n_bits), dtype=torch.float32, device=audio.device)
    check(plan.lib.aware_detect(plan.h, det.h, batch.h, _ptr(audio), _ptr(out), _ptr(ws), nbytes, _stream()), "aware_detect")
    return out


def detector_forward(plan: Plan, det: DetectorWeights, batch: Batch, mag: torch.Tensor) -> torch.Tensor:
    nbytes = plan.lib.aware_detect_workspace_bytes(batch.h, det.h)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=mag.device)
    out = torch.empty((batch.B, det.n_bits), dtype=torch.float32, device=mag.device)
    check(plan.lib.aware_detector_forward(det.h, batch.h, _ptr(mag), _ptr(out), _ptr(ws), nbytes, _stream()),
          "aware_detector_forward")
    return out


class EmbedSession:
    """One batched run of AWAREEmbedder._optimize (aware_embed)."""

    def __init__(self, plan: Plan, det: DetectorWeights, batch: Batch, num_iterations=400, tolerance_db=6.0,
                 loss="push_extremes", lr=0.1, beta1=0.9, beta2=0.999, eps=1e-8, momentum_decay=4e-3,
                 use_graph=True, conv_pipe="f16x2", fused_readout=True, dsp_path="stream", l1_weight=0.0, mel="taps"):
        """conv_pipe: "f16x2" (default: conv blocks of chip-filling uniform batches on the f16 matrix pipe, two-term operand
        split, three products -- gemm_h2.hip; everything else as "bf16x3"), "bf16x3" (bf16 matrix pipe, exact three-way operand
        split, six products -- gemm_x3.hip) or "f32" (f32-input MFMA);
        fused_readout=False selects the three-kernel read-out that ragged batches use; dsp_path: "stream" (default:
        streaming wave kernels) or "staged" (workgroup-staged kernels) for the STFT / iSTFT stages (aware_embed_config)."""
        self.lib = load_library()
        self.plan, self.det, self.batch = plan, det, batch
        if loss not in LOSS_KINDS:
            raise ValueError(f"Unknown loss type: {loss}. Available on the HIP path: {list(LOSS_KINDS)}")
        if conv_pipe not in CONV_PIPES:
            raise ValueError(f"Unknown conv_pipe: {conv_pipe}")
        self.cfg = EmbedConfig(int(num_iterations), float(tolerance_db), LOSS_KINDS[loss], lr, beta1, beta2, eps,
                               momentum_decay, int(bool(use_graph)), CONV_PIPES[conv_pipe],
                               0 if fused_readout else 1, {"stream": 0, "staged": 1}[dsp_path], float(l1_weight),
                               {"taps": 0, "dense": 1}[mel])
        self.nbytes = self.lib.aware_embed_workspace_bytes(batch.h, det.h)
        self.ws = torch.empty(self.nbytes, dtype=torch.uint8, device=_dev())
        h = C.c_void_p()
        check(self.lib.aware_embed_create(C.byref(h), plan.h, det.h, batch.h, C.byref(self.cfg), _ptr(self.ws),
                                          self.nbytes, _stream()), "aware_embed_create")
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.aware_embed_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_optimizer(self, opt: dict, sched: dict):
        """Any optimiser / scheduler of the reference's registries (embedding.optimizers.get_optimizer /
        embedding.schedulers.get_scheduler dicts) instead of the model card's fused NAdam (aware_embed_set_optimizer)."""
        from .embedding.optimizers import hyper_parameters, step_table
        tab = np.ascontiguousarray(step_table(opt, self.cfg.num_iterations, sched["torch"]), dtype=np.float64)
        hyp, wd = hyper_parameters(opt)
        oc = _lib.OptimizerConfig()
        oc.kind = opt["kind"]
        oc.hyp = (C.c_float * 8)(*hyp)
        oc.weight_decay = wd
        oc.table = tab.ctypes.data_as(C.POINTER(C.c_double))
        pl = sched["plateau"]
        oc.plateau = int(pl is not None)
        if pl:
            oc.patience, oc.factor, oc.threshold, oc.min_lr, oc.eps = pl["patience"], pl["factor"], pl["threshold"], pl["min_lr"], pl["eps"]
        oc.lr0 = float(tab[0, 3])
        check(self.lib.aware_embed_set_optimizer(self.h, C.byref(oc), _stream()), "aware_embed_set_optimizer")
        self._opt_table = tab

    def begin(self, audio: torch.Tensor, target: torch.Tensor):
        self._audio, self._target = audio, target.contiguous().float()
        check(self.lib.aware_embed_begin(self.h, _ptr(audio), _ptr(self._target), _stream()), "aware_embed_begin")

    def iterate(self, n: int):
        rc = self.lib.aware_embed_iterate(self.h, int(n), _stream())
        if rc == -1:
            raise ValueError(f"iterate({n}): more than num_iterations = {self.cfg.num_iterations} steps since begin()")
        check(rc, "aware_embed_iterate")

    def gradient(self) -> torch.Tensor:
        g = torch.zeros((self.batch.total_frames, SPEC_STRIDE), dtype=torch.float32, device=self.ws.device)
        check(self.lib.aware_embed_gradient(self.h, _ptr(g), _stream()), "aware_embed_gradient")
        return g

    def finish(self, rescale: torch.Tensor | None = None) -> torch.Tensor:
        out = torch.empty(self.batch.total_out, dtype=torch.float32, device=self.ws.device)
        check(self.lib.aware_embed_finish(self.h, _ptr(rescale), _ptr(out), _stream()), "aware_embed_finish")
        return out

    def _view(self, which, shape, dtype=torch.float32):
        p = self.lib.aware_embed_buffer(self.h, which)
        n = int(np.prod(shape))
        off = p - self.ws.data_ptr()
        itemsize = torch.empty((), dtype=dtype).element_size()
        return self.ws[off: off + n * itemsize].view(dtype).view(*shape)

    @property
    def loss(self):
        return self._view(0, (self.batch.B,))

    @property
    def best_loss(self):
        return self._view(1, (self.batch.B,))

    @property
    def pred(self):
        return self._view(2, (self.batch.B, self.det.n_bits))

    @property
    def coef(self):
        return self._view(3, (self.batch.total_frames, SPEC_STRIDE))

    @property
    def best_coef(self):
        return self._view(4, (self.batch.total_frames, SPEC_STRIDE))

    @property
    def bounds(self):
        return (self._view(5, (self.batch.total_frames, SPEC_STRIDE)),
                self._view(6, (self.batch.total_frames, SPEC_STRIDE)))

    @property
    def step(self):
        return self._view(8, (1,), torch.int32)

    def clip_learning_rates(self) -> np.ndarray:
        """Per-clip learning rates of a session whose optimiser was set with set_optimizer (device state, float64; they only
        differ from the table's rate under a firing ReduceLROnPlateau)."""
        if not self.lib.aware_embed_buffer(self.h, 11):
            raise AwareHipError("no optimiser was set with set_optimizer")
        return self._view(11, (self.batch.B,), torch.float64).cpu().numpy()


def gemm_nt(a: torch.Tensor, bt: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    lib = load_library()
    M, K = a.shape
    N = bt.shape[0]
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    check(lib.aware_gemm_nt(_ptr(a), a.stride(0), _ptr(bt), bt.stride(0), _ptr(bias), _ptr(c), N, M, N, K, _stream()),
          "aware_gemm_nt")
    return c


def x3_pack(wt: torch.Tensor) -> torch.Tensor:
    """[N][K] f32 weights -> the three bf16 planes in MFMA fragment order (device uint8 tensor)."""
    lib = load_library()
    w = np.ascontiguousarray(wt.detach().cpu().numpy(), dtype=np.float32)
    N, K = w.shape
    nbytes = int(lib.aware_x3_packed_bytes(N, K))
    if nbytes == 0:
        raise AwareHipError("aware_x3_pack: N must be a positive multiple of 16")
    out = np.zeros(nbytes, dtype=np.uint8)
    check(lib.aware_x3_pack(w.ctypes.data, N, K, out.ctypes.data), "aware_x3_pack")
    return torch.from_numpy(out).cuda()


def gemm_clip(a: torch.Tensor, bt: torch.Tensor, bias, B: int, Tp: int, epi: int = 0, rstd=None, act=None,
              mode: int = 0, packed=None):
    """One clip-aligned conv block (tests / roofline).  a: [B*32*ceil(Tp/32), K]; bt: [N, K].
    Returns (C, rstd)."""
    lib = load_library()
    M, K = a.shape
    N = bt.shape[0]
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    if rstd is None:
        rstd = torch.zeros((B, N), dtype=torch.float32, device=a.device)
    if mode == 1 and packed is None:
        packed = x3_pack(bt)
    check(lib.aware_gemm_clip(_ptr(a), a.stride(0), _ptr(bt), bt.stride(0), _ptr(packed), _ptr(bias), _ptr(c), N, B, Tp, N, K,
                              epi, _ptr(rstd), _ptr(act), mode, _stream()), "aware_gemm_clip")
    return c, rstd


def gemm_clip_h2(a: torch.Tensor, bt: torch.Tensor, bias, B: int, Tp: int, epi: int = 0, rstd=None, act=None, w_last=None):
    """One clip-aligned conv block on the f16 two-term kernel (aware_gemm_clip_h2; the embed loop's default conv pipe).
    Returns (C, rstd, amax_out[B, 64]) and, with w_last [CL, N] (epi 1), also zpart [N/128, M, CL]."""
    lib = load_library()
    M, K = a.shape
    N = bt.shape[0]
    dev = a.device
    btd = bt.to(dev).contiguous()
    c = torch.empty((M, N), dtype=torch.float32, device=dev)
    if rstd is None:
        rstd = torch.zeros((B, N), dtype=torch.float32, device=dev)
    amax = torch.zeros((B, 64), dtype=torch.float32, device=dev)
    nbytes = int(lib.aware_gemm_clip_h2_workspace_bytes(B, N, K))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    pkl, zpart, CL = None, None, 0
    if w_last is not None:
        CL = w_last.shape[0]
        wl = torch.zeros((16 * ((CL + 15) // 16), N), dtype=torch.float32)
        wl[:CL] = w_last.detach().cpu().float()
        pkl = x3_pack(wl)
        zpart = torch.zeros((N // 128, M, CL), dtype=torch.float32, device=dev)
    check(lib.aware_gemm_clip_h2(_ptr(a), a.stride(0), _ptr(btd), btd.stride(0), _ptr(bias), _ptr(c), N, B, Tp, N, K, epi, _ptr(rstd),
                                 _ptr(act), _ptr(pkl), _ptr(zpart), CL, _ptr(amax), _ptr(ws), nbytes, _stream()), "aware_gemm_clip_h2")
    torch.cuda.current_stream().synchronize()
    return (c, rstd, amax) if w_last is None else (c, rstd, amax, zpart)


def gemm_clip_last(a: torch.Tensor, bt: torch.Tensor, bias, w_last: torch.Tensor, B: int, Tp: int):
    """The forward conv block + split-K partials of the next (skinny) conv, as the embed loop runs block 2
    (aware_gemm_clip_last).  a: [B*32*ceil(Tp/32), K]; bt: [N, K]; w_last: [CL, N].  Returns (C, rstd, zpart[N/128, M, CL])."""
    lib = load_library()
    M, K = a.shape
    N = bt.shape[0]
    CL = w_last.shape[0]
    clp = 16 * ((CL + 15) // 16)
    wl = torch.zeros((clp, N), dtype=torch.float32)
    wl[:CL] = w_last.detach().cpu().float()
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    rstd = torch.zeros((B, N), dtype=torch.float32, device=a.device)
    zpart = torch.zeros((N // 128, M, CL), dtype=torch.float32, device=a.device)
    pk, pkl = x3_pack(bt), x3_pack(wl)                 # (named: both must stay allocated until the launch has been enqueued)
    check(lib.aware_gemm_clip_last(_ptr(a), a.stride(0), _ptr(pk), _ptr(bias), _ptr(c), N, B, Tp, N, K, _ptr(rstd),
                                   _ptr(pkl), _ptr(zpart), CL, _stream()), "aware_gemm_clip_last")
    torch.cuda.current_stream().synchronize()
    return c, rstd, zpart


# ---------------------------------------------------------------------------------------------
# ragged signal batches and the attack-stage entry points
# ---------------------------------------------------------------------------------------------
class Ragged:
    """B mono clips packed back to back in one device tensor (plumbing for the attack stage)."""

    def __init__(self, data: torch.Tensor, lengths: Sequence[int]):
        self.data = data
        self.lengths = [int(n) for n in lengths]
        self.offsets = np.concatenate([[0], np.cumsum(self.lengths)[:-1]]).astype(np.int64).tolist()
        self.B = len(self.lengths)
        self.max_len = max(self.lengths)
        dev = data.device
        self.d_off = torch.tensor(self.offsets, dtype=torch.int32, device=dev)
        self.d_len = torch.tensor(self.lengths, dtype=torch.int32, device=dev)

    @classmethod
    def from_list(cls, clips, dtype=torch.float32):
        lengths = [len(c) for c in clips]
        data = torch.cat([torch.as_tensor(np.asarray(c), dtype=dtype) for c in clips]).to(_dev())
        return cls(data, lengths)

    def select(self, indices):
        """sub-batch of the given clips (device-side gather of their spans)"""
        parts = [self.data[self.offsets[i]: self.offsets[i] + self.lengths[i]] for i in indices]
        return Ragged(torch.cat(parts) if len(parts) > 1 else parts[0].clone(), [self.lengths[i] for i in indices])

    def like(self, dtype=None):
        return Ragged(torch.empty_like(self.data, dtype=dtype or self.data.dtype), self.lengths)

    def to_list(self):
        flat = self.data.detach().cpu().numpy()
        return [flat[o:o + n] for o, n in zip(self.offsets, self.lengths)]

    def batch(self) -> Batch:
        return Batch(self.lengths)

    def _scratch(self):
        ps = (self.max_len + 4095) // 4096
        return torch.empty(self.B * (ps * 8 + 8) + 64, dtype=torch.uint8, device=self.data.device)


def waveform_normalize(x: Ragged) -> Ragged:
    lib = load_library()
    out = x.like()
    scr = x._scratch()
    check(lib.aware_waveform_normalize(_ptr(x.data), _ptr(out.data), _ptr(x.d_off), _ptr(x.d_len), x.B, x.max_len,
                                       _ptr(scr), _stream()), "aware_waveform_normalize")
    return out


def pcm_quantize(x: Ragged, bits: int) -> Ragged:
    lib = load_library()
    out = x.like()
    scr = x._scratch()
    rc = lib.aware_pcm_quantize(_ptr(x.data), _ptr(out.data), _ptr(x.d_off), _ptr(x.d_len), x.B, x.max_len, int(bits),
                                _ptr(scr), _stream())
    if rc == -1:
        raise ValueError(f"Unsupported PCM bit depth: {bits}")            # scripts/attacks.py:69
    check(rc, "aware_pcm_quantize")
    return out


def upfirdn(x: Ragged, h: torch.Tensor, up: int, down: int, half_len: int) -> Ragged:
    """polyphase resampler core; output length ceil(n*up/down) per clip (scipy.resample_poly)."""
    lib = load_library()
    out_len = [-(-n * up // down) for n in x.lengths]
    out = Ragged(torch.empty(sum(out_len), dtype=torch.float32, device=x.data.device), out_len)
    check(lib.aware_upfirdn(_ptr(x.data), _ptr(x.d_off), _ptr(x.d_len), _ptr(out.data), _ptr(out.d_off), _ptr(out.d_len),
                            x.B, out.max_len, _ptr(h), h.numel(), up, down, half_len, _stream()), "aware_upfirdn")
    return out


def iir(x: Ragged, b: np.ndarray, a: np.ndarray, zi: np.ndarray | None = None, filtfilt=False, out_f64=False) -> Ragged:
    """b, a: [B, ncoef] float64 (a[:,0] == 1) per clip; zi [B, ncoef-1] for filtfilt."""
    lib = load_library()
    dev = x.data.device
    bd = torch.as_tensor(np.ascontiguousarray(b, dtype=np.float64), device=dev)
    ad = torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=dev)
    ncoef = bd.shape[1]
    if filtfilt and min(x.lengths) <= 3 * ncoef:           # scipy.signal.filtfilt raises the same way (padlen = 3*max(len(a), len(b)))
        raise ValueError(f"The length of the input vector x must be greater than padlen, which is {3 * ncoef}.")
    zd = None if zi is None else torch.as_tensor(np.ascontiguousarray(zi, dtype=np.float64), device=dev)
    out = x.like(torch.float64 if out_f64 else torch.float32)
    scr = torch.empty(x.B * (x.max_len + 6 * ncoef), dtype=torch.float64, device=dev) if filtfilt else None
    check(lib.aware_iir(_ptr(x.data), _ptr(x.d_off), _ptr(x.d_len), x.B, x.max_len, _ptr(out.data), int(out_f64),
                        _ptr(bd), _ptr(ad), _ptr(zd), ncoef, int(filtfilt), _ptr(scr), _stream()), "aware_iir")
    return out


def decimate_interp(x: Ragged, factor: int) -> Ragged:
    """x[::factor] then np.interp back to the original length (float64 result), per clip."""
    lib = load_library()
    out = x.like(torch.float64)
    check(lib.aware_decimate_interp(_ptr(x.data), _ptr(x.d_off), _ptr(x.d_len), x.B, x.max_len, int(factor), _ptr(out.data),
                                    _stream()), "aware_decimate_interp")
    return out


def segment_cut(x: Ragged, starts: Sequence[int], cuts: Sequence[int], zero_fill: bool) -> Ragged:
    lib = load_library()
    dev = x.data.device
    out_len = x.lengths if zero_fill else [n - k for n, k in zip(x.lengths, cuts)]
    out = Ragged(torch.empty(sum(out_len), dtype=torch.float32, device=dev), out_len)
    st = torch.tensor(list(starts), dtype=torch.int32, device=dev)
    ct = torch.tensor(list(cuts), dtype=torch.int32, device=dev)
    check(lib.aware_segment_cut(_ptr(x.data), _ptr(x.d_off), _ptr(out.data), _ptr(out.d_off), _ptr(out.d_len), _ptr(st),
                                _ptr(ct), int(zero_fill), x.B, out.max_len, _stream()), "aware_segment_cut")
    return out


def gaussian_noise(x: Ragged, snr_db: float, seeds: Sequence[int]) -> Ragged:
    lib = load_library()
    dev = x.data.device
    out = x.like()
    sd = torch.tensor([int(s) & 0x7FFFFFFF for s in seeds], dtype=torch.int32, device=dev)
    scr = torch.empty(x.B * 8 + 64, dtype=torch.uint8, device=dev)
    check(lib.aware_gaussian_noise(_ptr(x.data), _ptr(out.data), _ptr(x.d_off), _ptr(x.d_len), x.B, x.max_len, _ptr(sd),
                                   float(snr_db), _ptr(scr), _stream()), "aware_gaussian_noise")
    return out


def phase_vocoder_frames(frames: Sequence[int], rate: float):
    """Frames per clip after a phase vocoder at `rate`: len(np.arange(0, T, rate))."""
    return [int(len(np.arange(0, int(t), float(rate)))) for t in frames]


def time_stretch(plan: "Plan", x: Ragged, rate: float) -> Ragged:
    """STFT -> phase vocoder -> iSTFT of every clip (EXTENSION, see aware_phase_vocoder).  Output length
    256*(ceil(T/rate) - 1) samples per clip."""
    lib = load_library()
    dev = x.data.device
    bin_ = Batch(x.lengths)
    spec = stft(plan, bin_, x.data if x.data.dtype == torch.float32 else x.data.float(), normalize=False)
    to = phase_vocoder_frames(bin_.frames, rate)
    if min(to) < 3:
        raise ValueError("time_stretch: a clip would shrink below 3 frames")
    bout = Batch([256 * (t - 1) + 1 for t in to])                    # a batch whose clips have exactly `to` frames
    assert bout.frames == to
    fin = torch.tensor(bin_.frame_offsets, dtype=torch.int32, device=dev)
    fout = torch.tensor(bout.frame_offsets, dtype=torch.int32, device=dev)
    out_spec = torch.empty((bout.total_frames, FULL_STRIDE), dtype=torch.complex64, device=dev)
    check(lib.aware_phase_vocoder(_ptr(spec), _ptr(fin), _ptr(out_spec), _ptr(fout), x.B, float(rate), _stream()),
          "aware_phase_vocoder")
    y = istft(plan, bout, out_spec, normalize=False)
    return Ragged(y, bout.out_lengths)


def snr_db(output: Ragged, target: Ragged) -> torch.Tensor:
    """Per-clip SNR in dB of `output` against `target` over the common length (metrics/audio.py:68-89);
    device float64 tensor [B]."""
    if output.B != target.B:
        raise ValueError("snr_db: batches differ in size")
    lib = load_library()
    dev = output.data.device
    n = torch.minimum(output.d_len, target.d_len)
    out = torch.empty(output.B, dtype=torch.float64, device=dev)
    check(lib.aware_snr(_ptr(output.data), _ptr(output.d_off), _ptr(target.data), _ptr(target.d_off), _ptr(n), output.B,
                        _ptr(out), _stream()), "aware_snr")
    return out


KERNEL_KINDS = ["synth", "analysis", "gemm_nt", "mel_norm", "in_lrelu", "readout_tail", "synth_adjoint",
                "analysis_adjoint_nadam", "misc", "gemm_clip_fwd", "gemm_clip_bwd", "gemm_x3_fwd", "gemm_x3_bwd"]


def embed_profile(sess: EmbedSession, n_iters: int = 3):
    """Per-launch milliseconds of `n_iters` eager loop bodies, measured with HIP events recorded
    on the launch stream (aware_embed_profile).  Returns a list of (kind_name, ms)."""
    cap = 64 * n_iters
    ms = (C.c_float * cap)()
    kind = (C.c_int * cap)()
    n = sess.lib.aware_embed_profile(sess.h, n_iters, cap, ms, kind, _stream())
    if n < 0:
        check(n, "aware_embed_profile")
    return [(KERNEL_KINDS[kind[i]], float(ms[i])) for i in range(n)]


def spectral_quantize(spec: torch.Tensor, step_db: float, floor_db: float) -> torch.Tensor:
    """in place on a [frames, 520] complex64 spectrum (EXTENSION: MP3-like surrogate)"""
    lib = load_library()
    check(lib.aware_spectral_quantize(_ptr(spec), spec.shape[0], float(step_db), float(floor_db), _stream()),
          "aware_spectral_quantize")
    return spec


def spectral_quantize_bwd(spec_in: torch.Tensor, grad_out: torch.Tensor, step_db: float, floor_db: float) -> torch.Tensor:
    """Backward of the surrogate (straight-through on the magnitude, exact through the phase); spec_in: the spectrum before
    quantisation."""
    lib = load_library()
    gin = torch.empty_like(spec_in)
    go = grad_out.contiguous()
    check(lib.aware_spectral_quantize_bwd(_ptr(spec_in), _ptr(go), _ptr(gin), spec_in.shape[0], float(step_db), float(floor_db),
                                          _stream()), "aware_spectral_quantize_bwd")
    return gin


class SpectralQuantizeSTE(torch.autograd.Function):
    """The MP3-like surrogate as a differentiable torch op on a [frames, 520] complex64 spectrum (aware_spectral_quantize /
    aware_spectral_quantize_bwd)."""

    @staticmethod
    def forward(ctx, spec, step_db, floor_db):
        ctx.save_for_backward(spec)
        ctx.step_db, ctx.floor_db = float(step_db), float(floor_db)
        return spectral_quantize(spec.detach().clone(), step_db, floor_db)

    @staticmethod
    def backward(ctx, grad_out):
        (spec,) = ctx.saved_tensors
        return spectral_quantize_bwd(spec, grad_out, ctx.step_db, ctx.floor_db), None, None
