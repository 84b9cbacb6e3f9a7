"""DSP plug-ins with the reference's call shape, backed by the HIP kernels.

Reference: src/AWARE/utils/audio/stft.py:4-69 and waveform.py:8-46 -- objects held in ordered
Python lists (`audio_preprocess_pipeline`, `audio_postprocess_pipeline`) and called tensor ->
tensor.  Layout at this seam is the reference's: a spectrum is complex64 [n_fft/2+1, T]
(freq-major); the frame-major device layout of the hot loop stays behind the C ABI."""
from __future__ import annotations

import numpy as np
import torch

from ...interfaces import BaseAudioProcessor
from ... import runtime as rt
from ..logger import logger

_PLANS = {}


def band_bins(sample_rate: int, n_fft: int, bands) -> tuple:
    """AWAREEmbedder._get_embedding_frequency_indices (multibit_embedder.py:43-47): bins whose
    centre np.linspace(0, sr/2, n_fft/2+1)[k] lies in [bands[0], bands[1]]."""
    f = np.linspace(0.0, sample_rate / 2.0, 1 + n_fft // 2, endpoint=True)
    idx = np.where((f >= bands[0]) & (f <= bands[1]))[0]
    if len(idx) == 0:
        raise ValueError("embedding band contains no FFT bin")
    return int(idx[0]), int(idx[-1])


def get_plan(n_fft=1024, hop=256, window="hann", bins=(32, 256)) -> "rt.Plan":
    key = (n_fft, hop, window, tuple(bins), torch.cuda.current_device() if torch.cuda.is_available() else -1)
    if key not in _PLANS:
        _PLANS[key] = rt.Plan(n_fft, hop, n_fft, window, bins)
    return _PLANS[key]


def default_plan() -> "rt.Plan":
    return get_plan()


def _as_device(x: torch.Tensor) -> torch.Tensor:
    return x.to("cuda", torch.float32).contiguous()


_BATCHES = {}


def get_batch(lengths) -> "rt.Batch":
    """Geometry handles are cached per (device, lengths): a plug-in call does no hipMalloc / host-to-device table copy
    after its first use of a shape."""
    key = (torch.cuda.current_device(),) + tuple(int(n) for n in lengths)
    if key not in _BATCHES:
        if len(_BATCHES) > 256:
            _BATCHES.clear()
        _BATCHES[key] = rt.Batch(list(key[1:]))
    return _BATCHES[key]


# ---- differentiable forms: torch.autograd.Function wrappers whose forward and backward are C-ABI calls ----------
# (the reference's loop differentiates through these objects with autograd, multibit_embedder.py:49-67,:111)
class _NormalizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        flat = x.reshape(-1)
        rg = rt.Ragged(flat, [flat.numel()])
        ctx.save_for_backward(flat)
        return rt.waveform_normalize(rg).data.reshape(x.shape)

    @staticmethod
    def backward(ctx, g):
        (flat,) = ctx.saved_tensors
        rg = rt.Ragged(flat, [flat.numel()])
        return rt.waveform_normalize_bwd(rg, g.contiguous().reshape(-1).float()).reshape(g.shape)


class _STFTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_fft, hop, window):
        plan = get_plan(n_fft, hop, window)
        batch = get_batch([x.numel()])
        ctx.geom = (plan, batch, n_fft)
        spec = rt.stft(plan, batch, x, normalize=False)
        return spec[:, : n_fft // 2 + 1].transpose(0, 1).contiguous()        # [F, T]

    @staticmethod
    def backward(ctx, g):
        plan, batch, n_fft = ctx.geom
        F = n_fft // 2 + 1
        gs = torch.zeros((batch.total_frames, rt.FULL_STRIDE), dtype=torch.complex64, device=g.device)
        gs[:, :F] = g.transpose(0, 1)
        return rt.stft_bwd(plan, batch, gs), None, None, None


class _ISTFTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, n_fft, hop, window):
        F, T = X.shape
        plan = get_plan(n_fft, hop, window)
        batch = get_batch([max(hop * (T - 1), n_fft // 2 + 1)])
        ctx.geom = (plan, batch, F)
        spec = torch.zeros((T, rt.FULL_STRIDE), dtype=torch.complex64, device=X.device)
        spec[:, :F] = X.transpose(0, 1)
        return rt.istft(plan, batch, spec, normalize=False)

    @staticmethod
    def backward(ctx, g):
        plan, batch, F = ctx.geom
        gs = rt.istft_bwd(plan, batch, g.contiguous().float())
        return gs[:, :F].transpose(0, 1).contiguous(), None, None, None


class _DecomposeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, S):
        S = S.contiguous()
        ctx.save_for_backward(S)
        return rt.polar_decompose(S)

    @staticmethod
    def backward(ctx, gmag, gphase):
        (S,) = ctx.saved_tensors
        gm = None if gmag is None else gmag.contiguous().float()
        gp = None if gphase is None else gphase.contiguous().float()
        if gm is None and gp is None:
            return torch.zeros_like(S)
        return rt.polar_decompose_bwd(S, gm, gp)


class _AssembleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mag, phase):
        mag, phase = mag.contiguous().float(), phase.contiguous().float()
        ctx.save_for_backward(mag, phase)
        return rt.polar_assemble(mag, phase)

    @staticmethod
    def backward(ctx, g):
        mag, phase = ctx.saved_tensors
        gm, gp = rt.polar_assemble_bwd(mag, phase, g.contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gm, gp


class WaveformNormalizer(BaseAudioProcessor):
    """x / max(|x| + 1e-8) over the whole tensor (waveform.py:18-19); differentiable (through the max, as autograd does)."""

    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        return _NormalizeFn.apply(_as_device(data))


class STFT(BaseAudioProcessor):
    """torch.stft(center=True, window, return_complex=True) (stft.py:14-28); differentiable for every input it accepts
    (more than n_fft/2 samples)."""

    def __init__(self, n_fft: int = 2048, hop_length: int = 512, window: str = "hann", win_length: int = 2048):
        if window not in ("hann", "hamming"):
            raise ValueError(f"Invalid window type: {window}")
        self.n_fft, self.hop_length, self.window_name, self.win_length = n_fft, hop_length, window, win_length

    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        return _STFTFn.apply(_as_device(data), self.n_fft, self.hop_length, self.window_name)


class ISTFT(BaseAudioProcessor):
    """torch.istft(center=True, window) without `length` (stft.py:34-48); differentiable."""

    def __init__(self, n_fft: int = 2048, hop_length: int = 512, window: str = "hann", win_length: int = 2048):
        if window not in ("hann", "hamming"):
            raise ValueError(f"Invalid window type: {window}")
        self.n_fft, self.hop_length, self.window_name, self.win_length = n_fft, hop_length, window, win_length

    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        return _ISTFTFn.apply(data.to("cuda", torch.complex64), self.n_fft, self.hop_length, self.window_name)


class STFTDecomposer(BaseAudioProcessor):
    """(|S|, angle S) (stft.py:54-55); differentiable (d|S| = 0 at S = 0, torch's convention)."""

    def __call__(self, data: torch.Tensor):
        return _DecomposeFn.apply(data.to("cuda", torch.complex64))


class STFTAssembler(BaseAudioProcessor):
    """mag * exp(i phase) (stft.py:61-62); differentiable."""

    def __call__(self, magnitude: torch.Tensor, phase: torch.Tensor) -> torch.Tensor:
        return _AssembleFn.apply(magnitude.to("cuda"), phase.to("cuda"))


class STFTNormalizer(BaseAudioProcessor):
    """stft.py:64-69 (unused by the reference's pipelines)."""

    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        return data / torch.max(torch.abs(data) + 1e-8)


class SilenceChecker(BaseAudioProcessor):
    """Voice-activity gate of embed_watermark (waveform.py:22-46).

    The reference gates on webrtcvad (third-party C, absent offline).  When that module is
    importable it is used exactly as the reference does; otherwise the gate reports "not
    silent" -- its parity is unpinned."""

    _warned = False

    def __init__(self, sample_rate=16000, aggr=3, frame_ms=30.0, min_speech_seconds=0.01):
        self.sample_rate, self.aggr, self.frame_ms, self.min_speech_seconds = sample_rate, aggr, frame_ms, min_speech_seconds

    def __call__(self, data: np.ndarray) -> bool:
        try:
            import webrtcvad
        except ImportError:
            if not SilenceChecker._warned:
                logger.warning("webrtcvad is not installed: the silence gate is disabled")
                SilenceChecker._warned = True
            return False
        pcm = (np.asarray(data) * 32767).astype(np.int16).tobytes()
        vad = webrtcvad.Vad(self.aggr)
        step = int(self.sample_rate * self.frame_ms / 1000) * 2
        voiced = sum(vad.is_speech(pcm[i:i + step], self.sample_rate)
                     for i in range(0, len(pcm) - step + 1, step))
        return voiced * (self.frame_ms / 1000.0) < self.min_speech_seconds
