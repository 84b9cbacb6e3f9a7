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


class WaveformNormalizer(BaseAudioProcessor):
    """x / max(|x| + 1e-8) over the whole tensor (waveform.py:18-19)."""

    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        x = _as_device(data).reshape(-1)
        out = rt.waveform_normalize(rt.Ragged(x, [x.numel()]))
        return out.data.reshape(data.shape)


class STFT(BaseAudioProcessor):
    """torch.stft(center=True, window, return_complex=True) (stft.py:14-28)."""

    def __init__(self, n_fft: int = 2048, hop_length: int = 512, window: str = "hann", win_length: int = 2048):
        if window not in ("hann", "hamming"):
            raise ValueError(f"Invalid window type: {window}")
        self.n_fft, self.hop_length, self.window_name, self.win_length = n_fft, hop_length, window, win_length

    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        x = _as_device(data)
        plan = get_plan(self.n_fft, self.hop_length, self.window_name)
        batch = rt.Batch([x.numel()])
        spec = rt.stft(plan, batch, x, normalize=False)
        return spec[:, : self.n_fft // 2 + 1].transpose(0, 1).contiguous()        # [F, T]


class ISTFT(BaseAudioProcessor):
    """torch.istft(center=True, window) without `length` (stft.py:34-48)."""

    def __init__(self, n_fft: int = 2048, hop_length: int = 512, window: str = "hann", win_length: int = 2048):
        if window not in ("hann", "hamming"):
            raise ValueError(f"Invalid window type: {window}")
        self.n_fft, self.hop_length, self.window_name, self.win_length = n_fft, hop_length, window, win_length

    def __call__(self, data: torch.Tensor) -> torch.Tensor:
        X = data.to("cuda", torch.complex64)
        F, T = X.shape
        plan = get_plan(self.n_fft, self.hop_length, self.window_name)
        batch = rt.Batch([max(self.hop_length * (T - 1), self.n_fft // 2 + 1)])
        spec = torch.zeros((T, rt.FULL_STRIDE), dtype=torch.complex64, device="cuda")
        spec[:, :F] = X.transpose(0, 1)
        return rt.istft(plan, batch, spec, normalize=False)


class STFTDecomposer(BaseAudioProcessor):
    """(|S|, angle S) (stft.py:54-55)."""

    def __call__(self, data: torch.Tensor):
        return torch.abs(data), torch.angle(data)


class STFTAssembler(BaseAudioProcessor):
    """mag * exp(i phase) (stft.py:61-62)."""

    def __call__(self, magnitude: torch.Tensor, phase: torch.Tensor) -> torch.Tensor:
        return torch.polar(magnitude.float(), phase.float())


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
