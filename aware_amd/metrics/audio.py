"""BER (percent) and SNR (dB).  Reference: src/AWARE/metrics/audio.py:8-17, :68-89.
PESQ / STOI wrap third-party perceptual models that are out of scope."""
import numpy as np
import torch

from ..interfaces import BaseMetrics


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


class BER(BaseMetrics):
    def __call__(self, output, target) -> float:
        return float(np.mean(_np(output) != _np(target)) * 100)


class SNR(BaseMetrics):
    def __call__(self, output, target) -> float:
        o, t = _np(output), _np(target)
        if o.ndim == 2 and o.shape[1] == 2:
            o, t = o.mean(axis=1), t.mean(axis=1)
        n = min(len(o), len(t))
        o, t = o[:n], t[:n]
        if np.all(o == t):
            return float("inf")
        return float(10 * np.log10(np.mean(o ** 2) / np.mean((o - t) ** 2)))


def snr_batch(output, target):
    """SNR of every clip of two ragged device batches (aware_amd.runtime.Ragged), on the GPU: float64 tensor [B]."""
    from .. import runtime as rt
    return rt.snr_db(output, target)
