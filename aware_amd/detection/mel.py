"""Slaney-scale mel filter bank (host, float64 -> float32), built once per detector.

Reference: src/AWARE/detection/modules/mel.py:105-149 (get_mel_filter_bank with htk=False,
norm="slaney", fmin=0, fmax=sr/2), mel scale :6-69, bin centres :72-74."""
import numpy as np

_F_SP = 200.0 / 3.0
_BREAK_HZ = 1000.0
_BREAK_MEL = _BREAK_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(hz):
    hz = np.asarray(hz, dtype=np.float64)
    lin = hz / _F_SP
    with np.errstate(divide="ignore", invalid="ignore"):
        log = _BREAK_MEL + np.log(np.maximum(hz, 1e-300) / _BREAK_HZ) / _LOGSTEP
    return np.where(hz >= _BREAK_HZ, log, lin)


def mel_to_hz(mel):
    mel = np.asarray(mel, dtype=np.float64)
    return np.where(mel >= _BREAK_MEL, _BREAK_HZ * np.exp(_LOGSTEP * (mel - _BREAK_MEL)), _F_SP * mel)


def fft_frequencies(sr, n_fft):
    return np.linspace(0.0, sr / 2.0, 1 + n_fft // 2, endpoint=True)


def mel_filter_bank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, dtype=np.float32):
    """[n_mels, 1 + n_fft//2] triangular filters, area-normalised (slaney)."""
    fmax = sr / 2.0 if fmax is None else fmax
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    bins = fft_frequencies(sr, n_fft)
    width = np.diff(edges)
    rel = edges[:, None] - bins[None, :]                 # [n_mels+2, n_bins]
    rising = -rel[:-2] / width[:-1, None]
    falling = rel[2:] / width[1:, None]
    tri = np.maximum(0.0, np.minimum(rising, falling)).astype(dtype)
    tri *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return tri
