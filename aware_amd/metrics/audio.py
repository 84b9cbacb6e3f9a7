"""BER (percent) and SNR (dB).  Reference: src/AWARE/metrics/audio.py:8-17, :68-89.
STOI: a restatement of the published measure (pystoi is absent: parity unpinned), below.  PESQ (ITU-T P.862) wraps a third-party
perceptual model that is out of scope."""
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


# ---------------------------------------------------------------------------------------------------------------------
# STOI (reference: src/AWARE/metrics/audio.py:46-64 -- mono mix, common length, resample to 16 kHz, pystoi.stoi(target, output,
# 16000)).  pystoi and librosa are not importable here and the reference holds no STOI fixture: this is a restatement of the
# published measure (Taal, Hendriks, Heusdens, Jensen, "An Algorithm for Intelligibility Prediction of Time-Frequency Weighted
# Noisy Speech", IEEE TASLP 2011) with the constants and framing of pystoi 0.3 -- PARITY UNPINNED.  Host metric (numpy), not
# part of the hot path.
# ---------------------------------------------------------------------------------------------------------------------
_STOI_FS, _STOI_FRAME, _STOI_NFFT, _STOI_BANDS, _STOI_MINF, _STOI_N, _STOI_BETA, _STOI_DYN = 10000, 256, 512, 15, 150.0, 30, -15.0, 40.0
_EPS = np.finfo(np.float64).eps


def _resample_window_oct(p, q):
    """Kaiser-windowed sinc of Octave's resample() (what pystoi.utils.resample_oct designs)."""
    g = np.gcd(p, q)
    p, q = p // g, q // g
    cutoff = 1.0 / (2 * max(p, q))
    roll_off = cutoff / 10.0
    rejection_db = 60.0
    half = int(np.ceil((rejection_db - 8.0) / (28.714 * roll_off)))
    t = np.arange(-half, half + 1)
    ideal = 2 * p * cutoff * np.sinc(2 * cutoff * t)
    beta = 0.1102 * (rejection_db - 8.7)
    return np.kaiser(2 * half + 1, beta) * ideal


def _resample_oct(x, p, q):
    from scipy.signal import resample_poly
    h = _resample_window_oct(p, q)
    return resample_poly(x, p, q, window=h / np.sum(h))


def _third_octave_matrix(fs, nfft, num_bands, min_freq):
    f = np.linspace(0, fs, nfft + 1)[: nfft // 2 + 1]
    k = np.arange(num_bands, dtype=np.float64)
    lo = min_freq * 2.0 ** ((2 * k - 1) / 6)
    hi = min_freq * 2.0 ** ((2 * k + 1) / 6)
    obm = np.zeros((num_bands, len(f)))
    for i in range(num_bands):
        a = int(np.argmin((f - lo[i]) ** 2))
        b = int(np.argmin((f - hi[i]) ** 2))
        obm[i, a:b] = 1.0
    return obm


def _frames(x, size, hop, window):
    idx = range(0, len(x) - size, hop)
    return np.array([window * x[i:i + size] for i in idx]) if len(x) > size else np.zeros((0, size))


def _remove_silent_frames(x, y, dyn_range, size, hop):
    w = np.hanning(size + 2)[1:-1]
    xf, yf = _frames(x, size, hop, w), _frames(y, size, hop, w)
    if len(xf) == 0:
        return x[:0], y[:0]
    energy = 20 * np.log10(np.linalg.norm(xf, axis=1) + _EPS)
    keep = (np.max(energy) - dyn_range - energy) < 0
    xf, yf = xf[keep], yf[keep]
    n = (len(xf) - 1) * hop + size if len(xf) else 0
    xs, ys = np.zeros(n), np.zeros(n)
    for i in range(len(xf)):                                    # overlap-add of the frames that are kept
        xs[i * hop:i * hop + size] += xf[i]
        ys[i * hop:i * hop + size] += yf[i]
    return xs, ys


def stoi(clean, processed, fs_sig: int) -> float:
    """Short-time objective intelligibility of `processed` against `clean` (1-D arrays of equal length, rate fs_sig)."""
    x, y = np.asarray(clean, dtype=np.float64), np.asarray(processed, dtype=np.float64)
    if x.shape != y.shape:
        raise ValueError("x and y should have the same length")       # pystoi's check
    if fs_sig != _STOI_FS:
        x, y = _resample_oct(x, _STOI_FS, fs_sig), _resample_oct(y, _STOI_FS, fs_sig)
    x, y = _remove_silent_frames(x, y, _STOI_DYN, _STOI_FRAME, _STOI_FRAME // 2)
    w = np.hanning(_STOI_FRAME + 2)[1:-1]
    xs = np.fft.rfft(_frames(x, _STOI_FRAME, _STOI_FRAME // 2, w), n=_STOI_NFFT).T       # [bins][frames]
    ys = np.fft.rfft(_frames(y, _STOI_FRAME, _STOI_FRAME // 2, w), n=_STOI_NFFT).T
    if xs.shape[-1] < _STOI_N:
        return 1e-5                                                 # not enough speech frames for one 384 ms segment
    obm = _third_octave_matrix(_STOI_FS, _STOI_NFFT, _STOI_BANDS, _STOI_MINF)
    xt, yt = np.sqrt(obm @ np.abs(xs) ** 2), np.sqrt(obm @ np.abs(ys) ** 2)     # [bands][frames]
    m = np.arange(_STOI_N, xt.shape[1] + 1)
    xseg = np.stack([xt[:, i - _STOI_N:i] for i in m])              # [segments][bands][N]
    yseg = np.stack([yt[:, i - _STOI_N:i] for i in m])
    norm = lambda a: np.linalg.norm(a, axis=2, keepdims=True)
    yn = yseg * (norm(xseg) / (norm(yseg) + _EPS))
    yp = np.minimum(yn, xseg * (1 + 10 ** (-_STOI_BETA / 20)))      # clipping at -15 dB SDR
    yp = yp - yp.mean(axis=2, keepdims=True)
    xc = xseg - xseg.mean(axis=2, keepdims=True)
    yp = yp / (norm(yp) + _EPS)
    xc = xc / (norm(xc) + _EPS)
    return float(np.sum(yp * xc) / (xseg.shape[0] * xseg.shape[1]))


class STOI(BaseMetrics):
    """metrics/audio.py:46-64.  The reference resamples both signals to 16 kHz with librosa (soxr) first; here the polyphase
    resampler of the attack stack's host twin (scipy.signal.resample_poly) takes that step -- PARITY UNPINNED (see above)."""

    def __call__(self, output, target, sampling_rate: int) -> float:
        from scipy.signal import resample_poly
        o, t = _np(output).astype(np.float64), _np(target).astype(np.float64)
        if o.ndim == 2 and o.shape[1] == 2:
            o, t = o.mean(axis=1), t.mean(axis=1)
        n = min(len(o), len(t))
        o, t = o[:n], t[:n]
        if sampling_rate != 16000:
            g = int(np.gcd(16000, int(sampling_rate)))
            o, t = resample_poly(o, 16000 // g, int(sampling_rate) // g), resample_poly(t, 16000 // g, int(sampling_rate) // g)
        return stoi(t, o, 16000)


class PESQ(BaseMetrics):
    """metrics/audio.py:19-43 wraps the `pesq` package (ITU-T P.862 wide-band, a third-party C model).  It is not importable
    here and P.862 is not restated: the name exists so that code written against the reference imports, and says so when called."""

    def __call__(self, output, target, sampling_rate: int) -> float:
        try:
            from pesq import pesq                                    # the reference's own dependency, when the host has it
        except ImportError as e:
            raise NotImplementedError("PESQ needs the third-party `pesq` package (ITU-T P.862); it is outside this build's scope "
                                      "(SURVEY.md 8f) -- SNR and STOI are available") from e
        from scipy.signal import resample_poly
        o, t = _np(output).astype(np.float64), _np(target).astype(np.float64)
        if o.ndim == 2 and o.shape[1] == 2:
            o, t = o.mean(axis=1), t.mean(axis=1)
        n = min(len(o), len(t))
        o, t = o[:n], t[:n]
        if sampling_rate != 16000:
            g = int(np.gcd(16000, int(sampling_rate)))
            o, t = resample_poly(o, 16000 // g, int(sampling_rate) // g), resample_poly(t, 16000 // g, int(sampling_rate) // g)
        return float(pesq(16000, t, o, "wb"))
