"""Attack stage between embed and detect, on the GPU.

Reference: /root/reference/scripts/attacks.py -- an `Attack` ABC with `.apply(audio, sr)` and a
`.name`, instantiated into a hand-built list by the harness (scripts/test.py:15-18).  The same
classes, constructor arguments and names are kept; each also has `apply_batch(Ragged, sr)` that
runs a whole ragged batch through libaware_hip.so without leaving HBM.  A name -> class registry
(`ATTACKS`, `make_attack`) replaces the hand-built list so that attack chains can be sampled by
name per clip (BASELINE.json config 5).

Random draws stay on the host with the reference's generators (`random.uniform`,
`np.random.randint`), one draw per clip in batch order.

Filter DESIGN (Butterworth coefficients, Kaiser FIR) is done once on the host with scipy.signal,
exactly the calls the reference makes; the filtering itself runs in the HIP kernels.

Out of scope here (external binaries, no offline oracle): MP3Compression (ffmpeg),
TimeStretch / PitchShift (rubberband) -- see DESIGN.md."""
from __future__ import annotations

import random
from abc import ABC, abstractmethod

import numpy as np
import torch

from . import runtime as rt

ATTACKS = {}


def register(cls):
    ATTACKS[cls.__name__] = cls
    return cls


def make_attack(kind: str, **kwargs) -> "Attack":
    if kind not in ATTACKS:
        raise ValueError(f"Unknown attack: {kind}. Available: {sorted(ATTACKS)}")
    return ATTACKS[kind](**kwargs)


class Attack(ABC):
    """scripts/attacks.py:16-30"""
    name = "attack"

    @abstractmethod
    def apply_batch(self, x: "rt.Ragged", sr: int) -> "rt.Ragged":
        ...

    def apply(self, audio, sr):
        out = self.apply_batch(rt.Ragged.from_list([np.asarray(audio, dtype=np.float32)]), sr)
        return out.to_list()[0]


@register
class PCMBitDepthConversion(Attack):
    """:33-70"""

    def __init__(self, pcm=16):
        self.pcm = pcm
        self.name = f"pcm_{pcm}"

    def apply_batch(self, x, sr):
        return rt.pcm_quantize(x, self.pcm)


_FIR_CACHE = {}


def _resample_filter(up, down, device):
    """scipy.signal.resample_poly's default design: firwin(20*max+1, 1/max, ('kaiser', 5.0)),
    cast to float32 (the input dtype) and scaled by `up` (scipy/_signaltools.py resample_poly)."""
    key = (up, down, str(device))
    if key not in _FIR_CACHE:
        from scipy.signal import firwin
        mx = max(up, down)
        half = 10 * mx
        h = firwin(2 * half + 1, 1.0 / mx, window=("kaiser", 5.0)).astype(np.float32)
        h *= up
        _FIR_CACHE[key] = (torch.from_numpy(h).to(device), half)
    return _FIR_CACHE[key]


def resample_poly_batch(x: "rt.Ragged", up: int, down: int) -> "rt.Ragged":
    g = int(np.gcd(up, down))
    up, down = up // g, down // g
    h, half = _resample_filter(up, down, x.data.device)
    return rt.upfirdn(x, h, up, down, half)


@register
class Resample(Attack):
    """:256-294.  sr // target_sr > 1: decimate without anti-alias filter + np.interp back (float64, :275-288);
    otherwise (the harness's sr == target == 16 kHz) the polyphase 441/160 round trip (:290-293)."""

    def __init__(self, target_sr=16000):
        self.target_sr = target_sr
        self.name = f"resample_{target_sr}"

    def apply_batch(self, x, sr):
        if sr // self.target_sr > 1:
            return rt.decimate_interp(x, sr // self.target_sr)
        return resample_poly_batch(resample_poly_batch(x, 441, 160), 160, 441)


def _butter(order, wn, btype):
    from scipy.signal import butter
    return butter(order, wn, btype=btype, analog=False)


class _LFilterAttack(Attack):
    btype = None

    def apply_batch(self, x, sr):
        b, a = _butter(self.order, self.cut_off / (0.5 * sr), self.btype)
        return rt.iir(x, np.tile(b, (x.B, 1)), np.tile(a, (x.B, 1)), out_f64=True)

    def apply(self, audio, sr):
        # the reference's lfilter returns float64 (:414, :452)
        out = self.apply_batch(rt.Ragged.from_list([np.asarray(audio, dtype=np.float32)]), sr)
        return out.to_list()[0]


@register
class LowPassFilter(_LFilterAttack):
    """:388-423"""
    btype = "low"

    def __init__(self, cut_off=4000.0, order=6):
        self.order, self.cut_off, self.name = order, cut_off, "low_pass"


@register
class HighPassFilter(_LFilterAttack):
    """:426-455"""
    btype = "highpass"

    def __init__(self, cut_off=500.0, order=4):
        self.order, self.cut_off, self.name = order, cut_off, "high_pass"


@register
class RandomBandstop(Attack):
    """:298-356 -- one random stop band per clip, zero-phase filtfilt in float64."""

    def __init__(self, band_width=200.0, min_freq=300.0, max_freq=4000.0, order=4):
        self.band_width, self.min_freq, self.max_freq, self.order = float(band_width), float(min_freq), float(max_freq), int(order)
        self.name = f"bandstop_{int(band_width)}Hz"

    def apply_batch(self, x, sr, f_low=None):
        from scipy.signal import lfilter_zi
        nyq = sr / 2.0
        bs, as_, zs = [], [], []
        for i in range(x.B):
            fl = random.uniform(self.min_freq, self.max_freq - self.band_width) if f_low is None else f_low[i]
            b, a = _butter(self.order, [fl / nyq, (fl + self.band_width) / nyq], "bandstop")
            bs.append(b), as_.append(a), zs.append(lfilter_zi(b, a))
        return rt.iir(x, np.stack(bs), np.stack(as_), np.stack(zs), filtfilt=True, out_f64=False)


@register
class DeleteSamples(Attack):
    """:151-178"""

    def __init__(self, percentage):
        self.percentage = percentage
        self.name = f"delete_{percentage}"

    def apply_batch(self, x, sr, starts=None):
        cuts = [int(self.percentage * n) for n in x.lengths]
        if starts is None:
            starts = [int(np.random.randint(0, n - k)) for n, k in zip(x.lengths, cuts)]
        return rt.segment_cut(x, starts, cuts, zero_fill=False)


@register
class Cropout(Attack):
    """:181-205 -- drops the first percentage*sr samples."""

    def __init__(self, percentage):
        self.percentage = percentage
        self.name = f"cropout_{percentage}"

    def apply_batch(self, x, sr):
        k = int(self.percentage * sr)
        return rt.segment_cut(x, [0] * x.B, [k] * x.B, zero_fill=False)


@register
class SampleSupression(Attack):
    """:359-385 (spelling as in the reference)"""

    def __init__(self, percentage):
        self.percentage = percentage
        self.name = f"sample_supression_{percentage}"

    def apply_batch(self, x, sr, starts=None):
        k = int(self.percentage * sr)
        if starts is None:
            starts = [int(np.random.randint(0, n - k)) for n in x.lengths]
        return rt.segment_cut(x, starts, [k] * x.B, zero_fill=True)


@register
class GaussianNoise(Attack):
    """EXTENSION (not in the reference; BASELINE.json north_star / config 3): additive white
    Gaussian noise at `snr_db`, Philox-4x32-10 keyed by the clip's seed.  Specified by
    oracle/aware_oracle.py::gaussian_noise_attack -- parity unpinned."""

    def __init__(self, snr_db=20.0, seed=0):
        self.snr_db, self.seed = float(snr_db), int(seed)
        self.name = f"gaussian_{int(snr_db)}dB"

    def apply_batch(self, x, sr, seeds=None):
        if seeds is None:
            seeds = [self.seed + i for i in range(x.B)]
        return rt.gaussian_noise(x, self.snr_db, seeds)


@register
class MP3Surrogate(Attack):
    """EXTENSION (not in the reference; BASELINE.json north_star): MP3-like quantisation surrogate
    -- STFT -> per-frame log-magnitude quantisation (`step_db` grid, bins more than `-floor_db` below
    the frame maximum dropped) -> iSTFT.  It is NOT a codec and does not replace the reference's
    ffmpeg-based MP3Compression (out of scope: external binary).  Specified by
    oracle/aware_oracle.py::mp3_surrogate_attack -- parity unpinned.  Output length 256*(T-1)."""

    def __init__(self, step_db=1.5, floor_db=-60.0):
        self.step_db, self.floor_db = float(step_db), float(floor_db)
        self.name = f"mp3_surrogate_{step_db}dB"

    def apply_batch(self, x, sr):
        from .utils.audio import default_plan
        plan = default_plan()
        batch = x.batch()
        spec = rt.stft(plan, batch, x.data, normalize=False)
        rt.spectral_quantize(spec, self.step_db, self.floor_db)
        y = rt.istft(plan, batch, spec, normalize=False)
        return rt.Ragged(y, batch.out_lengths)


@register
class TimeStretch(Attack):
    """EXTENSION in place of scripts/attacks.py:208-228 (pyrubberband -> rubberband binary, absent): phase-vocoder
    time-scale modification on the STFT kernels.  rate > 1: faster / shorter, rate < 1: slower / longer.  Specified by
    oracle/aware_oracle.py::time_stretch_attack -- parity with rubberband unpinned.  Output length 256*(ceil(T/rate)-1)."""

    def __init__(self, rate=1.0):
        self.rate = float(rate)
        self.name = f"ts_{rate}"

    def apply_batch(self, x, sr):
        from .utils.audio import default_plan
        return rt.time_stretch(default_plan(), x, self.rate)


@register
class PitchShift(Attack):
    """EXTENSION in place of scripts/attacks.py:231-252: pitch shift by `cents`/100 semitones (the reference's own
    unit conversion, :249) = phase-vocoder stretch by 2^(semitones/12) followed by polyphase resampling back to the
    original duration.  Specified by oracle/aware_oracle.py::pitch_shift_attack -- parity with rubberband unpinned."""

    def __init__(self, cents=5):
        self.cents = cents
        self.name = f"ps_{cents}"

    def ratio(self):
        from fractions import Fraction
        factor = 2.0 ** ((self.cents / 100.0) / 12.0)
        fr = Fraction(1.0 / factor).limit_denominator(512)          # resampling ratio up/down ~ 1/factor
        return factor, fr.numerator, fr.denominator

    def apply_batch(self, x, sr):
        from .utils.audio import default_plan
        factor, up, down = self.ratio()
        y = rt.time_stretch(default_plan(), x, 1.0 / factor)
        return y if up == down else resample_poly_batch(y, up, down)


def reference_attack_list():
    """The subset of the harness's 22-entry list (scripts/test.py:15-18) that runs here."""
    return [PCMBitDepthConversion(8), PCMBitDepthConversion(12), PCMBitDepthConversion(16), PCMBitDepthConversion(24),
            DeleteSamples(0.1), DeleteSamples(0.15), DeleteSamples(0.2), Resample(), RandomBandstop(),
            SampleSupression(0.1), SampleSupression(0.25), LowPassFilter(), HighPassFilter()]


def config3_attack_stack():
    """BASELINE.json config 3: resample 44.1k<->16k + lowpass + Gaussian noise + quantisation."""
    return [Resample(), LowPassFilter(), GaussianNoise(20.0), PCMBitDepthConversion(16)]
