"""Host metrics (reference: src/AWARE/metrics/audio.py).  BER / SNR are checked elsewhere against their definitions; STOI is a
restatement of the published measure -- pystoi and librosa are not importable here and the reference holds no STOI fixture, so
its parity is UNPINNED: these tests hold the restatement to the measure's defining properties and to the constants of the paper
(10 kHz, 256-sample Hann frames with 50 % overlap, 15 third-octave bands from 150 Hz, 384 ms segments, -15 dB clipping)."""
import numpy as np
import pytest

from aware_amd.metrics.audio import STOI, SNR, BER, stoi, _third_octave_matrix, _resample_window_oct


def _speechlike(seconds=3.0, fs=16000, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(int(seconds * fs)) / fs
    x = sum(np.sin(2 * np.pi * 150 * (k + 1) * t) / (k + 1) for k in range(20)) * (0.5 + 0.5 * np.sin(2 * np.pi * 4 * t)) ** 2
    return x + 0.01 * rng.standard_normal(len(x)), rng


def test_third_octave_bands_of_the_paper():
    obm = _third_octave_matrix(10000, 512, 15, 150.0)
    assert obm.shape == (15, 257) and set(np.unique(obm)) == {0.0, 1.0}
    f = np.linspace(0, 10000, 513)[:257]
    for k in range(15):
        cols = np.nonzero(obm[k])[0]
        cf = 150.0 * 2.0 ** (k / 3)
        assert len(cols) >= 2 and np.all(np.diff(cols) == 1)                      # one contiguous run of bins
        assert f[cols[0]] <= cf <= f[cols[-1]] + (f[1] - f[0])                    # around the band's centre frequency
        assert abs(f[cols[0]] - cf * 2 ** (-1 / 6)) <= (f[1] - f[0])              # edges on the bins nearest cf * 2^(+-1/6)
    assert np.nonzero(obm[14])[0][-1] < 257 and obm.sum(0).max() == 1.0            # bands do not overlap


def test_octave_resampling_window():
    h = _resample_window_oct(10000, 16000)                     # 5 / 8: cutoff 1/16, 60 dB Kaiser
    assert len(h) % 2 == 1 and np.allclose(h, h[::-1]) and abs(np.sum(h) - 5.0) < 0.05


def test_stoi_properties():
    x, rng = _speechlike()
    assert abs(stoi(x, x, 16000) - 1.0) < 1e-9
    scores = []
    for snr in (30, 10, 0, -10):
        n = rng.standard_normal(len(x))
        n *= np.sqrt(np.mean(x ** 2) / np.mean(n ** 2)) * 10 ** (-snr / 20)
        scores.append(stoi(x, x + n, 16000))
    assert all(a > b for a, b in zip(scores, scores[1:])) and 0.9 < scores[0] < 1.0 and scores[-1] < 0.7
    assert abs(stoi(x, 0.3 * x, 16000) - 1.0) < 1e-6                               # level-invariant
    assert stoi(x[:2000], x[:2000], 16000) == 1e-5                                 # shorter than one 384 ms segment
    with pytest.raises(ValueError):
        stoi(x, x[:-1], 16000)


def test_stoi_metric_follows_the_reference_wrapper():
    """metrics/audio.py:46-64: stereo is mixed to mono, both signals cut to the common length, anything not at 16 kHz is
    resampled first; `output` is the processed signal, `target` the clean one."""
    x, rng = _speechlike(seed=3)
    y = x + 0.3 * rng.standard_normal(len(x))
    m = STOI()
    s = m(y, x, 16000)
    assert abs(s - stoi(x, y, 16000)) < 1e-12
    assert abs(m(np.stack([y, y], 1), np.stack([x, x], 1), 16000) - s) < 1e-12     # stereo -> mono mean
    assert abs(m(y, x[:-100], 16000) - stoi(x[:-100], y[:-100], 16000)) < 1e-12    # common length
    x44, rng = _speechlike(fs=44100, seed=4)
    y44 = x44 + 0.3 * rng.standard_normal(len(x44))
    assert 0.3 < m(y44, x44, 44100) < 1.0
    assert BER()(np.array([0, 1, 1, 0]), np.array([0, 1, 0, 0])) == 25.0 and SNR()(x, x) == float("inf")


def test_pesq_name_exists_and_says_what_is_missing():
    from aware_amd.metrics import PESQ
    x, _ = _speechlike(seconds=1.0)
    try:
        import pesq  # noqa: F401
    except ImportError:
        with pytest.raises(NotImplementedError, match="P.862"):
            PESQ()(x, x, 16000)
