"""WAV reader / writer of the harness (aware_amd/utils/audio/io.py): round trips and hand-built files."""
import struct

import numpy as np
import pytest

from aware_amd.utils.audio import io


def test_pcm16_and_float_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    x = (0.3 * rng.standard_normal(4000)).astype(np.float32).clip(-0.99, 0.99)
    p = tmp_path / "a.wav"
    io.write_wav(p, x, 16000)
    y, sr = io.load(p)
    assert sr == 16000 and y.dtype == np.float32 and y.shape == x.shape
    assert np.max(np.abs(y - x)) <= 0.5 / 32768 + 1e-7                    # one 16-bit quantisation step
    io.write_wav(p, x, 44100, subtype="FLOAT")
    y, sr = io.load(p)
    assert sr == 44100
    np.testing.assert_array_equal(y, x)                                  # float WAV is exact
    st = np.stack([x, -x], axis=1)
    io.write_wav(p, st, 16000, subtype="FLOAT")
    y2, _ = io.read_wav(p)
    assert y2.shape == (4000, 2)
    ym, _ = io.load(p, mono=True)
    np.testing.assert_allclose(ym, 0.0, atol=1e-7)                       # channels averaged like librosa.load(mono=True)
    with pytest.raises(ValueError):
        io.load(p, sr=8000)


def test_hand_built_pcm24_pcm8_extensible(tmp_path):
    # 24-bit: values -2^23, -1, 0, 1, 2^23 - 1
    vals = [-(1 << 23), -1, 0, 1, (1 << 23) - 1]
    body = b"".join(struct.pack("<i", v)[:3] for v in vals)
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(body), b"WAVE", b"fmt ", 16, 1, 1, 8000, 8000 * 3, 3, 24, b"data", len(body))
    p = tmp_path / "b.wav"
    p.write_bytes(hdr + body + b"\0")                                    # odd data size is padded in RIFF
    y, sr = io.read_wav(p)
    assert sr == 8000
    np.testing.assert_allclose(y, np.array(vals, dtype=np.float64) / 8388608.0, atol=1e-7)
    # 8-bit unsigned, WAVE_FORMAT_EXTENSIBLE header with the PCM sub-format
    body = bytes([0, 128, 255])
    fmt = struct.pack("<HHIIHHHHIH14s", 0xFFFE, 1, 8000, 8000, 1, 8, 22, 8, 4, 1, b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71")
    raw = b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(body) + 1) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(body)) + body + b"\0"
    p.write_bytes(raw)
    y, _ = io.read_wav(p)
    np.testing.assert_allclose(y, [-1.0, 0.0, 127.0 / 128.0])
    p.write_bytes(b"not a wave file at all")
    with pytest.raises(ValueError):
        io.read_wav(p)


def test_malformed_files_raise_value_error(tmp_path):
    """Everything read_wav cannot decode raises ValueError (run_folder skips such files one by one, scripts/test.py:66-71): a
    short fmt chunk, a 4-bit ADPCM-style header, a truncated RIFF."""
    import struct
    from aware_amd.utils.audio import io

    def riff(fmt_chunk, body=b"\x00" * 64):
        payload = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt_chunk)) + fmt_chunk + b"data" + struct.pack("<I", len(body)) + body
        return b"RIFF" + struct.pack("<I", len(payload)) + payload

    cases = {
        "short_fmt.wav": riff(struct.pack("<HHIIH", 1, 1, 16000, 32000, 2)),                    # 14-byte fmt chunk
        "adpcm4.wav": riff(struct.pack("<HHIIHH", 2, 1, 16000, 8000, 256, 4)),                   # 4 bits per sample
        "zero_bits.wav": riff(struct.pack("<HHIIHH", 1, 1, 16000, 0, 0, 0)),
        "truncated.wav": b"RIFF\x10\x00\x00\x00WA",
    }
    for name, blob in cases.items():
        path = tmp_path / name
        path.write_bytes(blob)
        with pytest.raises(ValueError):
            io.read_wav(str(path))
