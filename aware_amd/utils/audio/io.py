"""WAV file input/output for the harness ("audio I/O").

The reference reads clips with `librosa.load(path, sr=None, mono=True)` (`scripts/test.py:52`) and writes with
`soundfile.write` (`scripts/attacks.py:124`); neither library exists offline, so this module reads and writes RIFF/WAVE
itself: PCM 8 (unsigned) / 16 / 24 / 32 bit, IEEE float 32 / 64, plain and WAVE_FORMAT_EXTENSIBLE headers.  Integer samples
are scaled the way libsndfile (the backend of both libraries) scales them: int / 2^(bits-1).  Other containers (mp3,
flac, ...) are out of scope and raise ValueError."""
import struct

import numpy as np

_PCM, _FLOAT, _EXTENSIBLE = 1, 3, 0xFFFE


def read_wav(path):
    """Returns (samples float32 [N] or [N, channels], sample_rate)."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, body = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        chunk = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            if len(chunk) < 16:
                raise ValueError(f"{path}: fmt chunk of {len(chunk)} bytes (16 needed)")
            tag, ch, sr, _, align, bits = struct.unpack_from("<HHIIHH", chunk, 0)
            if tag == _EXTENSIBLE and len(chunk) >= 26:
                tag = struct.unpack_from("<H", chunk, 24)[0]          # first two bytes of the sub-format GUID
            fmt = (tag, ch, sr, align, bits)
        elif cid == b"data":
            body = chunk
        pos += 8 + size + (size & 1)
    if fmt is None or body is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    tag, ch, sr, align, bits = fmt
    if ch < 1:
        raise ValueError(f"{path}: no channels")
    if bits not in (8, 16, 24, 32, 64):                       # (ADPCM and other sub-byte codings: not PCM samples)
        raise ValueError(f"{path}: unsupported WAVE format tag {tag} with {bits} bits")
    nbytes = bits // 8
    n = len(body) // (nbytes * ch)
    body = body[:n * nbytes * ch]
    if tag == _FLOAT and bits in (32, 64):
        x = np.frombuffer(body, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    elif tag == _PCM and bits == 8:
        x = (np.frombuffer(body, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif tag == _PCM and bits == 16:
        x = np.frombuffer(body, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == _PCM and bits == 24:
        b = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = v.astype(np.float32) / 8388608.0
    elif tag == _PCM and bits == 32:
        x = (np.frombuffer(body, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag} with {bits} bits")
    return (x if ch == 1 else x.reshape(n, ch)), int(sr)


def load(path, sr=None, mono=True):
    """`librosa.load(path, sr=None, mono=True)` for WAV files: float32, channels averaged.  A target rate other than the
    file's is refused (the harness resamples with the polyphase kernel, `aware_amd.attacks.resample_poly_batch`)."""
    x, file_sr = read_wav(path)
    if mono and x.ndim == 2:
        x = x.mean(axis=1).astype(np.float32)
    if sr is not None and int(sr) != file_sr:
        raise ValueError(f"{path}: file rate {file_sr} != requested {sr}; resample with resample_poly_batch")
    return x, file_sr


def write_wav(path, audio, sample_rate, subtype="PCM_16"):
    """`soundfile.write(path, audio, sr)` for WAV: audio [N] or [N, channels] float in [-1, 1).  subtype PCM_16 (soundfile's
    default for .wav: round(x * 32768) clipped) or FLOAT (32-bit IEEE)."""
    a = np.asarray(audio)
    if a.ndim not in (1, 2):
        raise ValueError("audio must be [N] or [N, channels]")
    ch = 1 if a.ndim == 1 else a.shape[1]
    if subtype == "PCM_16":
        v = np.clip(np.rint(a.astype(np.float64) * 32768.0), -32768, 32767).astype("<i2")
        tag, bits = _PCM, 16
    elif subtype == "FLOAT":
        v = a.astype("<f4")
        tag, bits = _FLOAT, 32
    else:
        raise ValueError(f"unsupported subtype {subtype}")
    body = v.tobytes()
    align = ch * bits // 8
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(body), b"WAVE", b"fmt ", 16, tag, ch, int(sample_rate),
                      int(sample_rate) * align, align, bits, b"data", len(body))
    with open(path, "wb") as f:
        f.write(hdr + body)
