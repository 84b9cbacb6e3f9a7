"""Watermark pattern codec: bits / bytes <-> bipolar, thresholding.

Reference: src/AWARE/utils/watermark/encoder.py:5-58, decoder.py:4-69.  Integer work stays on
the host (20 values per clip)."""
import numpy as np

from ...interfaces import BasePatternProcessor

_ENC_MODES = ("bits2bipolar", "bytes2bipolar", "bytes2bits", "bits")


def _unpack(data: bytes) -> np.ndarray:
    # MSB first, as format(b, '08b') does (encoder.py:31-33)
    return np.unpackbits(np.frombuffer(bytes(data), dtype=np.uint8)).astype(np.int32)


class PatternEncoder(BasePatternProcessor):
    def __init__(self, mode: str = "bits2bipolar"):
        self.mode = mode

    def __call__(self, inputs):
        if self.mode == "bits2bipolar":
            return (2 * np.asarray(inputs, dtype=np.int64) - 1).astype(np.int32)
        if self.mode == "bytes2bipolar":
            return (2 * _unpack(inputs) - 1).astype(np.int32)
        if self.mode == "bytes2bits":
            return _unpack(inputs)
        if self.mode == "bits":
            return inputs
        raise ValueError(f"Invalid mode: {self.mode}")


class PatternDecoder(BasePatternProcessor):
    def __init__(self, threshold: float = 0.5, encoder_mode: str = "bits2bipolar"):
        self.threshold = threshold
        self.encoder_mode = encoder_mode

    @staticmethod
    def _bipolar(values, threshold):
        return 2 * (np.asarray(values) > threshold).astype(np.int32) - 1

    def __call__(self, detected_values):
        m = self.encoder_mode
        if m == "bits2bipolar":
            return (self._bipolar(detected_values, self.threshold) > 0).astype(np.int32)
        if m == "bytes2bipolar":
            # the reference emits one byte per bit here (decoder.py:53-57); kept as is
            return bytes(int(b) for b in (self._bipolar(detected_values, self.threshold) > 0))
        if m == "bytes2bits":
            return bytes(int(b) for b in (np.asarray(detected_values) > self.threshold))
        if m == "bits":
            return (np.asarray(detected_values) > self.threshold).astype(np.int32)
        raise ValueError(f"Invalid mode: {m}")
