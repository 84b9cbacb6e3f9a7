"""detect_watermark (reference: src/AWARE/service/detect.py:7-55): 16 kHz only, mono 1-D or
stereo [N,2] (per bit, the channel with the larger |value| wins, :23-37), then PatternDecoder."""
import numpy as np

from ..utils.logger import logger
from ..utils.watermark import PatternDecoder


def detect_watermark(audio: np.ndarray, sample_rate: int, detector):
    decode = PatternDecoder(encoder_mode=detector.pattern_mode, threshold=detector.threshold)
    if sample_rate != 16000:
        logger.error(f"Invalid sample rate. Expected 16000Hz, got {sample_rate}Hz.")
        raise ValueError("Invalid sample rate. Expected 16000Hz.")
    audio = np.asarray(audio)
    if audio.ndim == 2 and audio.shape[1] == 2:
        vals = detector.detect_batch([audio[:, 0].astype(np.float32), audio[:, 1].astype(np.float32)], sample_rate)
        l, r = vals[0].cpu().numpy(), vals[1].cpu().numpy()
        return decode(np.where(np.abs(l) > np.abs(r), l, r))
    if audio.ndim == 1:
        return decode(detector.detect(audio, sample_rate))
    logger.error("Invalid audio shape. Expected 1D or 2D numpy array.")
    raise ValueError("Invalid audio shape. Expected 1D or 2D numpy array.")


def detect_watermark_batch(clips, sample_rate: int, detector):
    if sample_rate != 16000:
        raise ValueError("Invalid sample rate. Expected 16000Hz.")
    decode = PatternDecoder(encoder_mode=detector.pattern_mode, threshold=detector.threshold)
    vals = detector.detect_batch([np.asarray(c, dtype=np.float32) for c in clips], sample_rate).cpu().numpy()
    return [decode(v) for v in vals]
