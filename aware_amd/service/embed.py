"""embed_watermark: the public numpy API (reference: src/AWARE/service/embed.py:7-80).

Same checks and the same ValueErrors in the same order: 16 kHz only (:24-26), watermark length
== detector output_length (:32-34), mono [N] / [N,1] or stereo [N,2] (:37, :61, :75-77), silence
gate (:47-49, :65-67), rescale by the SIGNED maximum of the input (:69, :73).  Stereo is two
independent mono problems (:52-53) -- here they run as one batch of two clips."""
import numpy as np

from ..utils.audio import SilenceChecker
from ..utils.logger import logger
from ..utils.watermark import PatternEncoder

_SILENT_MSG = "Signal you provided doesn't contain any speach. Please provide signal that contains speach."


def _encode(watermark_bits, model):
    wm = PatternEncoder(mode=model.pattern_mode)(watermark_bits)
    if len(wm) != model.detection_net.output_length:
        logger.error(f"Invalid watermark length. Expected {model.detection_net.output_length}, got {len(wm)}.")
        raise ValueError("Invalid watermark length.")
    return wm


def embed_watermark(audio: np.ndarray, sample_rate: int, watermark_bits, model) -> np.ndarray:
    if sample_rate != 16000:
        logger.error(f"Invalid sample rate. Expected 16000Hz, got {sample_rate}Hz.")
        raise ValueError("Invalid sample rate. Expected 16000Hz.")
    wm = _encode(watermark_bits, model)
    gate = SilenceChecker(sample_rate=sample_rate)
    audio = np.asarray(audio)
    if audio.ndim == 2 and audio.shape[1] == 2:
        left, right = audio[:, 0], audio[:, 1]
        if gate(left) and gate(right):
            logger.error(_SILENT_MSG)
            raise ValueError(_SILENT_MSG)
        mx = np.array([np.max(left), np.max(right)], dtype=np.float32)
        outs = model.embed_batch([left.astype(np.float32), right.astype(np.float32)], sample_rate,
                                 np.stack([wm, wm]).astype(np.float32), rescale=mx)
        return np.column_stack([o.cpu().numpy() for o in outs])
    if audio.ndim == 1 or audio.shape[1] == 1:
        if gate(audio):
            logger.error(_SILENT_MSG)
            raise ValueError(_SILENT_MSG)
        audio_mx = np.max(audio)
        watermarked = model.embed(audio, sample_rate, wm)
        return audio_mx * watermarked
    logger.error("Invalid audio shape. Expected 1D or 2D numpy array.")
    raise ValueError("Invalid audio shape. Expected 1D or 2D numpy array.")


def embed_watermark_batch(clips, sample_rate: int, watermark_bits, model):
    """Batched form for the MI355X path: `clips` is a list of mono float arrays (ragged),
    `watermark_bits` one bit pattern per clip.  Returns a list of numpy arrays."""
    if sample_rate != 16000:
        raise ValueError("Invalid sample rate. Expected 16000Hz.")
    wms = np.stack([_encode(b, model) for b in watermark_bits]).astype(np.float32)
    for c in clips:
        if np.asarray(c).ndim != 1:
            raise ValueError("Invalid audio shape. Expected 1D or 2D numpy array.")
    mx = np.array([np.max(c) for c in clips], dtype=np.float32)
    outs = model.embed_batch([np.asarray(c, dtype=np.float32) for c in clips], sample_rate, wms, rescale=mx)
    return [o.cpu().numpy() for o in outs]
