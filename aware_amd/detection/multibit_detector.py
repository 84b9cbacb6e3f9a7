"""AWAREDetector.detect on the HIP path.

Reference: src/AWARE/detection/multibit_detector.py:9-42 -- normalise, STFT, magnitude, zero the
bins outside the embedding band, network forward.  Batched entry point: detect_batch."""
from __future__ import annotations

import numpy as np
import torch

from ..interfaces import BaseDetector
from ..utils.audio import STFT, STFTDecomposer, WaveformNormalizer, band_bins, get_plan
from .. import runtime as rt


class AWAREDetector(BaseDetector):
    def __init__(self, model, threshold: float = 0.0, frame_length: int = 1024, hop_length: int = 256,
                 window: str = "hann", win_length: int = 1024, pattern_mode: str = "bits2bipolar",
                 embedding_bands=(500, 4000)):
        self.threshold = threshold
        self.device = torch.device("cuda")
        self.pattern_mode = pattern_mode
        self.embedding_bands = tuple(embedding_bands)
        self.win_length = frame_length
        self.frame_length = frame_length
        self.hop_length = hop_length
        self.window = window
        self.detection_net = model
        self.audio_preprocess_pipeline = [WaveformNormalizer(), STFT(frame_length, hop_length, window, win_length),
                                          STFTDecomposer()]

    def _plan(self, sample_rate):
        return get_plan(self.frame_length, self.hop_length, self.window,
                        band_bins(sample_rate, self.frame_length, self.embedding_bands))

    def detect_batch(self, clips, sample_rate: int) -> torch.Tensor:
        """list of 1-D float arrays (any lengths) -> device tensor [B, n_bits] of raw values."""
        plan = self._plan(sample_rate)
        batch = rt.Batch([len(c) for c in clips])
        return rt.detect(plan, self.detection_net.device_weights(plan), batch, batch.pack(clips))

    def detect_device(self, audio: torch.Tensor, batch: "rt.Batch", sample_rate: int) -> torch.Tensor:
        plan = self._plan(sample_rate)
        return rt.detect(plan, self.detection_net.device_weights(plan), batch, audio)

    def detect(self, audio: np.ndarray, sample_rate: int) -> np.ndarray:
        vals = self.detect_batch([np.asarray(audio, dtype=np.float32)], sample_rate)
        return vals[0].detach().cpu().numpy()
