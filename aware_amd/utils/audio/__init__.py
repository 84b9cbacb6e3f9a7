from .plugins import (STFT, ISTFT, STFTDecomposer, STFTAssembler, STFTNormalizer, WaveformNormalizer,
                      SilenceChecker, band_bins, get_plan, default_plan, get_batch)

__all__ = ["STFT", "STFTDecomposer", "STFTAssembler", "ISTFT", "WaveformNormalizer", "SilenceChecker",
           "STFTNormalizer", "band_bins", "get_plan", "default_plan", "get_batch"]
