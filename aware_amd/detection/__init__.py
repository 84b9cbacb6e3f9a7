"""Frozen detector network (device weights + HIP forward) and the detect() wrapper."""
from .mel import mel_filter_bank
from .multibit_detector_net import AWAREDetectorNet, DETECTOR_SEED
from .multibit_detector import AWAREDetector

__all__ = ("AWAREDetector", "AWAREDetectorNet", "DETECTOR_SEED", "mel_filter_bank")
