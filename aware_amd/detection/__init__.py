from .multibit_detector import AWAREDetector
from .multibit_detector_net import AWAREDetectorNet

__all__ = ["AWAREDetector", "AWAREDetectorNet"]
