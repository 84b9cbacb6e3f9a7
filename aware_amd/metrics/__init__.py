from .audio import BER, SNR

__all__ = ["BER", "SNR"]
