from .audio import BER, SNR, snr_batch

__all__ = ["BER", "SNR", "snr_batch"]
