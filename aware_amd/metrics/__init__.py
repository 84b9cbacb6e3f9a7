from .audio import BER, SNR, STOI, snr_batch, stoi

__all__ = ["BER", "SNR", "STOI", "snr_batch", "stoi"]
