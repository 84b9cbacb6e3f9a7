from .audio import BER, PESQ, SNR, STOI, snr_batch, stoi

__all__ = ["BER", "PESQ", "SNR", "STOI", "snr_batch", "stoi"]
