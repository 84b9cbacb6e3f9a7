"""aware_amd -- MI355X-native hot path of the AWARE audio watermark (embed -> attack -> detect).

Host-side mirror of deepmarkpy/aware's public surface over libaware_hip.so (HIP, gfx950).
"""
__version__ = "0.1.0"
