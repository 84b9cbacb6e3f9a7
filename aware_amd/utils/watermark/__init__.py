"""Watermark pattern codec (bits / bytes <-> bipolar)."""
from .codec import PatternDecoder, PatternEncoder  # noqa: F401

__all__ = ("PatternEncoder", "PatternDecoder")
