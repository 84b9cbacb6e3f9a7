from .codec import PatternEncoder, PatternDecoder

__all__ = ["PatternEncoder", "PatternDecoder"]
