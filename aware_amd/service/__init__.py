from .embed import embed_watermark, embed_watermark_batch
from .detect import detect_watermark, detect_watermark_batch

__all__ = ["embed_watermark", "detect_watermark", "embed_watermark_batch", "detect_watermark_batch"]
