from .load_model import load

__all__ = ["load"]
