"""Factory: `load()` builds the (embedder, detector) pair from cards/config.yaml."""
from .load_model import load  # noqa: F401

__all__ = ("load",)
