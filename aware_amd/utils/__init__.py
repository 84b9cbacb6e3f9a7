"""Host-side helpers: logger, YAML card loading, ndarray -> tensor."""
from .utils import to_tensor, load_config
from .logger import logger

__all__ = ["load_config", "to_tensor", "logger"]
