from .logger import logger
from .utils import load_config, to_tensor

__all__ = ["logger", "load_config", "to_tensor"]
