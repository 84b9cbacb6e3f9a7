"""Small host helpers with the reference's names (src/AWARE/utils/utils.py:5-23)."""
from pathlib import Path

import numpy as np
import torch
import yaml


def load_config(config_path) -> dict:
    """Parse a YAML card; any failure surfaces as RuntimeError, as in the reference."""
    try:
        return yaml.safe_load(Path(config_path).read_text())
    except Exception as exc:
        raise RuntimeError(f"Error loading config from {config_path}: {exc}") from exc


def to_tensor(data):
    """float32 tensor view of an ndarray; tensors pass through untouched (same object, which is why
    the reference's `coeffs` aliases its initial value); anything else maps to None."""
    if isinstance(data, torch.Tensor):
        return data
    if isinstance(data, np.ndarray):
        return torch.as_tensor(np.ascontiguousarray(data)).to(torch.float32)
    return None
