"""load_config / to_tensor (reference: src/AWARE/utils/utils.py:5-23)."""
import numpy as np
import torch
import yaml


def load_config(config_path) -> dict:
    try:
        with open(config_path, "r") as fh:
            return yaml.safe_load(fh)
    except Exception as exc:
        raise RuntimeError(f"Error loading config from {config_path}: {exc}")


def to_tensor(data):
    """ndarray -> float32 tensor; a tensor is returned as is (the reference aliases it too)."""
    if isinstance(data, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(data)).float()
    if isinstance(data, torch.Tensor):
        return data
    return None
