"""load(): (embedder, detector) from the YAML model card.

Reference: src/AWARE/utils/models/load_model.py:6-76.  Behaviour kept: every key is optional with the
reference's default, the detector is built around the embedder's own detection_net (:56), and a
failure at any stage is logged and reported as `None` instead of raising (:15-17, :45-49, :69-73)."""
from pathlib import Path

from ..logger import logger
from ..utils import load_config

_CARD = Path(__file__).resolve().parents[2] / "cards" / "config.yaml"

# constructor argument -> (card key, default); tuples for the band edges like the reference
_SHARED = {"frame_length": ("frame_length", 1024), "hop_length": ("hop_length", 256),
           "window": ("window", "hann"), "win_length": ("win_length", 1024)}
_EMBEDDER = {"pattern_mode": ("pattern_mode", "bits2bipolar"), "tolerance_db": ("tolerance_db", 6.0),
             "num_iterations": ("num_iterations", 400), "detection_net_cfg": ("detection_net_cfg", {}),
             "optimizer_cfg": ("optimizer_cfg", {"name": "nadam", "params": {"lr": 0.1}}),
             "scheduler_cfg": ("scheduler_cfg", {"name": "reduce_lr_on_plateau", "params": {"factor": 0.9, "patience": 500}}),
             "loss": ("loss", "push_extremes"), "verbose": ("verbose", True)}
_DETECTOR = {"threshold": ("threshold", 0.0), "pattern_mode": ("pattern_mode", "bipolar")}


def _pick(card: dict, table: dict) -> dict:
    return {arg: card.get(key, default) for arg, (key, default) in table.items()}


def _stage(what, fn):
    try:
        return fn()
    except Exception as exc:                      # the reference swallows everything here
        logger.error(f"Error {what}: {exc}")
        return None


def load(config_path=None):
    from ...detection import AWAREDetector
    from ...embedding import AWAREEmbedder

    card = _stage("loading configs", lambda: load_config(config_path or _CARD))
    if card is None:
        return None
    bands = tuple(card.get("embedding_bands", [500, 4000]))
    embedder = _stage("creating embedder", lambda: AWAREEmbedder(
        embedding_bands=bands, **_pick(card, _SHARED), **_pick(card, _EMBEDDER)))
    if embedder is None:
        return None
    detector = _stage("creating detector", lambda: AWAREDetector(
        model=embedder.detection_net, embedding_bands=bands, **_pick(card, _SHARED), **_pick(card, _DETECTOR)))
    if detector is None:
        return None
    return embedder, detector
