"""load(): build (embedder, detector) from cards/config.yaml.

Reference: src/AWARE/utils/models/load_model.py:6-76 -- every key has a default, the detector
shares the embedder's detection_net (:56), and any failure is logged and turned into `None`."""
from pathlib import Path

from ..logger import logger
from ..utils import load_config


def load(config_path=None):
    from ...detection import AWAREDetector
    from ...embedding import AWAREEmbedder

    path = Path(config_path) if config_path else Path(__file__).resolve().parents[2] / "cards" / "config.yaml"
    try:
        cfg = load_config(path)
    except Exception as exc:
        logger.error(f"Error loading configs: {exc}")
        return None
    try:
        embedder = AWAREEmbedder(
            frame_length=cfg.get("frame_length", 1024), hop_length=cfg.get("hop_length", 256),
            window=cfg.get("window", "hann"), win_length=cfg.get("win_length", 1024),
            pattern_mode=cfg.get("pattern_mode", "bits2bipolar"),
            embedding_bands=tuple(cfg.get("embedding_bands", [500, 4000])),
            tolerance_db=cfg.get("tolerance_db", 6.0), num_iterations=cfg.get("num_iterations", 400),
            detection_net_cfg=cfg.get("detection_net_cfg", {}),
            optimizer_cfg=cfg.get("optimizer_cfg", {"name": "nadam", "params": {"lr": 0.1}}),
            scheduler_cfg=cfg.get("scheduler_cfg", {"name": "reduce_lr_on_plateau", "params": {"factor": 0.9, "patience": 500}}),
            loss=cfg.get("loss", "push_extremes"), verbose=cfg.get("verbose", True))
    except Exception as exc:
        logger.error(f"Error creating embedder: {exc}")
        return None
    try:
        detector = AWAREDetector(
            model=embedder.detection_net, threshold=cfg.get("threshold", 0.0),
            frame_length=cfg.get("frame_length", 1024), hop_length=cfg.get("hop_length", 256),
            window=cfg.get("window", "hann"), win_length=cfg.get("win_length", 1024),
            pattern_mode=cfg.get("pattern_mode", "bipolar"),
            embedding_bands=tuple(cfg.get("embedding_bands", [500, 4000])))
    except Exception as exc:
        logger.error(f"Error creating detector: {exc}")
        return None
    return embedder, detector
