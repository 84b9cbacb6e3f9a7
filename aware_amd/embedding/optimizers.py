"""Optimizer registry (reference: src/AWARE/embedding/optimizers.py:3-20).

The reference maps ten names to torch.optim classes; its model card uses "nadam"
(cards/config.yaml:17-20).  The HIP path implements NAdam (torch single-tensor semantics) fused
with the box clamp and best-snapshot in the adjoint-analysis kernel's epilogue."""

_REFERENCE_NAMES = ("adam", "nadam", "sgd", "rmsprop", "adagrad", "adadelta", "adamax", "adamw", "sparse_adam", "lbfgs")
NADAM_DEFAULTS = {"lr": 2e-3, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "momentum_decay": 4e-3}


def get_optimizer(name: str, params=None, **kwargs) -> dict:
    """Validated hyper-parameter dict for the fused HIP optimiser step."""
    if name not in _REFERENCE_NAMES:
        raise ValueError(f"Optimizer {name} not found")
    if name != "nadam":
        raise NotImplementedError(f"optimizer '{name}' is registered by the reference but only 'nadam' runs on the HIP path")
    hp = dict(NADAM_DEFAULTS)
    unknown = set(kwargs) - set(hp)
    if unknown:
        raise TypeError(f"NAdam got unexpected arguments {sorted(unknown)}")
    hp.update(kwargs)
    if hp["weight_decay"] != 0:
        raise NotImplementedError("NAdam weight_decay != 0 is not on the HIP path")
    return hp
