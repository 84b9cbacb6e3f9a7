"""Scheduler registry (reference: src/AWARE/embedding/schedulers.py:3-16).

The model card's ReduceLROnPlateau(factor 0.9, patience 500) cannot fire inside 400 iterations
(cards/config.yaml:21-26), so the learning rate is constant; that case is accepted, anything that
would change the rate is refused rather than silently ignored."""

_REFERENCE_NAMES = ("reduce_lr_on_plateau", "cosine_annealing", "cosine_annealing_warm_restarts", "step",
                    "multi_step", "exponential", "cyclic")


def get_scheduler(name: str, num_iterations: int, **kwargs) -> dict:
    if name not in _REFERENCE_NAMES:
        raise ValueError(f"Scheduler {name} not found")
    if name == "reduce_lr_on_plateau":
        patience = kwargs.get("patience", 10)
        if patience + 1 >= num_iterations or kwargs.get("factor", 0.1) == 1.0:
            return {"name": name, "constant_lr": True}
        raise NotImplementedError("ReduceLROnPlateau with patience < num_iterations is not on the HIP path")
    raise NotImplementedError(f"scheduler '{name}' is registered by the reference but not implemented on the HIP path")
