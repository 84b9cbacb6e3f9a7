"""Scheduler registry (reference: src/AWARE/embedding/schedulers.py:3-16).

Six of the seven registered schedulers change the learning rate as a function of the step count alone: they are attached to
the stand-in torch optimiser of optimizers.get_optimizer, stepped once per iteration on the host, and their rates are baked
into the per-step table the device applies (optimizers.step_table).  ReduceLROnPlateau depends on the loss -- which differs
per clip in a batch -- so its state lives on the device, one per clip (csrc/seam_kernels.hip::plateau_kernel, torch's
arithmetic in double); the model card's patience of 500 cannot fire inside 400 iterations (cards/config.yaml:21-26), which is
the constant-rate case the fused NAdam epilogue serves."""
from __future__ import annotations

import torch

_REFERENCE = {
    "reduce_lr_on_plateau": torch.optim.lr_scheduler.ReduceLROnPlateau,
    "cosine_annealing": torch.optim.lr_scheduler.CosineAnnealingLR,
    "cosine_annealing_warm_restarts": torch.optim.lr_scheduler.CosineAnnealingWarmRestarts,
    "step": torch.optim.lr_scheduler.StepLR,
    "multi_step": torch.optim.lr_scheduler.MultiStepLR,
    "exponential": torch.optim.lr_scheduler.ExponentialLR,
    "cyclic": torch.optim.lr_scheduler.CyclicLR,
}


def get_scheduler(name: str, optimizer: dict, num_iterations: int, **kwargs) -> dict:
    """optimizer: the dict of optimizers.get_optimizer.  Returns {"name", "constant_lr", "torch" (a table scheduler attached
    to the stand-in optimiser, or None), "plateau" (parameters of the per-clip device state, or None)}."""
    if name not in _REFERENCE:
        raise ValueError(f"Scheduler {name} not found")                        # schedulers.py:14-15
    sched = _REFERENCE[name](optimizer["torch"], **kwargs)                     # torch validates the arguments
    if name != "reduce_lr_on_plateau":
        return {"name": name, "constant_lr": False, "torch": sched, "plateau": None}
    if sched.mode != "min" or sched.threshold_mode != "rel" or sched.cooldown != 0:
        raise NotImplementedError("ReduceLROnPlateau: only mode='min', threshold_mode='rel', cooldown=0 are on the HIP path")
    never_fires = sched.patience + 1 >= num_iterations or sched.factor == 1.0
    plateau = None if never_fires else {"factor": float(sched.factor), "patience": int(sched.patience),
                                        "threshold": float(sched.threshold), "min_lr": float(sched.min_lrs[0]),
                                        "eps": float(sched.eps)}
    return {"name": name, "constant_lr": never_fires, "torch": None, "plateau": plateau}
