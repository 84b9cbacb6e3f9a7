"""Optimizer registry (reference: src/AWARE/embedding/optimizers.py:3-20).

The reference maps ten names to torch.optim classes and runs `get_optimizer(name, [coeffs], **params).step()` on its CPU
tensors every iteration (multibit_embedder.py:85,112).  Here the update runs on the device: the model card's NAdam fused in
the epilogue of the adjoint kernel, every other registered optimiser as one element-wise launch per iteration
(csrc/dsp_args.hpp::opt_clamp_update -- the arithmetic of torch's single-tensor implementations).  This module is the host
half: it validates the YAML arguments the way the reference does (by constructing the torch class, so an unknown keyword
raises the same TypeError), and turns (optimizer, scheduler) into the table of per-step scalars the device applies.

`sparse_adam` and `lbfgs` cannot run in the reference's loop either (torch.optim.SparseAdam refuses dense gradients,
LBFGS.step needs a closure the loop does not pass): they are recognised and refused."""
from __future__ import annotations

import math

import numpy as np
import torch

_REFERENCE = {
    "adam": torch.optim.Adam, "nadam": torch.optim.NAdam, "sgd": torch.optim.SGD, "rmsprop": torch.optim.RMSprop,
    "adagrad": torch.optim.Adagrad, "adadelta": torch.optim.Adadelta, "adamax": torch.optim.Adamax, "adamw": torch.optim.AdamW,
    "sparse_adam": torch.optim.SparseAdam, "lbfgs": torch.optim.LBFGS,
}
KINDS = {"nadam": 0, "adam": 1, "adamw": 2, "sgd": 3, "rmsprop": 4, "adagrad": 5, "adamax": 6, "adadelta": 7}


def get_optimizer(name: str, params=None, **kwargs) -> dict:
    """{"name", "kind", "torch": a torch.optim instance on a one-element stand-in parameter (carries the validated
    hyper-parameters and is what a scheduler attaches to), "group": its param_group}."""
    if name not in _REFERENCE:
        raise ValueError(f"Optimizer {name} not found")                        # optimizers.py:17-18
    if name == "sparse_adam":
        raise NotImplementedError("sparse_adam: torch.optim.SparseAdam does not support dense gradients -- the reference's loop "
                                  "fails with it as well")
    if name == "lbfgs":
        raise NotImplementedError("lbfgs: torch.optim.LBFGS.step needs a closure -- the reference's loop (optimizer.step() "
                                  "without one) fails with it as well")
    stand_in = torch.nn.Parameter(torch.zeros(1))
    opt = _REFERENCE[name]([stand_in], **kwargs)                               # TypeError / ValueError on bad arguments, like torch
    g = opt.param_groups[0]
    unsupported = {"amsgrad": False, "maximize": False, "centered": False, "decoupled_weight_decay": name == "adamw",
                   "differentiable": False, "capturable": False}
    for key, allowed in unsupported.items():
        if key in g and bool(g[key]) != allowed:
            raise NotImplementedError(f"optimizer '{name}': {key}={g[key]} is not on the HIP path")
    if name == "rmsprop" and g["momentum"] != 0:
        raise NotImplementedError("rmsprop with momentum is not on the HIP path")
    if name == "adagrad" and g.get("initial_accumulator_value", 0) != 0:
        raise NotImplementedError("adagrad with an initial accumulator value is not on the HIP path")
    return {"name": name, "kind": KINDS[name], "torch": opt, "group": g}


def hyper_parameters(opt: dict) -> tuple[list, float]:
    """(hyp[8] of opt_clamp_update, decoupled weight decay)."""
    g, name = opt["group"], opt["name"]
    h = [0.0] * 8
    wd_decoupled = 0.0
    if name in ("nadam", "adam", "adamw", "adamax"):
        b1, b2 = g["betas"]
        h[0], h[1], h[2], h[3] = 1.0 - b1, b2, 1.0 - b2, g["eps"]
    elif name == "sgd":
        h[0], h[5], h[6] = g["momentum"], float(bool(g["nesterov"])), 1.0 - g["dampening"]
    elif name == "rmsprop":
        h[1], h[2], h[3] = g["alpha"], 1.0 - g["alpha"], g["eps"]
    elif name == "adagrad":
        h[3] = g["eps"]
    elif name == "adadelta":
        h[1], h[2], h[3] = g["rho"], 1.0 - g["rho"], g["eps"]
    if name == "adamw":
        wd_decoupled = float(g["weight_decay"])
    else:
        h[4] = float(g["weight_decay"])
    return h, wd_decoupled


def step_table(opt: dict, num_iterations: int, scheduler=None) -> np.ndarray:
    """[num_iterations][5] doubles (ux, uy, z, lr_t, h0_t): the scalars torch's single-tensor step computes from the step
    count and the param_group at step t, with the learning rate divided out of ux / uy.  `scheduler`: a torch LR scheduler
    attached to opt["torch"] and stepped once per iteration after the optimiser, like the reference's loop
    (multibit_embedder.py:112-113), or None for a constant rate (also the ReduceLROnPlateau case: its rate is per clip and
    lives on the device)."""
    g, name, topt = opt["group"], opt["name"], opt["torch"]
    tab = np.zeros((num_iterations, 5), dtype=np.float64)
    mu_product = np.float32(1.0)                                   # torch keeps NAdam's mu_product in a float32 tensor
    for t in range(1, num_iterations + 1):
        lr = float(g["lr"])
        ux = uy = 0.0
        z = 1.0
        h0 = -1.0
        if name in ("nadam", "adam", "adamw", "adamax"):
            b1, b2 = g["betas"]
            h0 = 1.0 - b1
        if name == "nadam":
            md = g["momentum_decay"]
            mu = b1 * (1.0 - 0.5 * (0.96 ** (t * md)))
            mu_next = b1 * (1.0 - 0.5 * (0.96 ** ((t + 1) * md)))
            mu_product = np.float32(mu_product * np.float32(mu))
            mp = float(mu_product)
            ux, uy, z = -(1.0 - mu) / (1.0 - mp), -mu_next / (1.0 - mp * mu_next), 1.0 - b2 ** t
        elif name in ("adam", "adamw"):
            ux, z = -1.0 / (1.0 - b1 ** t), (1.0 - b2 ** t) ** 0.5
        elif name == "adamax":
            ux = -1.0 / (1.0 - b1 ** t)
        elif name == "sgd":
            ux, uy, h0 = -1.0, 1.0 if t == 1 else 0.0, float(g["momentum"])
        elif name == "adagrad":
            ux = -1.0 / (1.0 + (t - 1) * g["lr_decay"])
        else:                                                       # rmsprop, adadelta
            ux = -1.0
        tab[t - 1] = (ux, uy, z, lr, h0)
        if scheduler is not None:
            topt.step()                                             # (a no-op on the stand-in: it has no gradient)
            scheduler.step()
    return tab


def step_scalars(opt: dict, tab: np.ndarray, t: int, lr: float | None = None):
    """(coef4, hyp8) of step t (1-based) as float32 arrays for aware_opt_clamp_step."""
    ux, uy, z, lr_t, h0 = tab[t - 1]
    lr = lr_t if lr is None else lr
    h, wd = hyper_parameters(opt)
    if h0 >= 0:
        h[0] = h0
    cy = uy if opt["name"] == "sgd" else lr * uy
    return (np.asarray([lr * ux, cy, z, 1.0 - lr * wd], dtype=np.float32), np.asarray(h, dtype=np.float32))


def is_card_default(opt: dict) -> bool:
    """NAdam without weight decay: the optimiser the fused epilogue implements (its table is built by the library itself)."""
    g = opt["group"]
    return opt["name"] == "nadam" and g["weight_decay"] == 0 and not math.isnan(g["lr"])
