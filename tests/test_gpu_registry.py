"""The reference's third seam on the device: optimiser / scheduler registries (embedding/optimizers.py:3-20,
schedulers.py:3-16) -- element-wise updates against torch.optim on the CPU (what the reference runs), embed sessions against
the oracle's registry loop."""
import numpy as np
import pytest
import torch

from conftest import make_clip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    from aware_amd import runtime
    from aware_amd._lib import require_gpu
    require_gpu()
    return runtime


TORCH_OPT = {"adam": torch.optim.Adam, "nadam": torch.optim.NAdam, "sgd": torch.optim.SGD, "rmsprop": torch.optim.RMSprop,
             "adagrad": torch.optim.Adagrad, "adadelta": torch.optim.Adadelta, "adamax": torch.optim.Adamax, "adamw": torch.optim.AdamW}
SCH = torch.optim.lr_scheduler
TORCH_SCHED = {"cosine_annealing": SCH.CosineAnnealingLR, "cosine_annealing_warm_restarts": SCH.CosineAnnealingWarmRestarts,
               "step": SCH.StepLR, "multi_step": SCH.MultiStepLR, "exponential": SCH.ExponentialLR, "cyclic": SCH.CyclicLR}

CASES = [
    ("nadam", {"lr": 0.1}, None, None),
    ("nadam", {"lr": 0.05, "weight_decay": 0.01, "momentum_decay": 0.01}, "step", {"step_size": 5, "gamma": 0.7}),
    ("adam", {"lr": 0.02}, None, None),
    ("adam", {"lr": 0.02, "betas": (0.8, 0.99), "weight_decay": 0.05, "eps": 1e-6}, "cosine_annealing", {"T_max": 12, "eta_min": 1e-4}),
    ("adamw", {"lr": 0.03, "weight_decay": 0.1}, "exponential", {"gamma": 0.95}),
    ("sgd", {"lr": 0.05}, None, None),
    ("sgd", {"lr": 0.05, "momentum": 0.9, "nesterov": True, "weight_decay": 0.01}, "multi_step", {"milestones": [4, 11], "gamma": 0.5}),
    ("sgd", {"lr": 0.05, "momentum": 0.7, "dampening": 0.2}, "cyclic", {"base_lr": 0.01, "max_lr": 0.08, "step_size_up": 4}),
    ("rmsprop", {"lr": 0.01, "alpha": 0.95}, "cosine_annealing_warm_restarts", {"T_0": 6, "T_mult": 2}),
    ("adagrad", {"lr": 0.1, "lr_decay": 0.05}, None, None),
    ("adamax", {"lr": 0.02}, "cyclic", {"base_lr": 0.005, "max_lr": 0.05, "step_size_up": 3, "step_size_down": 5}),
    ("adadelta", {"lr": 1.0, "rho": 0.85}, None, None),
]


@pytest.mark.parametrize("name,params,sched,sparams", CASES)
def test_registry_optimizers_match_torch_optim(rt, name, params, sched, sparams):
    """25 steps of every optimiser the reference's registry can run, with and without a learning-rate scheduler, as ONE
    device kernel per step (aware_opt_clamp_step: update + clamp) against torch.optim + torch.clamp on the CPU with the same
    gradients (embedding/multibit_embedder.py:85-86,112-117).  Tolerance: a few f32 roundings per step (the CPU kernels
    fuse / order a handful of operations differently)."""
    n, steps = 5000, 25
    g = torch.Generator().manual_seed(len(name) + steps)
    p0 = torch.rand(n, generator=g) * 2 + 0.1
    lo, hi = p0 * 0.5, p0 * 1.5
    grads = [torch.randn(n, generator=g) * (0.5 + 0.1 * i) * 1e-2 for i in range(steps)]
    ref = p0.clone().requires_grad_(True)
    topt = TORCH_OPT[name]([ref], **params)
    tsch = TORCH_SCHED[sched](topt, **sparams) if sched else None
    mine = p0.clone().cuda()
    oc = rt.OptClamp(mine, name, steps, sched, sparams, **params)
    dlo, dhi = lo.cuda(), hi.cuda()
    worst = 0.0
    for i in range(steps):
        topt.zero_grad()
        ref.grad = grads[i].clone()
        topt.step()
        if tsch is not None:
            tsch.step()
        with torch.no_grad():
            ref.data = torch.clamp(ref.data, lo, hi)
        oc.step(grads[i].cuda(), dlo, dhi)
        err = ((mine.cpu() - ref.detach()).abs() / ref.detach().abs()).max().item()
        worst = max(worst, err)
    moved = (ref.detach() - p0).abs().max().item()
    print(f"{name} + {sched}: max relative difference over {steps} steps {worst:.2e}; parameters moved by up to {moved:.2e}")
    assert moved > 1e-3
    assert worst < 3e-6, worst


@pytest.fixture(scope="module")
def O():
    from oracle import aware_oracle
    return aware_oracle


EMBED_CASES = [
    ("adam", {"lr": 0.05}, "cosine_annealing", {"T_max": 30}),
    ("sgd", {"lr": 20.0, "momentum": 0.9}, "step", {"step_size": 10, "gamma": 0.5}),
    ("adamw", {"lr": 0.05, "weight_decay": 0.001}, "exponential", {"gamma": 0.97}),
    ("nadam", {"lr": 0.1}, "reduce_lr_on_plateau", {"factor": 0.5, "patience": 1, "threshold": 0.1}),
    ("rmsprop", {"lr": 0.02}, "reduce_lr_on_plateau", {"factor": 0.9, "patience": 500}),
]


@pytest.mark.parametrize("name,params,sched,sparams", EMBED_CASES)
def test_embed_session_with_registry_optimizer(rt, O, name, params, sched, sparams):
    """Embed sessions configured through the registries (AWAREEmbedder(optimizer_cfg=..., scheduler_cfg=...), the YAML strings of
    cards/config.yaml:17-26) against the oracle's registry loop (torch.optim on the CPU as the reference runs it, one clip at a
    time): 30 iterations of two clips, per-step losses within the 20-step drift band, the per-clip learning rates of a FIRING
    ReduceLROnPlateau equal to torch's."""
    from aware_amd.embedding import AWAREEmbedder
    steps = 30
    emb = AWAREEmbedder(num_iterations=steps, optimizer_cfg={"name": name, "params": params},
                        scheduler_cfg={"name": sched, "params": sparams}, loss="push_extremes", verbose=False, use_graph=True)
    lengths = [16000, 20000]
    pairs = [make_clip(400 + i, n) for i, n in enumerate(lengths)]
    wm = np.stack([O.bits_to_bipolar(p[1]) for p in pairs]).astype(np.float32)
    batch = rt.Batch(lengths)
    sess = emb.start_session(batch, 16000)
    sess.begin(batch.pack([p[0] for p in pairs]), torch.from_numpy(wm).cuda())
    mine, lrs = [], []
    for _ in range(steps):
        if sched == "reduce_lr_on_plateau" and sparams["patience"] < steps:
            lrs.append(sess.clip_learning_rates().copy())
        sess.iterate(1)
        mine.append(sess.loss.cpu().numpy().copy())
    mine = np.stack(mine)
    for i, (clip, _) in enumerate(pairs):
        ref = O.Embedder(num_iterations=steps, optimizer=name, optimizer_params=params, scheduler=sched, scheduler_params=sparams)
        rl, rlr = [], []
        ref.embed_registry(clip[None], wm[i][None], record=lambda it, l, lr: (rl.append(l), rlr.append(lr)))
        d = np.abs(mine[:, i] - np.asarray(rl))
        print(f"{name} + {sched}, clip {i}: |loss - oracle| step0 {d[0]:.1e} max {d.max():.2e}; loss {rl[0]:.4f} -> {rl[-1]:.4f}; "
              f"lr {rlr[0]:.4g} -> {rlr[-1]:.4g}")
        assert d[0] < 2e-5 and d.max() < 3 * 1.9e-3, d
        assert rl[-1] < rl[0] - 0.01                                   # the optimiser works on the loss
        if lrs:
            np.testing.assert_allclose(np.asarray(lrs)[:, i], np.asarray(rlr), rtol=1e-12)
            assert rlr[-1] < rlr[0]                                    # ... and the scheduler fired
