"""EXTENSION -- data-parallel detector training step (BASELINE.json north_star / configs 3-4: "training loop ... RCCL
all-reduce over xGMI on the embedder/detector gradients").

The reference trains nothing: its detector is frozen and seed-initialised (src/AWARE/embedding/multibit_embedder.py:76-77,
src/AWARE/detection/multibit_detector_net.py:77-80).  This module therefore has no reference counterpart -- parity unpinned;
its gradients are specified by torch autograd on oracle/aware_oracle.py::Detector and tested against it
(tests/test_gpu_seam.py::test_detector_weight_gradients_extension).

One step, on every rank's own clips (shard by clip, as everywhere else):
    raw = detector(audio)                                        HIP: normalise -> STFT -> |.| -> network
    loss = mse(raw, bipolar bits) - 0.1 * mean|raw|             the reference's own objective (losses.py:38-42), per rank
    dL/dW, dL/db                                                 HIP: aware_detector_weight_gradients
    all-reduce (average) of the gradients over the ranks         RCCL, one flat 6.7 MB bucket (aware_amd/parallel.py)
    optimiser step on the host copy (torch.optim, plumbing)      identical on every rank -> weights stay in sync
    device copy refreshed                                        aware_detector_update
"""
from __future__ import annotations

import numpy as np
import torch

from . import parallel
from . import runtime as rt


class DetectorTrainer:
    def __init__(self, detector, lr: float = 1e-4, sample_rate: int = 16000):
        """detector: an AWAREDetector (aware_amd.detection); its network's host weights are trained in place."""
        self.detector = detector
        self.net = detector.detection_net
        self.sample_rate = sample_rate
        self.params = [torch.nn.Parameter(torch.from_numpy(np.array(w, dtype=np.float32)).cuda()) for w in self.net.weights] + \
                      [torch.nn.Parameter(torch.from_numpy(np.array(b, dtype=np.float32)).cuda()) for b in self.net.biases]
        self.opt = torch.optim.Adam(self.params, lr=lr)
        self.nl = len(self.net.weights)

    def step(self, audio: "rt.Ragged", bits: torch.Tensor):
        """audio: ragged device clips at 16 kHz (e.g. watermarked + attacked); bits [B, n_bits] 0/1.  Returns the loss
        averaged over ranks (float) and the raw detector outputs."""
        plan = self.detector._plan(self.sample_rate)
        dw = self.net.device_weights(plan)
        batch = rt.Batch(audio.lengths)
        data = audio.data if audio.data.dtype == torch.float32 else audio.data.float()
        mag, _ = rt.stft_band(plan, batch, data, normalize=True)
        target = (2 * bits - 1).to(torch.float32)
        # loss and its gradient at the read-out (tiny; torch on [B, n_bits])
        raw0 = rt.detector_forward(plan, dw, batch, mag)
        p = raw0.detach().clone().requires_grad_(True)
        loss = (((p - target) ** 2).mean(dim=-1) - 0.1 * p.abs().mean(dim=-1)).mean()
        loss.backward()
        vals, _, gw, gb = rt.detector_weight_gradients(plan, dw, batch, mag, p.grad)
        grads = parallel.all_reduce_gradients(gw + gb, average=True)
        for prm, g in zip(self.params, grads):
            prm.grad = g
        self.opt.step()
        ws = [prm.detach().cpu().numpy() for prm in self.params[: self.nl]]
        bs = [prm.detach().cpu().numpy() for prm in self.params[self.nl:]]
        self.net.weights, self.net.biases = ws, bs
        dw.update(ws, bs)
        sums, _ = parallel.reduce_metrics({"loss": float(loss.detach()), "n": 1.0}, {}, device=data.device)
        return sums["loss"] / sums["n"], vals
