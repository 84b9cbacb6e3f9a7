"""EXTENSION -- data-parallel detector training step (BASELINE.json north_star / configs 3-4: "training loop ... RCCL
all-reduce over xGMI on the embedder/detector gradients").

The reference trains nothing: its detector is frozen and seed-initialised (src/AWARE/embedding/multibit_embedder.py:76-77,
src/AWARE/detection/multibit_detector_net.py:77-80).  This module therefore has no reference counterpart -- parity unpinned;
its gradients are specified by torch autograd on oracle/aware_oracle.py::Detector and tested against it
(tests/test_gpu_seam.py::test_detector_weight_gradients_extension).

One step, on every rank's own clips (shard by clip, as everywhere else), everything on the device:
    mag = |STFT(normalise(audio))| on the band                   HIP: aware_stft_band
    raw, per-clip loss, dL/dW, dL/db in ONE forward + backward   HIP: aware_detector_train_gradients (f32-input MFMA pipe, the
                                                                 reference's own objective mse - 0.1 mean|raw|, losses.py:38-42)
    all-reduce (sum) of the flat gradient bucket + clip count    RCCL, one 6.7 MB collective (aware_amd/parallel.py)
    Adam step on the flat parameter bucket                       HIP: aware_opt_clamp_step (torch.optim.Adam's arithmetic),
                                                                 gradient scaled by 1 / clips of all ranks
    device images of the parameters rebuilt                      HIP: aware_detector_update_device (copies, transposes, bf16
                                                                 three-term and f16 two-term fragment images; no host copy)
The host copy of the weights (net.weights / net.biases, numpy) is refreshed only on request (sync_host()); other device copies
of the same network (another plan, another AWAREDetectorNet object) are NOT refreshed by a step.
"""
from __future__ import annotations

import numpy as np
import torch

from . import parallel
from . import runtime as rt


class DetectorTrainer:
    def __init__(self, detector, lr: float = 1e-4, sample_rate: int = 16000, betas=(0.9, 0.999), eps: float = 1e-8):
        """detector: an AWAREDetector (aware_amd.detection); the device copy of its network (for the plan of `sample_rate`) is
        trained in place."""
        from .embedding.optimizers import get_optimizer, step_table
        self.detector = detector
        self.net = detector.detection_net
        self.sample_rate = sample_rate
        shapes = [np.asarray(w).shape for w in self.net.weights] + [np.asarray(b).shape for b in self.net.biases]
        sizes = [int(np.prod(sh)) for sh in shapes]
        self.nl = len(self.net.weights)
        host = np.concatenate([np.asarray(w, dtype=np.float32).reshape(-1) for w in self.net.weights] +
                              [np.asarray(b, dtype=np.float32).reshape(-1) for b in self.net.biases])
        self.flat = torch.from_numpy(host).cuda()                      # parameters, one bucket
        self.grad = torch.zeros_like(self.flat)                        # gradients, one bucket (the all-reduce operand)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        offs = np.concatenate([[0], np.cumsum(sizes)]).tolist()
        self.params = [self.flat[offs[i]: offs[i + 1]].view(*shapes[i]) for i in range(len(shapes))]
        self.grads = [self.grad[offs[i]: offs[i + 1]].view(*shapes[i]) for i in range(len(shapes))]
        self.opt = get_optimizer("adam", lr=lr, betas=tuple(betas), eps=eps)
        self._table_fn = step_table
        self._table = step_table(self.opt, 1024)
        self.t = 0

    def sync_host(self):
        """Copy the trained parameters back into the network's host arrays (numpy)."""
        torch.cuda.synchronize()
        self.net.weights = [p.detach().cpu().numpy().copy() for p in self.params[: self.nl]]
        self.net.biases = [p.detach().cpu().numpy().copy() for p in self.params[self.nl:]]

    def step(self, audio: "rt.Ragged", bits: torch.Tensor):
        """audio: ragged device clips at 16 kHz (e.g. watermarked + attacked); bits [B, n_bits] 0/1.  Returns the loss
        averaged over the clips of all ranks (float) and the raw detector outputs."""
        from .embedding.optimizers import step_scalars
        import ctypes as C
        plan = self.detector._plan(self.sample_rate)
        dw = self.net.device_weights(plan)
        batch = rt.Batch(audio.lengths)
        data = audio.data if audio.data.dtype == torch.float32 else audio.data.float()
        mag, _ = rt.stft_band(plan, batch, data, normalize=True)
        target = (2 * bits - 1).to(torch.float32)
        vals, losses, _, _ = rt.detector_train_gradients(plan, dw, batch, mag, target, "push_extremes",
                                                         self.grads[: self.nl], self.grads[self.nl:])
        parallel.all_reduce_gradients([self.grad], average=False)
        sums, _ = parallel.reduce_metrics({"loss": float(losses.sum()), "n": float(batch.B)}, {}, device=data.device)
        self.t += 1
        if self.t > self._table.shape[0]:
            self._table = self._table_fn(self.opt, 2 * self._table.shape[0])
        c4, h8 = step_scalars(self.opt, self._table, self.t)
        h8[7] = 1.0 / sums["n"]                                        # summed gradients -> gradient of the mean over all clips
        check = rt.check
        check(dw.lib.aware_opt_clamp_step(self.opt["kind"], rt._ptr(self.flat), rt._ptr(self.grad), rt._ptr(self.m), rt._ptr(self.v),
                                          None, None, self.flat.numel(), c4.ctypes.data_as(C.POINTER(C.c_float)),
                                          h8.ctypes.data_as(C.POINTER(C.c_float)), rt._stream()), "aware_opt_clamp_step")
        dw.update_device(self.params[: self.nl], self.params[self.nl:])
        return sums["loss"] / sums["n"], vals
