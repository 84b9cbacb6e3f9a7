"""AWAREDetectorNet: frozen, seed-initialised detector (mel -> InstanceNorm -> global standardise
-> pool -> 4 x [conv1x1, InstanceNorm, LeakyReLU] -> bitwise read-out head).

Reference: src/AWARE/detection/multibit_detector_net.py:17-140.  The weights are never trained
anywhere in the reference (multibit_embedder.py:76-77 freezes them; there is no checkpoint), so
this class owns host copies generated from the reference's seed and a device copy inside
libaware_hip.so; forward() runs on the GPU through the C ABI."""
from __future__ import annotations

import numpy as np
import torch

from ..interfaces import BaseDetectorNet
from .mel import mel_filter_bank

DETECTOR_SEED = 328656719      # multibit_detector_net.py:78


class AWAREDetectorNet(BaseDetectorNet):
    def __init__(self, sample_rate: int = 16000, n_fft: int = 1024, n_mels: int = 128,
                 initial_pool_size: int = 2, initial_pool_stride: int = 2, num_blocks: int = 3,
                 n_filters=(512, 1024, 1024), kernel_size: int = 1, stride: int = 1, padding: int = 0,
                 norm_layer: str = "instance", activation: str = "leaky_relu", output_length: int = 20,
                 final_activation: str = "tanh"):
        n_filters = list(n_filters)
        assert len(n_filters) == num_blocks, "Number of filters must match number of blocks"
        unsupported = []
        if (kernel_size, stride, padding) != (1, 1, 0):
            unsupported.append("kernel_size/stride/padding other than 1/1/0")
        if (initial_pool_size, initial_pool_stride) != (2, 2):
            unsupported.append("initial pool other than (2, 2)")
        if norm_layer.lower() != "instance" or activation.lower() != "leaky_relu" or final_activation.lower() != "tanh":
            unsupported.append("norm/activation other than instance/leaky_relu/tanh")
        if unsupported:
            raise NotImplementedError("the HIP detector implements the reference's model card only: " + "; ".join(unsupported))
        self.sample_rate, self.n_fft, self.n_mels = sample_rate, n_fft, n_mels
        self.num_blocks, self.initial_pool_size, self.output_length = num_blocks, initial_pool_size, output_length
        self.final_activation = final_activation
        self.channels = [n_mels] + n_filters + [2 * output_length]
        self.mel_basis = mel_filter_bank(sample_rate, n_fft, n_mels)
        # torch.manual_seed(seed); self.apply(_init_weights): xavier-uniform on each Conv1d weight
        # in registration order, zero bias (:77-80, :98-107).  A private generator with the same
        # seed draws the same mt19937 stream without reseeding the caller's global RNG.
        gen = torch.Generator().manual_seed(DETECTOR_SEED)
        self.weights, self.biases = [], []
        for cin, cout in zip(self.channels[:-1], self.channels[1:]):
            w = torch.empty(cout, cin, 1)
            torch.nn.init.xavier_uniform_(w, generator=gen)
            self.weights.append(w[:, :, 0].contiguous().numpy())
            self.biases.append(np.zeros(cout, dtype=np.float32))
        self._dev = None

    def eval(self):
        return self

    def to(self, device):
        return self

    def parameters(self):
        for w, b in zip(self.weights, self.biases):
            yield w
            yield b

    def device_weights(self, plan):
        """Device copy (aware_detector) bound to a plan; created on first use."""
        from ..runtime import DetectorWeights
        if self._dev is None or self._dev.plan is not plan:
            self._dev = DetectorWeights(plan, self.mel_basis, self.weights, self.biases)
        return self._dev

    def forward(self, stft_magnitude: torch.Tensor) -> torch.Tensor:
        """[B, n_fft/2+1, T] magnitudes -> [B, output_length, 1]  (net :109-140).

        Only the embedding band reaches the network (callers zero the rest,
        multibit_embedder.py:104, multibit_detector.py:34-37); the band is taken from the
        default plan (500-4000 Hz at 16 kHz)."""
        return _DetectorNetFn.apply(stft_magnitude.to("cuda", torch.float32), self)

    def get_model_info(self):
        total = int(sum(int(np.prod(p.shape)) for p in self.parameters()))
        return {"sample_rate": self.sample_rate, "n_fft": self.n_fft, "n_mels": self.n_mels,
                "num_blocks": self.num_blocks, "output_length": self.output_length,
                "final_activation": self.final_activation, "total_parameters": total, "trainable_parameters": 0}


def _frames_batch(B, T):
    """Batch geometry with exactly T frames per clip (n = 256*(T-1) samples); cached."""
    from ..utils.audio import get_batch
    return get_batch([max(256 * (T - 1), 513)] * B)


def _band_rows(stft_magnitude, plan):
    """[B, F, T] -> frame-major band rows [B*T, 256] (layout conversion at the seam)."""
    from .. import runtime as rt
    B, F, T = stft_magnitude.shape
    lo, hi = plan.band_bins
    mag = torch.zeros((B * T, rt.SPEC_STRIDE), dtype=torch.float32, device=stft_magnitude.device)
    mag[:, : hi - lo + 1] = stft_magnitude[:, lo:hi + 1, :].permute(0, 2, 1).reshape(B * T, -1)
    return mag


class _DetectorNetFn(torch.autograd.Function):
    """AWAREDetectorNet.forward under autograd (the reference back-propagates the loss through the frozen network to the
    magnitudes, multibit_embedder.py:107-111): forward = aware_detector_forward, backward = aware_detector_backward."""

    @staticmethod
    def forward(ctx, stft_magnitude, net):
        from .. import runtime as rt
        from ..utils.audio import default_plan
        plan = default_plan()
        B, F, T = stft_magnitude.shape
        batch = _frames_batch(B, T)
        mag = _band_rows(stft_magnitude, plan)
        ctx.save_for_backward(mag)
        ctx.geom = (plan, batch, net, (B, F, T))
        return rt.detector_forward(plan, net.device_weights(plan), batch, mag).unsqueeze(-1)

    @staticmethod
    def backward(ctx, g):
        from .. import runtime as rt
        (mag,) = ctx.saved_tensors
        plan, batch, net, (B, F, T) = ctx.geom
        _, gmag = rt.detector_backward(plan, net.device_weights(plan), batch, mag, g.reshape(B, -1))
        lo, hi = plan.band_bins
        out = torch.zeros((B, F, T), dtype=torch.float32, device=g.device)
        out[:, lo:hi + 1, :] = gmag[:, : hi - lo + 1].reshape(B, T, -1).permute(0, 2, 1)
        return out, None
