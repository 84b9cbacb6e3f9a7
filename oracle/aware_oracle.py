"""CPU restatement of deepmarkpy/aware's embed -> attack -> detect path.

*** TEST INFRASTRUCTURE -- NOT THE PRODUCT PATH ***
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker.  aware_amd/ never imports it.

Parity status: PINNED.  Every function below is checked in
tests/test_oracle_golden.py against tests/golden/*.npz, which were produced by
running the reference itself in the development container with
tools/make_golden.py (recipe: SURVEY.md 8(c); torch 2.10.0 / numpy 2.2.6 /
scipy 1.15.3; the reference pins torch 2.7.0 / numpy 1.26.4, scipy unpinned).
Stand-ins used by that run: librosa.fft_frequencies (same formula as the
reference's own detection/modules/mel.py:72-74), webrtcvad (gate bypassed ->
the VAD gate's parity is UNPINNED), resampy (imported, never called).
Extensions that do not exist in the reference (gaussian_noise, mp3_surrogate)
are marked "EXTENSION -- parity unpinned": this file is their specification.

All file:line citations are relative to /root/reference.
Written with explicit primitives (reflect-pad / unfold / rfft / overlap-add)
rather than torch.stft / torch.istft so that it states the algorithm the HIP
kernels implement; tests also check it against torch.stft / torch.istft.
"""
from __future__ import annotations

import math
import random as _pyrandom

import numpy as np
import torch

# cards/config.yaml:1-46 (defaults mirrored by utils/models/load_model.py:22-36)
N_FFT = 1024
HOP = 256
SAMPLE_RATE = 16000
BANDS = (500, 4000)
TOLERANCE_DB = 6.0
NUM_ITERATIONS = 400
N_MELS = 128
N_FILTERS = (512, 1024, 1024)
OUTPUT_LENGTH = 20
DETECTOR_SEED = 328656719            # detection/multibit_detector_net.py:78
LR = 0.1


# --------------------------------------------------------------------------
# DSP plugins  (utils/audio/stft.py, utils/audio/waveform.py)
# --------------------------------------------------------------------------
def hann_window(n=N_FFT, dtype=torch.float32):
    """torch.hann_window(n) (periodic) -- utils/audio/stft.py:19-22."""
    return torch.hann_window(n, dtype=dtype)


def waveform_normalize(x):
    """WaveformNormalizer -- utils/audio/waveform.py:18-19 (whole-tensor max)."""
    return x / torch.max(torch.abs(x) + 1e-8)


def stft(x, n_fft=N_FFT, hop=HOP, window=None):
    """STFT.__call__ -- utils/audio/stft.py:27-28.

    torch.stft(center=True [reflect], onesided, unnormalised) restated as
    reflect-pad(n_fft/2) -> frames(n_fft, hop) -> * window -> rfft.
    x: [n] or [B, n] real.  Returns [513, T] or [B, 513, T] complex.
    """
    if window is None:
        window = hann_window(n_fft, x.dtype)
    squeeze = x.dim() == 1
    xb = x.unsqueeze(0) if squeeze else x
    xp = torch.nn.functional.pad(xb.unsqueeze(1), (n_fft // 2, n_fft // 2), mode="reflect").squeeze(1)
    frames = xp.unfold(-1, n_fft, hop)                      # [B, T, n_fft]
    S = torch.fft.rfft(frames * window, dim=-1).transpose(1, 2)   # [B, 513, T]
    return S[0] if squeeze else S


def ola_envelope(T, n_fft=N_FFT, hop=HOP, window=None, dtype=torch.float32):
    """Sum of squared windows over the padded length n_fft + hop*(T-1)."""
    if window is None:
        window = hann_window(n_fft, dtype)
    L = n_fft + hop * (T - 1)
    env = torch.zeros(L, dtype=window.dtype)
    w2 = window * window
    for t in range(T):
        env[t * hop: t * hop + n_fft] += w2
    return env


def istft(X, n_fft=N_FFT, hop=HOP, window=None):
    """ISTFT.__call__ -- utils/audio/stft.py:47-48 (no `length=` argument).

    torch.istft(center=True) restated as irfft(n_fft) -> * window -> overlap-add
    -> / overlap-add(window^2) -> [n_fft/2 : n_fft/2 + hop*(T-1)].
    X: [513, T] or [B, 513, T] complex.  Returns [hop*(T-1)] or [B, hop*(T-1)].
    """
    squeeze = X.dim() == 2
    Xb = X.unsqueeze(0) if squeeze else X
    B, F, T = Xb.shape
    rdtype = Xb.real.dtype
    if window is None:
        window = hann_window(n_fft, rdtype)
    frames = torch.fft.irfft(Xb.transpose(1, 2), n=n_fft, dim=-1) * window   # [B, T, n_fft]
    L = n_fft + hop * (T - 1)
    # overlap-add via fold (differentiable)
    y = torch.nn.functional.fold(frames.transpose(1, 2), output_size=(1, L),
                                 kernel_size=(1, n_fft), stride=(1, hop)).reshape(B, L)
    env = ola_envelope(T, n_fft, hop, window, rdtype)
    start = n_fft // 2
    end = start + hop * (T - 1)
    y = y[:, start:end] / env[start:end]
    return y[0] if squeeze else y


def band_indices(sr=SAMPLE_RATE, n_fft=N_FFT, bands=BANDS):
    """_get_embedding_frequency_indices -- embedding/multibit_embedder.py:43-47.

    librosa.fft_frequencies == np.linspace(0, sr/2, 1+n_fft//2)
    (the reference's own copy: detection/modules/mel.py:72-74)."""
    freqs = np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)
    mask = (freqs >= bands[0]) & (freqs <= bands[1])
    return np.where(mask)[0], np.where(~mask)[0]


# --------------------------------------------------------------------------
# Detector  (detection/*)
# --------------------------------------------------------------------------
def _hz_to_mel(f):
    """Slaney mel scale -- detection/modules/mel.py:6-35 (htk=False)."""
    f = np.atleast_1d(np.asarray(f, dtype=np.float64))
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    t = f >= min_log_hz
    mels[t] = min_log_mel + np.log(f[t] / min_log_hz) / logstep
    return mels


def _mel_to_hz(m):
    """detection/modules/mel.py:38-69."""
    m = np.atleast_1d(np.asarray(m, dtype=np.float64))
    f_sp = 200.0 / 3
    hz = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    t = m >= min_log_mel
    hz[t] = min_log_hz * np.exp(logstep * (m[t] - min_log_mel))
    return hz


def mel_filter_bank(sr=SAMPLE_RATE, n_fft=N_FFT, n_mels=N_MELS, fmin=0.0, fmax=None):
    """get_mel_filter_bank -- detection/modules/mel.py:105-149 (slaney norm, f32)."""
    if fmax is None:
        fmax = float(sr) / 2
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    fftfreqs = np.linspace(0, sr / 2, 1 + n_fft // 2, endpoint=True)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin)[0], _hz_to_mel(fmax)[0], n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def detector_weights(seed=DETECTOR_SEED, n_mels=N_MELS, n_filters=N_FILTERS,
                     output_length=OUTPUT_LENGTH):
    """Seeded xavier-uniform conv weights, zero bias.

    detection/multibit_detector_net.py:77-80 (torch.manual_seed then
    self.apply(_init_weights)), :98-107 (xavier_uniform_ on Conv1d weight, bias 0).
    Module.apply visits conv_blocks[0..3].conv in registration order; only the
    four Conv1d weights draw from the generator.  A local generator with the same
    seed replaces the reference's global reseed (same mt19937 stream)."""
    g = torch.Generator().manual_seed(seed)
    ch = [n_mels] + list(n_filters) + [2 * output_length]
    ws, bs = [], []
    for i in range(len(ch) - 1):
        w = torch.empty(ch[i + 1], ch[i], 1)
        torch.nn.init.xavier_uniform_(w, generator=g)
        ws.append(w[:, :, 0].contiguous())
        bs.append(torch.zeros(ch[i + 1]))
    return ws, bs


class Detector:
    """AWAREDetectorNet.forward -- detection/multibit_detector_net.py:109-140.

    Batched restatement: GlobalStandardize (modules/globalStandardize.py:16-21)
    is whole-tensor in the reference where B is always 1, so it is per sample
    here.  global_norm1's result is discarded by the reference (:121 vs :124)."""

    def __init__(self, dtype=torch.float32):
        self.mel = torch.from_numpy(mel_filter_bank()).to(dtype)        # [128, 513]
        ws, bs = detector_weights()
        self.ws = [w.to(dtype) for w in ws]
        self.bs = [b.to(dtype) for b in bs]

    @staticmethod
    def instance_norm(x, eps=1e-5):
        """nn.InstanceNorm1d (no affine, biased var) over the last dim."""
        mu = x.mean(dim=-1, keepdim=True)
        var = x.var(dim=-1, unbiased=False, keepdim=True)
        return (x - mu) / torch.sqrt(var + eps)

    def forward(self, mag):
        """mag: [B, 513, T] -> [B, 20]."""
        x = torch.matmul(self.mel, mag)                                  # mel.py:185-201
        x = self.instance_norm(x)                                        # net :50,126
        mean = x.mean(dim=(1, 2), keepdim=True)                          # globalStandardize.py:17
        std = x.std(dim=(1, 2), keepdim=True)                            # unbiased, :18
        x = (x - mean) / (std + 1e-8)
        x = torch.nn.functional.avg_pool1d(x, 2, 2)                      # net :53,131
        for w, b in zip(self.ws, self.bs):                               # modules/conv1d.py:38-42
            x = torch.matmul(w, x) + b[:, None]
            x = self.instance_norm(x)
            x = torch.nn.functional.leaky_relu(x, 0.2)
        x = x.mean(dim=-1)                                               # modules/BRH.py:18
        x = x[:, 0::2] - x[:, 1::2]                                      # :21-23
        return torch.tanh(x)                                             # :25


def push_extremes_loss(pred, target, penalty_weight=0.1):
    """PushToExtremesLoss -- embedding/losses.py:38-42.  Per-sample over the last dim."""
    mse = ((pred - target) ** 2).mean(dim=-1)
    return mse - penalty_weight * pred.abs().mean(dim=-1)


LOSSES = {
    # embedding/losses.py:95-103
    "hinge": lambda p, t: torch.clamp(1 - p * t, min=0).mean(dim=-1),
    "mse": lambda p, t: ((p - t) ** 2).mean(dim=-1),
    "push_extremes": push_extremes_loss,
    "sign": lambda p, t: torch.clamp(-p * t, min=0).mean(dim=-1),
    "push_sigmoid": lambda p, t: ((p - t) ** 2).mean(dim=-1) - 0.1 * (p - 0.5).abs().mean(dim=-1),
    "ber": lambda p, t: (torch.sign(p) != torch.sign(t)).float().mean(dim=-1) + 0.0 * p.sum(dim=-1),
    # EXTENSION (not in the reference; BASELINE.json config 3 "BER + L1 loss"): the read-out part is push_extremes,
    # Embedder.forward_loss adds l1_weight * mean|c - c0| over the clip's coefficients.  Parity unpinned: this
    # restatement IS the specification of loss id 6 (AWARE_LOSS_PUSH_L1).
    "push_extremes_l1": push_extremes_loss,
}


def nadam_schedule(num_steps, lr=LR, beta1=0.9, beta2=0.999, momentum_decay=4e-3):
    """Per-step scalars of torch.optim.NAdam's single-tensor CPU path.

    torch/optim/nadam.py `_single_tensor_nadam` (reference call site:
    embedding/optimizers.py:3-20, cards/config.yaml:17-20).  mu_product is kept in
    a float32 0-dim tensor by torch and read back with .item(): restated here with
    np.float32.  Returns arrays c_grad, c_mom, bias_correction2 (float64)."""
    mu_product = np.float32(1.0)
    cg, cm, bc2 = [], [], []
    for step in range(1, num_steps + 1):
        bias_correction2 = 1 - beta2 ** step
        mu = beta1 * (1.0 - 0.5 * (0.96 ** (step * momentum_decay)))
        mu_next = beta1 * (1.0 - 0.5 * (0.96 ** ((step + 1) * momentum_decay)))
        mu_product = np.float32(mu_product * np.float32(mu))
        mp = float(mu_product)
        cg.append(-lr * (1.0 - mu) / (1.0 - mp))
        cm.append((-lr * mu_next) / (1.0 - mp * mu_next))
        bc2.append(bias_correction2)
    return np.asarray(cg), np.asarray(cm), np.asarray(bc2)


def nadam_step(p, g, m, v, cg, cm, bc2, beta1=0.9, beta2=0.999, eps=1e-8):
    """One NAdam update in place on tensors p, m, v (see nadam_schedule)."""
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    denom = v.div(bc2).sqrt().add_(eps)
    p.addcdiv_(g, denom, value=cg)
    p.addcdiv_(m, denom, value=cm)


# --------------------------------------------------------------------------
# Embedder  (embedding/multibit_embedder.py)
# --------------------------------------------------------------------------
class Embedder:
    """AWAREEmbedder.embed / _optimize -- embedding/multibit_embedder.py:70-197."""

    def __init__(self, num_iterations=NUM_ITERATIONS, tolerance_db=TOLERANCE_DB,
                 loss="push_extremes", dtype=torch.float32, l1_weight=0.0,
                 optimizer=None, optimizer_params=None, scheduler=None, scheduler_params=None):
        """optimizer / scheduler: names of the reference's registries (embedding/optimizers.py:3-20, schedulers.py:3-16) run
        through torch.optim exactly as the reference's loop does (:85-86, :112-113); None = the model card's NAdam(lr 0.1)
        with a constant rate, restated by hand below (nadam_step)."""
        self.optimizer, self.optimizer_params = optimizer, dict(optimizer_params or {})
        self.scheduler, self.scheduler_params = scheduler, dict(scheduler_params or {})
        self.det = Detector(dtype)
        self.l1_weight = l1_weight if loss == "push_extremes_l1" else 0.0
        self.num_iterations = num_iterations
        self.tolerance_db = tolerance_db
        self.loss = LOSSES[loss]
        self.dtype = dtype
        self.band, self.nonband = band_indices()

    def analyse(self, audio):
        """:143-147 -- [WaveformNormalizer, STFT, STFTDecomposer] on [B, n]."""
        x = audio / torch.amax(torch.abs(audio) + 1e-8, dim=-1, keepdim=True)
        S = stft(x)
        return torch.abs(S), torch.angle(S)

    def bounds(self, coeffs0):
        """:157-160 (+ :89-90): lo = max(0, c - d), hi = c + d, d = c*10^(-tol/20)."""
        delta = coeffs0 * 10 ** (-self.tolerance_db / 20)
        return torch.clamp(coeffs0 - delta, min=0), coeffs0 + delta

    def recompute_magnitude(self, mag_full, phase):
        """_recompute_watermarked_magnitude -- :49-67.

        Assembler -> ISTFT -> normalise -> normalise -> STFT -> |.|  (the
        post-process list ends with a normaliser and the pre-process list starts
        with one, so y is normalised twice)."""
        X = mag_full * torch.exp(1j * phase)                              # stft.py:62
        y = istft(X)
        y = y / torch.amax(torch.abs(y) + 1e-8, dim=-1, keepdim=True)
        y = y / torch.amax(torch.abs(y) + 1e-8, dim=-1, keepdim=True)
        return torch.abs(stft(y)), y

    def forward_loss(self, coeffs, mag0, phase, target):
        """One pass of the loop body :99-109.  coeffs: [B, Fb, T]."""
        mag = mag0.clone()
        mag[:, self.band] = coeffs
        mag2, _ = self.recompute_magnitude(mag, phase)
        mag2 = mag2.clone()
        mag2[:, self.nonband] = 0.0
        pred = self.det.forward(mag2)
        loss = self.loss(pred, target)
        if self.l1_weight:
            # EXTENSION: imperceptibility as an L1 penalty on the coefficient change, per clip
            loss = loss + self.l1_weight * (coeffs - mag0[:, self.band]).abs().mean(dim=(-2, -1))
        return loss, pred

    def embed_registry(self, audio, watermark, record=None):
        """The loop with an optimiser / scheduler from the reference's registries, ONE clip (audio [1, n]) like the reference:
        optimizer = registry[name]([coeffs], **params); scheduler = registry[name](optimizer, **params); per iteration
        loss.backward(); optimizer.step(); scheduler.step(...); clamp; best tracking (:85-122).  ReduceLROnPlateau is stepped
        with the loss as the reference does (:113).  The reference passes the loss to EVERY scheduler -- for the step-count
        schedulers that lands in torch's deprecated `epoch` argument (last_epoch := loss, a usage error torch warns about);
        here they are stepped without an argument, their documented per-iteration use.  record(it, loss, lr)."""
        opt_reg = {"adam": torch.optim.Adam, "nadam": torch.optim.NAdam, "sgd": torch.optim.SGD, "rmsprop": torch.optim.RMSprop,
                   "adagrad": torch.optim.Adagrad, "adadelta": torch.optim.Adadelta, "adamax": torch.optim.Adamax,
                   "adamw": torch.optim.AdamW}
        sch = torch.optim.lr_scheduler
        sched_reg = {"reduce_lr_on_plateau": sch.ReduceLROnPlateau, "cosine_annealing": sch.CosineAnnealingLR,
                     "cosine_annealing_warm_restarts": sch.CosineAnnealingWarmRestarts, "step": sch.StepLR,
                     "multi_step": sch.MultiStepLR, "exponential": sch.ExponentialLR, "cyclic": sch.CyclicLR}
        audio = torch.as_tensor(audio, dtype=self.dtype)
        target = torch.as_tensor(watermark, dtype=self.dtype)
        assert audio.shape[0] == 1
        with torch.no_grad():
            mag0, phase = self.analyse(audio)
            c0 = mag0[:, self.band].clone()
            lo, hi = self.bounds(c0)
        c = c0.clone().requires_grad_(True)
        optimizer = opt_reg[self.optimizer or "nadam"]([c], **(self.optimizer_params or {"lr": LR}))
        scheduler = sched_reg[self.scheduler](optimizer, **self.scheduler_params) if self.scheduler else None
        best_loss, best_c = float("inf"), c0.clone()
        for it in range(self.num_iterations):
            optimizer.zero_grad()
            loss, pred = self.forward_loss(c, mag0, phase, target)
            lr_used = float(optimizer.param_groups[0]["lr"])
            loss.sum().backward()
            optimizer.step()
            if scheduler is not None:
                if self.scheduler == "reduce_lr_on_plateau":
                    scheduler.step(loss.detach()[0])
                else:
                    scheduler.step()
            with torch.no_grad():
                c.data = torch.clamp(c.data, lo, hi)
                if float(loss) < best_loss:
                    best_loss, best_c = float(loss), c.detach().clone()
            if record is not None:
                record(it, float(loss), lr_used)
        with torch.no_grad():
            mag = mag0.clone()
            mag[:, self.band] = best_c
            y = istft(mag * torch.exp(1j * phase))
            y = y / torch.amax(torch.abs(y) + 1e-8, dim=-1, keepdim=True)
        return y, best_loss

    def embed(self, audio, watermark, record=None):
        """embed (:141-197).  audio [B, n] f32, watermark [B, 20] bipolar.

        Returns watermarked audio [B, 256*(T-1)] (normalised, before the
        service-level rescale)."""
        audio = torch.as_tensor(audio, dtype=self.dtype)
        target = torch.as_tensor(watermark, dtype=self.dtype)
        with torch.no_grad():
            mag0, phase = self.analyse(audio)
            c0 = mag0[:, self.band].clone()
            lo, hi = self.bounds(c0)
        c = c0.clone().requires_grad_(True)
        m = torch.zeros_like(c0)
        v = torch.zeros_like(c0)
        cg, cm, bc2 = nadam_schedule(self.num_iterations)
        B = audio.shape[0]
        best_loss = torch.full((B,), float("inf"), dtype=self.dtype)
        best_c = c0.clone()
        for it in range(self.num_iterations):
            if c.grad is not None:
                c.grad = None
            loss, pred = self.forward_loss(c, mag0, phase, target)
            loss.sum().backward()                       # clips are independent
            with torch.no_grad():
                nadam_step(c.data, c.grad, m, v, cg[it], cm[it], bc2[it])
                c.data = torch.clamp(c.data, lo, hi)                      # :116-117
                # :120-122 -- loss of the PRE-step coeffs, snapshot POST-step+clamp
                better = loss.detach() < best_loss
                best_loss = torch.where(better, loss.detach(), best_loss)
                best_c[better] = c.data[better]
            if record is not None:
                record(it, loss.detach().clone(), pred.detach().clone(),
                       c.grad.detach().clone() if it == 0 else None)
        with torch.no_grad():
            mag = mag0.clone()
            mag[:, self.band] = best_c                                    # :173-174
            y = istft(mag * torch.exp(1j * phase))                        # :185-192
            y = y / torch.amax(torch.abs(y) + 1e-8, dim=-1, keepdim=True)
        return y, best_loss

    def detect_raw(self, audio):
        """AWAREDetector.detect -- detection/multibit_detector.py:28-42."""
        audio = torch.as_tensor(audio, dtype=self.dtype)
        with torch.no_grad():
            x = audio / torch.amax(torch.abs(audio) + 1e-8, dim=-1, keepdim=True)
            mag = torch.abs(stft(x)).clone()
            mag[:, self.nonband] = 0.0
            return self.det.forward(mag)


# --------------------------------------------------------------------------
# Codec / service / metric
# --------------------------------------------------------------------------
def bits_to_bipolar(bits):
    """PatternEncoder._bits_to_bipolar -- utils/watermark/encoder.py:35-45."""
    return (2 * np.asarray(bits, dtype=np.int32) - 1).astype(np.int32)


def decode_bits(values, threshold=0.0):
    """PatternDecoder bits2bipolar path -- utils/watermark/decoder.py:17,51,63."""
    bip = 2 * (np.asarray(values) > threshold).astype(np.int32) - 1
    return (bip > 0).astype(np.int32)


def ber_percent(out, tgt):
    """BER.__call__ -- metrics/audio.py:8-17 (percent)."""
    return float(np.mean(np.asarray(out) != np.asarray(tgt)) * 100)


def embed_watermark(audio, bits, embedder: Embedder):
    """service/embed.py:7-80, mono path, VAD gate bypassed (unpinned)."""
    audio = np.asarray(audio)
    wm = bits_to_bipolar(bits)
    audio_mx = np.max(audio)                                              # :69 signed max
    y, _ = embedder.embed(audio[None].astype(np.float32), wm[None])
    return audio_mx * y[0].numpy()                                        # :73


def detect_watermark(audio, embedder: Embedder):
    """service/detect.py:7-55, mono path."""
    raw = embedder.detect_raw(np.asarray(audio, dtype=np.float32)[None])[0].numpy()
    return decode_bits(raw), raw


# --------------------------------------------------------------------------
# Attacks  (scripts/attacks.py)
# --------------------------------------------------------------------------
def pcm_bit_depth(audio, bits):
    """PCMBitDepthConversion.apply -- scripts/attacks.py:44-70 (truncating cast).

    The reference's "12-bit" branch uses the 13-bit range -4096..4095 / 4095
    (:56-59); restated as written."""
    table = {8: (127.0, -128, 127), 12: (4095.0, -4096, 4095),
             16: (32767.0, -32768, 32767), 24: (8388607.0, -8388608, 8388607)}
    if bits not in table:
        raise ValueError(f"Unsupported PCM bit depth: {bits}")
    q, lo, hi = table[bits]
    audio = np.asarray(audio)
    audio = audio / np.max(np.abs(audio) + 1e-8)
    ai = np.trunc(np.clip(audio * q, lo, hi))
    return ai.astype(np.float32) / np.float32(q)


def resample_poly_design(up, down):
    """scipy.signal.resample_poly's default filter: firwin(2*half_len+1, f_c,
    window=('kaiser', 5.0)) with half_len = 10*max(up,down), f_c = 1/max(up,down),
    scaled by `up`.  (Third-party: scipy, unpinned by the reference; call site
    scripts/attacks.py:290-293.)"""
    g = math.gcd(up, down)
    up //= g
    down //= g
    max_rate = max(up, down)
    f_c = 1.0 / max_rate
    half_len = 10 * max_rate
    n = 2 * half_len + 1
    # firwin: windowed-sinc low-pass, cutoff f_c (Nyquist = 1), scaled to unit DC gain
    m = np.arange(n) - (n - 1) / 2.0
    h = f_c * np.sinc(f_c * m)
    beta = 5.0
    w = np.i0(beta * np.sqrt(np.clip(1 - (2 * m / (n - 1)) ** 2, 0, 1))) / np.i0(beta)
    h = h * w
    h = h / np.sum(h)
    return h * up, up, down, half_len


def resample_poly(x, up, down):
    """scipy.signal.resample_poly restated (upfirdn with the padding scipy uses).

    Output length ceil(n*up/down); the filter is centred (zero phase)."""
    x = np.asarray(x)
    h, up, down, half_len = resample_poly_design(up, down)
    if x.dtype == np.float32:
        # scipy casts the (unscaled) filter to x.dtype before `h *= up`
        h = (h / up).astype(np.float32) * np.float32(up)
    n_in = x.shape[0]
    n_out = -(-n_in * up // down)
    # y[j] = sum_i x[i] * h[j*down - i*up + half_len]
    nh = h.shape[0]
    K = (nh - 1) // up + 1
    pos = np.arange(n_out, dtype=np.int64) * down + half_len
    i_hi = pos // up
    k = np.arange(K, dtype=np.int64)
    idx = i_hi[:, None] - k[None, :]
    tap = (pos % up)[:, None] + k[None, :] * up
    ok = (idx >= 0) & (idx < n_in) & (tap < nh)
    xv = x.astype(np.float64)[np.clip(idx, 0, n_in - 1)]
    hv = h.astype(np.float64)[np.clip(tap, 0, nh - 1)]
    out = np.sum(np.where(ok, xv * hv, 0.0), axis=1)
    return out.astype(x.dtype if x.dtype in (np.float32, np.float64) else np.float64)


def resample_attack(audio, sr=SAMPLE_RATE, target_sr=16000):
    """Resample.apply -- scripts/attacks.py:267-294."""
    k = sr // target_sr
    if k > 1:
        down = audio[::k]
        return np.interp(np.arange(len(audio)), np.arange(0, len(audio), k), down)
    a = resample_poly(audio, 441, 160)
    return resample_poly(a, 160, 441)


def butter(order, wn, btype):
    """scipy.signal.butter(order, Wn, btype, analog=False) -> (b, a), restated:
    analog prototype poles -> frequency transform -> bilinear (fs = 2)."""
    wn = np.atleast_1d(np.asarray(wn, dtype=np.float64))
    # prototype
    mm = np.arange(-order + 1, order, 2)
    p = -np.exp(1j * np.pi * mm / (2 * order))
    z = np.array([], dtype=complex)
    k = 1.0
    fs = 2.0
    warped = 2 * fs * np.tan(np.pi * wn / fs)
    if btype in ("low", "lowpass"):
        wo = warped[0]
        z2 = z * wo
        p2 = p * wo
        k2 = k * wo ** (len(p) - len(z))
    elif btype in ("high", "highpass"):
        wo = warped[0]
        z2 = wo / z if len(z) else np.array([], dtype=complex)
        p2 = wo / p
        z2 = np.append(z2, np.zeros(len(p) - len(z)))
        k2 = k * np.real(np.prod(-z) / np.prod(-p))
    elif btype == "bandstop":
        bw = warped[1] - warped[0]
        wo = np.sqrt(warped[0] * warped[1])
        z_hp = (bw / 2) / z if len(z) else np.array([], dtype=complex)
        p_hp = (bw / 2) / p
        z_hp = z_hp.astype(complex)
        p_hp = p_hp.astype(complex)
        z2 = np.concatenate((z_hp + np.sqrt(z_hp ** 2 - wo ** 2), z_hp - np.sqrt(z_hp ** 2 - wo ** 2)))
        p2 = np.concatenate((p_hp + np.sqrt(p_hp ** 2 - wo ** 2), p_hp - np.sqrt(p_hp ** 2 - wo ** 2)))
        degree = len(p) - len(z)
        z2 = np.append(z2, np.full(degree, +1j * wo))
        z2 = np.append(z2, np.full(degree, -1j * wo))
        k2 = k * np.real(np.prod(-z) / np.prod(-p))
    else:
        raise ValueError(btype)
    # bilinear
    fs2 = 2.0 * fs
    degree = len(p2) - len(z2)
    zd = (fs2 + z2) / (fs2 - z2)
    pd = (fs2 + p2) / (fs2 - p2)
    zd = np.append(zd, -np.ones(degree))
    kd = k2 * np.real(np.prod(fs2 - z2) / np.prod(fs2 - p2))
    b = kd * np.real(np.poly(zd))
    a = np.real(np.poly(pd))
    return b, a


def lfilter(b, a, x, zi=None):
    """scipy.signal.lfilter: direct-form II transposed, float64."""
    b = np.asarray(b, dtype=np.float64) / a[0]
    a = np.asarray(a, dtype=np.float64) / a[0]
    n = max(len(a), len(b))
    b = np.concatenate([b, np.zeros(n - len(b))])
    a = np.concatenate([a, np.zeros(n - len(a))])
    z = np.zeros(n - 1) if zi is None else np.array(zi, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    y = np.empty_like(x)
    for i in range(x.shape[0]):
        xi = x[i]
        yi = b[0] * xi + z[0]
        for k in range(n - 2):
            z[k] = z[k + 1] + b[k + 1] * xi - a[k + 1] * yi
        z[n - 2] = b[n - 1] * xi - a[n - 1] * yi
        y[i] = yi
    return y, z


def lfilter_zi(b, a):
    """scipy.signal.lfilter_zi: steady-state state of a unit step input."""
    b = np.asarray(b, dtype=np.float64) / a[0]
    a = np.asarray(a, dtype=np.float64) / a[0]
    n = max(len(a), len(b))
    b = np.concatenate([b, np.zeros(n - len(b))])
    a = np.concatenate([a, np.zeros(n - len(a))])
    # companion(a).T
    comp = np.zeros((n - 1, n - 1))
    comp[0, :] = -a[1:]
    comp[1:, :-1] = np.eye(n - 2)
    IminusA = np.eye(n - 1) - comp.T
    B = b[1:] - a[1:] * b[0]
    zi = np.zeros(n - 1)
    zi[0] = B.sum() / IminusA[:, 0].sum()
    asum = 1.0
    csum = 0.0
    for k in range(1, n - 1):
        asum += a[k]
        csum += b[k] - a[k] * b[0]
        zi[k] = asum * zi[0] - csum
    return zi


def filtfilt(b, a, x):
    """scipy.signal.filtfilt (padtype='odd', padlen=3*max(len(a),len(b)))."""
    x = np.asarray(x, dtype=np.float64)
    ntaps = max(len(a), len(b))
    edge = 3 * ntaps
    left = 2 * x[0] - x[edge:0:-1]
    right = 2 * x[-1] - x[-2:-(edge + 2):-1]
    ext = np.concatenate([left, x, right])
    zi = lfilter_zi(b, a)
    y, _ = lfilter(b, a, ext, zi * ext[0])
    y, _ = lfilter(b, a, y[::-1], zi * y[-1])
    y = y[::-1]
    return y[edge:-edge]


def lowpass_attack(audio, sr=SAMPLE_RATE, cut_off=4000.0, order=6):
    """LowPassFilter.apply -- scripts/attacks.py:400-423 (causal lfilter, f64 out)."""
    b, a = butter(order, cut_off / (0.5 * sr), "low")
    return lfilter(b, a, audio)[0]


def highpass_attack(audio, sr=SAMPLE_RATE, cut_off=500.0, order=4):
    """HighPassFilter.apply -- scripts/attacks.py:438-455."""
    b, a = butter(order, cut_off / (0.5 * sr), "highpass")
    return lfilter(b, a, audio)[0]


def bandstop_attack(audio, sr=SAMPLE_RATE, f_low=None, band_width=200.0, min_freq=300.0,
                    max_freq=4000.0, order=4, rng=_pyrandom):
    """RandomBandstop.apply -- scripts/attacks.py:324-356 (python `random` draw)."""
    if f_low is None:
        f_low = rng.uniform(min_freq, max_freq - band_width)
    nyq = sr / 2.0
    b, a = butter(order, [f_low / nyq, (f_low + band_width) / nyq], "bandstop")
    audio = np.asarray(audio)
    return filtfilt(b, a, audio.astype(np.float64)).astype(audio.dtype)


def delete_samples_attack(audio, percentage, start=None):
    """DeleteSamples.apply -- scripts/attacks.py:162-178."""
    k = int(percentage * len(audio))
    if start is None:
        start = np.random.randint(0, len(audio) - k)
    return np.concatenate([audio[:start], audio[start + k:]])


def cropout_attack(audio, percentage, sr=SAMPLE_RATE):
    """Cropout.apply -- scripts/attacks.py:192-205."""
    return audio[int(percentage * sr):]


def sample_suppression_attack(audio, percentage, sr=SAMPLE_RATE, start=None):
    """SampleSupression.apply -- scripts/attacks.py:370-385."""
    k = int(percentage * sr)
    if start is None:
        start = np.random.randint(0, len(audio) - k)
    out = audio.copy()
    out[start:start + k] = 0
    return out


# ---- EXTENSIONS -- parity unpinned (not in the reference; BASELINE.json north_star) ----
def philox4x32(counter, key, rounds=10):
    """Philox-4x32-10 (Salmon et al. 2011), vectorised over `counter` [n,4] uint32."""
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
    c = counter.astype(np.uint32).copy()
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    for _ in range(rounds):
        p0 = M0 * c[:, 0].astype(np.uint64)
        p1 = M1 * c[:, 2].astype(np.uint64)
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        c = np.stack([hi1 ^ c[:, 1] ^ k0, lo1, hi0 ^ c[:, 3] ^ k1, lo0], axis=1)
        with np.errstate(over="ignore"):
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c


def gaussian_noise_attack(audio, snr_db=20.0, seed=0):
    """EXTENSION x1 -- additive white Gaussian noise at a target SNR.

    noise_i = sigma * z_i, sigma = sqrt(mean(x^2) / 10^(snr/10)); z from
    Philox-4x32-10 (key = (seed, 0x5eed), counter = (i//4, 0, 0, 0)), lanes
    paired through Box-Muller: u = (r + 0.5) / 2^32,
    z0 = sqrt(-2 ln u0) cos(2 pi u1), z1 = sqrt(-2 ln u0) sin(2 pi u1)."""
    x = np.asarray(audio, dtype=np.float32)
    n = x.shape[0]
    nblk = (n + 3) // 4
    ctr = np.zeros((nblk, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(nblk, dtype=np.uint32)
    r = philox4x32(ctr, (seed & 0xFFFFFFFF, 0x5EED))
    u = (r.astype(np.float64) + 0.5) / 4294967296.0
    rad0 = np.sqrt(-2.0 * np.log(u[:, 0]))
    rad1 = np.sqrt(-2.0 * np.log(u[:, 2]))
    z = np.stack([rad0 * np.cos(2 * np.pi * u[:, 1]), rad0 * np.sin(2 * np.pi * u[:, 1]),
                  rad1 * np.cos(2 * np.pi * u[:, 3]), rad1 * np.sin(2 * np.pi * u[:, 3])], axis=1)
    z = z.reshape(-1)[:n]
    power = np.mean(x.astype(np.float64) ** 2)
    sigma = math.sqrt(power / (10.0 ** (snr_db / 10.0)))
    return (x.astype(np.float64) + sigma * z).astype(np.float32)


def mp3_surrogate_attack(audio, n_levels_db=1.5, floor_db=-60.0):
    """EXTENSION x2 -- MP3-like quantisation surrogate (not a codec).

    STFT (1024/256, hann) -> per-bin magnitude quantised on a log grid of
    `n_levels_db` dB, bins more than `floor_db` below the frame maximum zeroed
    (masking surrogate), phase kept -> iSTFT.  Output length 256*(T-1)."""
    x = torch.as_tensor(np.asarray(audio, dtype=np.float32))
    S = stft(x)
    mag = torch.abs(S)
    ph = torch.angle(S)
    fmax = mag.amax(dim=0, keepdim=True).clamp_min(1e-12)
    db = 20.0 * torch.log10(mag.clamp_min(1e-12) / fmax)
    q = torch.round(db / n_levels_db) * n_levels_db
    mq = fmax * torch.pow(10.0, q / 20.0)
    mq = torch.where(db < floor_db, torch.zeros_like(mq), mq)
    return istft(mq * torch.exp(1j * ph)).numpy()


def mp3_surrogate_spectrum(S, n_levels_db=1.5, floor_db=-60.0):
    """EXTENSION x2 as a DIFFERENTIABLE op on a spectrum S [513, T] (complex): the quantiser of mp3_surrogate_attack with a
    straight-through magnitude path -- d(quantised magnitude)/d|S| := 1 on kept bins, 0 on bins dropped below the floor, the
    frame maximum a constant -- and the exact phase path.  Forward values equal mp3_surrogate_attack's spectrum; under torch
    autograd this function IS the specification of aware_spectral_quantize_bwd."""
    mag = torch.abs(S)
    ph = torch.angle(S)
    with torch.no_grad():
        fmax = mag.amax(dim=0, keepdim=True).clamp_min(1e-12)
        db = 20.0 * torch.log10(mag.clamp_min(1e-12) / fmax)
        q = torch.round(db / n_levels_db) * n_levels_db
        mq = fmax * torch.pow(10.0, q / 20.0)
        keep = (~(db < floor_db)) & (mag > 0)
        mq = torch.where(keep, mq, torch.zeros_like(mq))
    ste = torch.where(keep, mag + (mq - mag).detach(), torch.zeros_like(mag))
    return ste * torch.exp(1j * ph)


def phase_vocoder(D, rate, hop=256, n_fft=1024):
    """EXTENSION -- textbook phase vocoder (librosa.phase_vocoder's algorithm) on a one-sided STFT
    D [513, T] complex: output frame t sits at input position t*rate; magnitudes interpolated linearly between
    the two neighbouring frames (zero frame past the end), phase accumulated from the per-bin phase increment minus
    the expected advance 2*pi*hop*k/n_fft, wrapped to [-pi, pi] with round-half-even.  float64 arithmetic."""
    D = np.asarray(D).astype(np.complex128)
    F, T = D.shape
    steps = np.arange(0, T, float(rate))
    adv = 2.0 * np.pi * hop * np.arange(F) / n_fft
    Dp = np.concatenate([D, np.zeros((F, 2), dtype=D.dtype)], axis=1)
    out = np.zeros((F, len(steps)), dtype=np.complex128)
    acc = np.angle(D[:, 0])
    for t, step in enumerate(steps):
        i = int(np.floor(step))
        alpha = step - i
        c0, c1 = Dp[:, i], Dp[:, i + 1]
        mag = (1.0 - alpha) * np.abs(c0) + alpha * np.abs(c1)
        out[:, t] = mag * np.exp(1j * acc)
        dp = np.angle(c1) - np.angle(c0) - adv
        dp = dp - 2.0 * np.pi * np.round(dp / (2.0 * np.pi))
        acc = acc + adv + dp
    return out.astype(np.complex64)


def time_stretch_attack(audio, rate):
    """EXTENSION in place of scripts/attacks.py:208-228 (rubberband): STFT -> phase_vocoder -> iSTFT."""
    x = torch.as_tensor(np.asarray(audio, dtype=np.float32))
    S = stft(x).numpy()
    return istft(torch.from_numpy(phase_vocoder(S, rate))).numpy()


def pitch_shift_attack(audio, cents):
    """EXTENSION in place of scripts/attacks.py:231-252: stretch by 2^(cents/1200) then polyphase resampling back
    (ratio from Fraction(1/factor).limit_denominator(512), scipy.signal.resample_poly)."""
    from fractions import Fraction
    from scipy.signal import resample_poly as _rp
    factor = 2.0 ** ((cents / 100.0) / 12.0)
    fr = Fraction(1.0 / factor).limit_denominator(512)
    y = time_stretch_attack(audio, 1.0 / factor)
    return y if fr.numerator == fr.denominator else _rp(y.astype(np.float32), fr.numerator, fr.denominator).astype(np.float32)
