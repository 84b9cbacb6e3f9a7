#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the reference itself (dev container only).

This script is TEST INFRASTRUCTURE.  It runs only where /root/reference exists
(the development container); nothing here travels to, or is imported on, the
GPU box.  It follows the recipe recorded in SURVEY.md section 8(c):

  * PYTHONDONTWRITEBYTECODE=1, so no __pycache__ is written into /root/reference;
  * a temporary alias directory on sys.path with the symlink
        aware -> /root/reference/src/AWARE
    (the reference's package directory is upper-case, its imports lower-case);
  * three third-party modules the image lacks are replaced by minimal stand-ins,
    created in the same temporary directory and never committed as libraries:
        librosa.fft_frequencies(sr, n_fft) = np.linspace(0, sr/2, 1 + n_fft//2)
            -- the same formula the reference itself defines in
               src/AWARE/detection/modules/mel.py:72-74;
        webrtcvad.Vad(...).is_speech(...) -> True   (VAD gate bypassed:
            the gate's parity is UNPINNED);
        resampy  -- imported by utils/audio/waveform.py:5, never called.
    For scripts/attacks.py additionally: soundfile, pyrubberband (never called by
    the attacks we pin; MP3/TimeStretch/PitchShift are out of scope).

Inputs regenerate from seeds and are not stored:
    rng = np.random.default_rng(seed)
    audio = (0.1 * rng.standard_normal(n)).astype(np.float32)
    bits  = rng.integers(0, 2, 20).astype(np.int32)

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import os
import sys
import json
import random
import tempfile
import textwrap

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _setup_import_path():
    d = tempfile.mkdtemp(prefix="aware_alias_")
    os.symlink(os.path.join(REF, "src", "AWARE"), os.path.join(d, "aware"))
    os.makedirs(os.path.join(d, "librosa"))
    with open(os.path.join(d, "librosa", "__init__.py"), "w") as f:
        f.write(textwrap.dedent("""
            import numpy as np
            def fft_frequencies(sr=22050, n_fft=2048):
                return np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)
            """))
    with open(os.path.join(d, "librosa", "display.py"), "w") as f:
        f.write("")
    with open(os.path.join(d, "webrtcvad.py"), "w") as f:
        f.write(textwrap.dedent("""
            class Vad:
                def __init__(self, mode=0): self.mode = mode
                def is_speech(self, buf, sample_rate): return True
            """))
    for name in ("resampy", "soundfile", "pyrubberband"):
        with open(os.path.join(d, name + ".py"), "w") as f:
            f.write("")
    sys.path.insert(0, d)
    sys.path.insert(1, os.path.join(REF, "scripts"))
    return d


def make_clip(seed, n):
    rng = np.random.default_rng(seed)
    audio = (0.1 * rng.standard_normal(n)).astype(np.float32)
    bits = rng.integers(0, 2, 20).astype(np.int32)
    return audio, bits


def f64sum(x):
    return float(np.sum(np.asarray(x, dtype=np.float64)))


def main():
    _setup_import_path()
    import matplotlib
    matplotlib.use("Agg")
    from aware.utils.models import load
    from aware.service import embed_watermark, detect_watermark
    from aware.utils.utils import to_tensor

    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    meta = {"torch": torch.__version__, "numpy": np.__version__}
    import scipy
    meta["scipy"] = scipy.__version__

    embedder, detector = load()
    net = embedder.detection_net

    # ---- (i) weights + mel basis -------------------------------------------------
    wsum = {}
    for name, p in net.named_parameters():
        wsum[name] = f64sum(p.detach().numpy())
    mel = net.mel_layer.mel_filter_bank.numpy()
    g = {
        "mel_basis_sum": f64sum(mel),
        "mel_basis_rows_sum": mel.astype(np.float64).sum(axis=1),
        "mel_basis_sample": mel[::8, 24:264:4].copy(),
    }
    for k, v in wsum.items():
        g["wsum/" + k] = v
    # sampled weights (first 4x8 corner of every conv) as known answers
    for i, blk in enumerate(net.conv_blocks):
        g[f"w{i}_corner"] = blk.conv.weight.detach().numpy()[:4, :8, 0].copy()
    np.savez(os.path.join(OUT, "weights.npz"), **g)

    # ---- (ii)/(iii) STFT / iSTFT -------------------------------------------------
    for tag, seed, n in (("1s", 1, 16000), ("3s", 0, 48000)):
        audio, bits = make_clip(seed, n)
        x = to_tensor(audio)
        pre = embedder.audio_preprocess_pipeline
        xn = pre[0](x)                       # WaveformNormalizer
        S = pre[1](xn)                       # STFT  [513, T] complex64
        mag, ph = pre[2](S)
        post = embedder.audio_postprocess_pipeline
        y = post[1](post[0](mag, ph))        # Assembler + ISTFT (un-normalised)
        yn = post[2](y)
        Sn = S.numpy()
        d = {
            "seed": seed, "n": n,
            "T": Sn.shape[1],
            "stft_abs_sum": f64sum(np.abs(Sn)),
            "stft_re_sum": f64sum(Sn.real), "stft_im_sum": f64sum(Sn.imag),
            "stft_cols": Sn[:, [0, 1, 2, Sn.shape[1] // 2, Sn.shape[1] - 2, Sn.shape[1] - 1]].copy(),
            "stft_rows": Sn[[0, 32, 100, 256, 400, 512], :].copy(),
            "istft_len": y.shape[0],
            "istft_sum": f64sum(y.numpy()), "istft_abs_sum": f64sum(np.abs(y.numpy())),
            "istft_head": y.numpy()[:1024].copy(), "istft_tail": y.numpy()[-1024:].copy(),
            "istft_mid": y.numpy()[y.shape[0] // 2 - 512: y.shape[0] // 2 + 512].copy(),
            "istft_norm_max": float(yn.abs().max()),
            "roundtrip_maxerr": float((y - xn[: y.shape[0]]).abs().max()),
        }
        np.savez(os.path.join(OUT, f"stft_{tag}.npz"), **d)

    # ---- (iv)-(vii) detector, first-iteration gradient, trajectory, final bits ----
    import librosa
    for tag, seed, n in (("1s", 1, 16000), ("3s", 0, 48000)):
        audio, bits = make_clip(seed, n)
        # unmarked detector output
        raw_unmarked = detector.detect(audio, 16000).astype(np.float32)

        # first-iteration loss + gradient using the reference's own objects
        from aware.utils.watermark import PatternEncoder
        wm = PatternEncoder(mode=embedder.pattern_mode)(bits)
        x = to_tensor(audio)
        for p in embedder.audio_preprocess_pipeline:
            x = p(x)
        magnitude, phase = x
        fi, nfi = embedder._get_embedding_frequency_indices(16000, embedder.frame_length)
        coeffs0 = magnitude[fi].flatten().clone()
        delta = coeffs0 * 10 ** (-embedder.tolerance_db / 20)
        lo = torch.clamp(coeffs0 - delta, min=0)
        hi = coeffs0 + delta
        for prm in net.parameters():
            prm.requires_grad = False
        c = coeffs0.clone().requires_grad_(True)
        wmag = magnitude.clone()
        wmag[fi] = c.reshape(len(fi), -1)
        wmag = embedder._recompute_watermarked_magnitude(wmag, phase)
        wmag[nfi] = 0.0
        pred = net(wmag.unsqueeze(0)).squeeze()
        loss = embedder.loss(pred, to_tensor(wm))
        loss.backward()
        grad = c.grad.detach().numpy().reshape(len(fi), -1)

        # trajectory: record every loss the reference computes (wrap, do not modify)
        losses = []
        orig_loss = embedder.loss

        class _Rec:
            def __call__(self, p, t):
                v = orig_loss(p, t)
                losses.append(float(v.detach()))
                return v

        embedder.loss = _Rec()
        wm_audio = embed_watermark(audio, 16000, bits, embedder)
        embedder.loss = orig_loss
        raw_marked = detector.detect(wm_audio, 16000).astype(np.float32)
        det_bits = detect_watermark(wm_audio, 16000, detector)

        step = 1 if n <= 16000 else 16
        d = {
            "seed": seed, "n": n, "bits": bits, "wm_bipolar": np.asarray(wm),
            "band_idx": np.asarray(fi), "nonband_idx": np.asarray(nfi),
            "raw_unmarked": raw_unmarked,
            "coeffs0_sum": f64sum(coeffs0.numpy()),
            "bound_hi_max": float(hi.max()), "bound_lo_min": float(lo.min()),
            "iter1_pred": pred.detach().numpy(),
            "iter1_loss": float(loss.detach()),
            "iter1_grad_sum": f64sum(grad), "iter1_grad_abs_sum": f64sum(np.abs(grad)),
            "iter1_grad_sample": grad[:, ::step].copy(),
            "grad_step": step,
            "losses": np.asarray(losses, dtype=np.float64),
            "out_len": wm_audio.shape[0],
            "out_sum": f64sum(wm_audio), "out_abs_sum": f64sum(np.abs(wm_audio)),
            "out_max": float(np.max(wm_audio)), "in_max": float(np.max(audio)),
            "out_sample": wm_audio[::step].astype(np.float32).copy(),
            "out_step": step,
            "raw_marked": raw_marked,
            "det_bits": np.asarray(det_bits),
        }
        np.savez(os.path.join(OUT, f"embed_{tag}.npz"), **d)
        print(tag, "bits", bits.tolist(), "det", np.asarray(det_bits).tolist(),
              "loss1", d["iter1_loss"], "lossN", losses[-1], "best", min(losses))
        if tag == "1s":
            wm_1s = wm_audio.astype(np.float32)

    # ---- config 1: 44.1 kHz clip -> resample_poly -> embed -> detect ---------------
    from scipy.signal import resample_poly
    rng = np.random.default_rng(0)
    a441 = (0.1 * rng.standard_normal(132300)).astype(np.float32)
    bits = rng.integers(0, 2, 20).astype(np.int32)
    a16 = resample_poly(a441, 16000, 44100)
    wm_audio = embed_watermark(a16, 16000, bits, embedder)
    det_bits = detect_watermark(wm_audio, 16000, detector)
    raw = detector.detect(wm_audio, 16000)
    np.savez(os.path.join(OUT, "config1_44k.npz"),
             bits=bits, a16_dtype=str(a16.dtype), a16_len=a16.shape[0],
             a16_sum=f64sum(a16), a16_abs_sum=f64sum(np.abs(a16)),
             a16_sample=np.asarray(a16[::16], dtype=np.float64),
             out_len=wm_audio.shape[0], out_sum=f64sum(wm_audio),
             out_sample=wm_audio[::16].astype(np.float32),
             raw_marked=raw.astype(np.float32), det_bits=np.asarray(det_bits))
    print("config1 bits", bits.tolist(), "det", np.asarray(det_bits).tolist())

    # ---- (viii) attacks on the 1 s watermarked clip ---------------------------------
    import attacks as A
    src = wm_1s
    d = {"src_sum": f64sum(src), "src_len": src.shape[0]}
    atk = [
        ("pcm_8", A.PCMBitDepthConversion(8)), ("pcm_12", A.PCMBitDepthConversion(12)),
        ("pcm_16", A.PCMBitDepthConversion(16)), ("pcm_24", A.PCMBitDepthConversion(24)),
        ("resample", A.Resample()), ("low_pass", A.LowPassFilter()),
        ("high_pass", A.HighPassFilter()), ("bandstop", A.RandomBandstop()),
        ("delete_0.1", A.DeleteSamples(0.1)), ("delete_0.2", A.DeleteSamples(0.2)),
        ("cropout_0.1", A.Cropout(0.1)),
        ("suppress_0.1", A.SampleSupression(0.1)), ("suppress_0.25", A.SampleSupression(0.25)),
    ]
    names = {}
    for key, a in atk:
        np.random.seed(1234)
        random.seed(1234)
        out = a.apply(src.copy(), 16000)
        # record the random draws the attack consumed so the port can replay them
        np.random.seed(1234)
        random.seed(1234)
        if key.startswith("bandstop"):
            d[key + "/f_low"] = random.uniform(a.min_freq, a.max_freq - a.band_width)
        if key.startswith("delete"):
            d[key + "/start"] = int(np.random.randint(0, len(src) - int(a.percentage * len(src))))
        if key.startswith("suppress"):
            d[key + "/start"] = int(np.random.randint(0, len(src) - int(a.percentage * 16000)))
        out = np.asarray(out)
        d[key + "/dtype"] = str(out.dtype)
        d[key + "/len"] = out.shape[0]
        d[key + "/sum"] = f64sum(out)
        d[key + "/abs_sum"] = f64sum(np.abs(out))
        d[key + "/out"] = out.astype(np.float32)
        d[key + "/det_raw"] = detector.detect(out.astype(np.float32), 16000).astype(np.float32)
        names[key] = a.name
    d["src"] = src
    np.savez_compressed(os.path.join(OUT, "attacks_1s.npz"), **d)
    meta["attack_names"] = names
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("golden fixtures written to", os.path.abspath(OUT))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference tree not present: this script only runs in the dev container")
    main()
