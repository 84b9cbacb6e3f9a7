#!/usr/bin/env python3
"""Round-2 additions to tests/golden/ produced by running the reference itself (dev container only; test
infrastructure, same import recipe as tools/make_golden.py, which stays the generator of the round-1 fixtures):

  attacks_r2.npz
    * Resample's decimate + np.interp branch (scripts/attacks.py:275-288): Resample(8000) at sr = 16000 (factor 2)
      and Resample(16000) at sr = 48000 (factor 3) on the 1 s watermarked clip of attacks_1s.npz;
    * DeleteSamples(0.15) (the harness's third delete setting, scripts/test.py:15-18) on the same clip;
    * every in-scope attack on the reference's own 3 s watermarked clip (seed 0, the clip of embed_3s.npz), with the
      detector's raw outputs after the attack (outputs stored as every 8th sample plus float64 checksums).

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_r2.py
"""
import os
import sys
import random

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import torch

import make_golden as G


def main():
    G._setup_import_path()
    import matplotlib
    matplotlib.use("Agg")
    from aware.utils.models import load
    from aware.service import embed_watermark
    import attacks as A

    torch.set_num_threads(8)
    embedder, detector = load()
    g1 = np.load(os.path.join(G.OUT, "attacks_1s.npz"))
    src1 = g1["src"]
    d = {}

    def record(key, a, src, sr):
        np.random.seed(1234)
        random.seed(1234)
        out = np.asarray(a.apply(src.copy(), sr))
        np.random.seed(1234)
        random.seed(1234)
        if key.split("/")[-1].startswith("bandstop"):
            d[key + "/f_low"] = random.uniform(a.min_freq, a.max_freq - a.band_width)
        if key.split("/")[-1].startswith("delete"):
            d[key + "/start"] = int(np.random.randint(0, len(src) - int(a.percentage * len(src))))
        if key.split("/")[-1].startswith("suppress"):
            d[key + "/start"] = int(np.random.randint(0, len(src) - int(a.percentage * sr)))
        d[key + "/dtype"] = str(out.dtype)
        d[key + "/len"] = out.shape[0]
        d[key + "/sum"] = G.f64sum(out)
        d[key + "/abs_sum"] = G.f64sum(np.abs(out))
        if key.startswith("3s/"):
            d[key + "/out_sample"] = out[::8].astype(np.float32)          # every 8th sample + the two checksums above
        else:
            d[key + "/out"] = out if "decimate" in key else out.astype(np.float32)       # decimate: float64 as returned
        if sr == 16000:
            d[key + "/det_raw"] = detector.detect(out.astype(np.float32), 16000).astype(np.float32)

    record("1s/decimate2", A.Resample(8000), src1, 16000)
    record("1s/decimate3", A.Resample(16000), src1, 48000)
    record("1s/delete_0.15", A.DeleteSamples(0.15), src1, 16000)

    audio, bits = G.make_clip(0, 48000)
    wm3 = embed_watermark(audio, 16000, bits, embedder).astype(np.float32)
    g3 = np.load(os.path.join(G.OUT, "embed_3s.npz"))
    # the same clip as embed_3s.npz (trajectories of two CPU runs agree to ~1e-3, bits exactly)
    print("3 s clip: rel L2 distance to embed_3s.npz out_sample:",
          float(np.linalg.norm(wm3[::16] - g3["out_sample"]) / np.linalg.norm(g3["out_sample"])))
    d["3s/src"] = wm3
    d["3s/bits"] = bits
    d["3s/det_raw_clean"] = detector.detect(wm3, 16000).astype(np.float32)
    for key, a in [("pcm_8", A.PCMBitDepthConversion(8)), ("pcm_16", A.PCMBitDepthConversion(16)),
                   ("resample", A.Resample()), ("low_pass", A.LowPassFilter()), ("high_pass", A.HighPassFilter()),
                   ("bandstop", A.RandomBandstop()), ("delete_0.1", A.DeleteSamples(0.1)),
                   ("delete_0.15", A.DeleteSamples(0.15)), ("delete_0.2", A.DeleteSamples(0.2)),
                   ("cropout_0.1", A.Cropout(0.1)), ("suppress_0.1", A.SampleSupression(0.1)),
                   ("suppress_0.25", A.SampleSupression(0.25))]:
        record("3s/" + key, a, wm3, 16000)
    np.savez_compressed(os.path.join(G.OUT, "attacks_r2.npz"), **d)
    print("written", os.path.join(G.OUT, "attacks_r2.npz"))
    for k in sorted(d):
        if k.endswith("/det_raw"):
            b = (d[k] > 0).astype(np.int32)
            print(f"  {k:28s} bit errors vs embedded: {int((b != bits).sum()) if k.startswith('3s') else '-'}  min|raw| {np.abs(d[k]).min():.3f}")


if __name__ == "__main__":
    if not os.path.isdir(G.REF):
        sys.exit("reference tree not present: this script only runs in the dev container")
    main()
