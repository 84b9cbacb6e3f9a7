#!/usr/bin/env python3
"""Measured drift of the HIP embed loop against the REFERENCE's recorded trajectories (tests/golden/embed_1s.npz,
embed_3s.npz, config1_44k.npz: the reference's own CPU run).  Runs on the GPU box; writes gpurun_out/drift.json,
which is committed as profiles/rNN_drift.json and is what the 400-step tolerances in tests/ are derived from
(<= 3x the measured value).  All three matrix pipes are measured -- f16 two-term (default), bf16 three-term, f32-input MFMA --
with the golden clip alone (latency kernels) and inside a batch of 192 / 256 clips (the throughput kernels the bench times)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from conftest import GOLDEN, make_clip
from aware_amd import runtime as rt
from aware_amd.utils.models import load
from oracle import aware_oracle as O


def run(plan, det, audio, bits, pipe, graph, B=1, slot=0):
    """B > 1: the golden clip travels in slot `slot` of a uniform batch of B seeded clips, i.e. through the throughput kernels
    the bench times (conv blocks on gemm_clip_h2_kernel for the default pipe, mel block / read-out in their large-batch forms)."""
    n = len(audio)
    pairs = [make_clip(5000 + i, n) for i in range(B)]
    pairs[slot] = (audio, bits)
    wm = np.stack([O.bits_to_bipolar(p[1]) for p in pairs]).astype(np.float32)
    batch = rt.Batch([n] * B)
    sess = rt.EmbedSession(plan, det, batch, use_graph=graph, conv_pipe=pipe)
    sess.begin(batch.pack([p[0] for p in pairs]), torch.from_numpy(wm).cuda())
    losses = []
    for _ in range(400):
        sess.iterate(1)
        losses.append(sess.loss[slot: slot + 1].clone())
    losses = torch.cat(losses).cpu().numpy()
    outs = batch.unpack_out(sess.finish(torch.tensor([float(np.max(p[0])) for p in pairs], device="cuda")))
    out = outs[slot].contiguous()
    vals = rt.detect(plan, det, rt.Batch([out.numel()]), out).cpu().numpy()[0]
    return np.asarray(losses), float(sess.best_loss.cpu()[slot]), out.cpu().numpy(), vals, sess


def main():
    embedder, detector = load()
    plan = embedder._plan(16000)
    det = embedder.detection_net.device_weights(plan)
    res = {}
    for tag, seed, n in (("1s", 1, 16000), ("3s", 0, 48000)):
        e = np.load(os.path.join(GOLDEN, f"embed_{tag}.npz"))
        audio, bits = make_clip(seed, n)
        ref = e["losses"]
        big = 256 if tag == "3s" else 192
        for pipe, B, slot in (("f16x2", big, 77), ("bf16x3", big, 77), ("f32", big, 77), ("bf16x3", 1, 0), ("f32", 1, 0)):
            losses, best, out, vals, sess = run(plan, det, audio, bits, pipe, True, B, slot)
            d = np.abs(losses - ref)
            step = int(e["out_step"])
            rel = float(np.linalg.norm(out[::step] - e["out_sample"]) / np.linalg.norm(e["out_sample"]))
            r = {"loss_absdiff_step0": float(d[0]), "loss_absdiff_step20": float(d[20]), "loss_absdiff_step100": float(d[100]),
                 "loss_absdiff_step200": float(d[200]), "loss_absdiff_step399": float(d[399]), "loss_absdiff_max": float(d.max()),
                 "loss_absdiff_max_first20": float(d[:20].max()),
                 "argmax_step": int(d.argmax()), "best_loss": best, "best_loss_ref": float(ref.min()),
                 "best_absdiff": abs(best - float(ref.min())), "out_rel_l2": rel,
                 "raw_marked_maxabs_diff": float(np.max(np.abs(vals - e["raw_marked"]))),
                 "min_abs_raw": float(np.min(np.abs(vals))),
                 "bits_equal": bool(np.array_equal(O.decode_bits(vals), e["det_bits"]))}
            if pipe == "bf16x3" and B == 1:
                lo, hi = sess.bounds
                r["bound_hi_max"] = float(hi[:, :225].max())
                r["bound_hi_max_ref"] = float(e["bound_hi_max"])
                r["bound_lo_min"] = float(lo[:, :225].min())
                r["bound_lo_min_ref"] = float(e["bound_lo_min"])
            res[f"{tag}/{pipe}/B{B}"] = r
            print(tag, pipe, f"B={B}", json.dumps(r), flush=True)
    # config 1: 44.1 kHz front end
    c = np.load(os.path.join(GOLDEN, "config1_44k.npz"))
    rng = np.random.default_rng(0)
    a441 = (0.1 * rng.standard_normal(132300)).astype(np.float32)
    bits = rng.integers(0, 2, 20).astype(np.int32)
    from aware_amd.attacks import resample_poly_batch
    a16 = resample_poly_batch(rt.Ragged.from_list([a441]), 16000, 44100).to_list()[0]
    losses, best, out, vals, _ = run(plan, det, a16, bits, "f16x2", True)
    res["config1_44k/f16x2/B1"] = {"raw_marked_maxabs_diff": float(np.max(np.abs(vals - c["raw_marked"]))),
                                 "out_rel_l2": float(np.linalg.norm(out[::16] - c["out_sample"]) / np.linalg.norm(c["out_sample"])),
                                 "bits_equal": bool(np.array_equal(O.decode_bits(vals), c["det_bits"])), "best_loss": best}
    print("config1_44k", json.dumps(res["config1_44k/f16x2/B1"]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "drift.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
