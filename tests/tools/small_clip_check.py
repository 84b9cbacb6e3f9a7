"""Development check: first-iteration gradient of short clips, streaming vs staged DSP kernels vs the oracle in float64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_clip
from aware_amd import runtime as rt
from oracle import aware_oracle as O

plan = rt.Plan()
ws, bs = O.detector_weights()
det = rt.DetectorWeights(plan, O.mel_filter_bank(), [w.numpy() for w in ws], [b.numpy() for b in bs])
for n in (1100, 1300, 2000, 3000, 5000):
    a, bits = make_clip(94, n)
    wm = O.bits_to_bipolar(bits).astype(np.float32)[None]
    batch = rt.Batch([n])
    emb = O.Embedder(dtype=torch.float64)
    x = torch.from_numpy(a).double()[None]
    mag0, phase = emb.analyse(x)
    c0 = mag0[:, emb.band].clone().requires_grad_(True)
    l, p = emb.forward_loss(c0, mag0, phase, torch.from_numpy(wm).double())
    l.sum().backward()
    ref = c0.grad[0].T
    out = []
    for path in ("stream", "staged"):
        sess = rt.EmbedSession(plan, det, batch, use_graph=False, dsp_path=path)
        sess.begin(batch.pack([a]), torch.from_numpy(wm).cuda())
        g = sess.gradient().cpu().double()[:, :225]
        out.append((float((g - ref).norm() / ref.norm()), float(sess.loss.cpu()[0])))
    print(n, "T", batch.frames[0], "loss64 %.6f" % float(l), "stream rel %.2e loss %.6f | staged rel %.2e loss %.6f" % (out[0] + out[1]))
