import sys, os, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import make_clip
from oracle import aware_oracle as O
from aware_amd import runtime as rt
lengths, seeds = [48000, 16000, 23456], [0, 1, 2]
plan = rt.Plan()
ws, bs = O.detector_weights()
det = rt.DetectorWeights(plan, O.mel_filter_bank(), [w.numpy() for w in ws], [b.numpy() for b in bs])
pairs = [make_clip(s, n) for s, n in zip(seeds, lengths)]
cl = [p[0] for p in pairs]
wm = np.stack([O.bits_to_bipolar(p[1]) for p in pairs]).astype(np.float32)
batch = rt.Batch(lengths)
for cfg in (4, 0):
    rt.tune(1, cfg)
    sess = rt.EmbedSession(plan, det, batch, use_graph=False)
    sess.begin(batch.pack(cl), torch.from_numpy(wm).cuda())
    g = sess.gradient(); torch.cuda.synchronize(); g = g.cpu()
    emb = O.Embedder()
    for i, c in enumerate(cl):
        a = torch.from_numpy(c)[None]
        mag0, phase = emb.analyse(a)
        c0 = mag0[:, emb.band].clone().requires_grad_(True)
        l, p = emb.forward_loss(c0, mag0, phase, torch.from_numpy(wm[i])[None])
        l.sum().backward()
        ref = c0.grad[0]
        mine = g[batch.frame_offsets[i]: batch.frame_offsets[i + 1], :225].T
        d = (mine - ref)
        print(os.environ.get("AWARE_HIP_LIB", "default")[-12:], "cfg", cfg, "clip", i, "rel", (d.norm() / ref.norm()).item(), "max|d|/max|ref|", (d.abs().max() / ref.abs().max()).item(), "argmax frame", int(d.abs().max(dim=0).values.argmax()), "of", ref.shape[1])
