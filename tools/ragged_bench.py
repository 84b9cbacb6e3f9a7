"""Ragged workload timing (BASELINE config 5 in miniature): clips of 1..10 s."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aware_amd import runtime as rt
from aware_amd.utils.models import load
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 48
uniform = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(0)
secs = rng.integers(1, 11, B) if not uniform else np.full(B, uniform)
if uniform and len(sys.argv) > 5:
    secs[0] += 1                                        # one odd clip: the batch takes the ragged route
lens = [int(s) * 16000 for s in secs]
emb, det = load()
batch = rt.Batch(lens)
audio = 0.1 * torch.randn(sum(lens), device="cuda")
target = (torch.randint(0, 2, (B, 20), device="cuda").float() * 2 - 1)
sess = emb.start_session(batch, 16000)
sess.begin(audio, target)
sess.iterate(16)
torch.cuda.synchronize()
t0 = time.time()
sess.iterate(iters)
torch.cuda.synchronize()
dt = (time.time() - t0) / iters
print(f"B={B} total {sum(secs)} s of audio: {dt*1e3:.3f} ms/iter -> {sum(secs)/(dt*400):.1f} wf-s/s (embed only)")
if len(sys.argv) > 4:                                   # per-kernel breakdown of three eager loop bodies
    acc = {}
    for kind, ms in rt.embed_profile(sess.session if hasattr(sess, "session") else sess, 3):
        acc[kind] = acc.get(kind, 0.0) + ms / 3
    print("  us per iteration by kind:", {k: round(v * 1e3, 1) for k, v in acc.items()})
