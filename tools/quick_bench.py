"""Ad-hoc timing of the embed iteration (development aid, not the contract bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aware_amd import runtime as rt
from aware_amd.detection import AWAREDetectorNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
graph = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = 48000
plan = rt.Plan()
pipe = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] in ("f32", "bf16x3", "f16x2") else "f16x2"
dsp = sys.argv[6] if len(sys.argv) > 6 else "stream"
mel = os.environ.get("QB_MEL", "taps")
det = AWAREDetectorNet().device_weights(plan)
batch = rt.Batch([n] * B)
g = torch.Generator(device="cuda").manual_seed(0)
audio = 0.1 * torch.randn(B * n, device="cuda", generator=g)
target = (torch.randint(0, 2, (B, 20), device="cuda", generator=g).float() * 2 - 1)
sess = rt.EmbedSession(plan, det, batch, use_graph=bool(graph), conv_pipe=pipe, num_iterations=iters + 16, dsp_path=dsp, mel=mel)
sess.begin(audio, target)
sess.iterate(5)
torch.cuda.synchronize()
t0 = time.time()
sess.iterate(iters)
torch.cuda.synchronize()
dt = (time.time() - t0) / iters
print(f"B={B} graph={graph}: {dt*1e3:.3f} ms/iter -> embed 400 iters = {dt*400:.3f} s -> {B*3/(dt*400):.1f} wf-s/s; loss[0]={float(sess.loss[0]):.4f}")
if len(sys.argv) > 5:                                   # per-kernel breakdown of three eager loop bodies
    acc = {}
    for kind, ms in rt.embed_profile(sess, 3):
        acc[kind] = acc.get(kind, 0.0) + ms / 3
    print("  us per iteration by kind:", {k: round(v * 1e3, 1) for k, v in acc.items()})
