"""Experiment: run two half-batches on two streams so DSP kernels overlap the other half's GEMMs."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aware_amd import runtime as rt
from aware_amd.detection import AWAREDetectorNet
B = int(sys.argv[1]); iters = int(sys.argv[2]); nsplit = int(sys.argv[3])
n = 48000
plan = rt.Plan()
det = AWAREDetectorNet().device_weights(plan)
g = torch.Generator(device="cuda").manual_seed(0)
streams = [torch.cuda.Stream() for _ in range(nsplit)]
sess = []
per = B // nsplit
for s in streams:
    with torch.cuda.stream(s):
        batch = rt.Batch([n] * per)
        audio = 0.1 * torch.randn(per * n, device="cuda", generator=g)
        target = (torch.randint(0, 2, (per, 20), device="cuda", generator=g).float() * 2 - 1)
        se = rt.EmbedSession(plan, det, batch, use_graph=True)
        se.begin(audio, target)
        se.iterate(16)
        sess.append((se, batch, audio, target))
torch.cuda.synchronize()
t0 = time.time()
chunk = 16
for it in range(0, iters, chunk):
    for s, (se, *_r) in zip(streams, sess):
        with torch.cuda.stream(s):
            se.iterate(chunk)
torch.cuda.synchronize()
dt = (time.time() - t0) / iters
print(f"B={B} split={nsplit}: {dt*1e3:.3f} ms/iter -> {B*3/(dt*400):.1f} wf-s/s")
