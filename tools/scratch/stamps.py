import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aware_amd import runtime as rt
from aware_amd._lib import load_library
from aware_amd.detection import AWAREDetectorNet
from aware_amd.pipeline import synthetic_clips
B = 64
plan = rt.Plan()
det = AWAREDetectorNet().device_weights(plan)
batch = rt.Batch([48000] * B)
g = torch.Generator().manual_seed(0)
audio = (0.1 * torch.randn(B * 48000, generator=g)).cuda()
wm = (torch.randint(0, 2, (B, 20), generator=g).float() * 2 - 1).cuda()
sess = rt.EmbedSession(plan, det, batch, use_graph=False)
sess.begin(audio, wm)
for _ in range(5):
    sess.iterate(1)
torch.cuda.synchronize()
lib = load_library()
lib.aware_debug_stamps.argtypes = [ctypes.c_void_p]
out = (ctypes.c_ulonglong * 16)()
lib.aware_debug_stamps(out)
v = [int(x) for x in out][:6]
print("stamps (100 MHz ticks?) deltas:", [v[i + 1] - v[i] for i in range(5)], "total", v[5] - v[0])
