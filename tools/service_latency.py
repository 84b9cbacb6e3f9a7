import time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from aware_amd.utils.models import load
from aware_amd.service import embed_watermark, detect_watermark
import torch
emb, det = load()
rng = np.random.default_rng(0)
a = (0.1 * rng.standard_normal(48000)).astype(np.float32)
bits = rng.integers(0, 2, 20).astype(np.int32)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    w = embed_watermark(a, 16000, bits, emb)
    torch.cuda.synchronize(); t1 = time.time()
    b = detect_watermark(w, 16000, det)
    torch.cuda.synchronize(); t2 = time.time()
    print(f"call {i}: embed {1e3*(t1-t0):.1f} ms, detect {1e3*(t2-t1):.1f} ms, bits ok {np.array_equal(b, bits)}")
a2 = (0.1 * rng.standard_normal(30000)).astype(np.float32)
torch.cuda.synchronize(); t0 = time.time()
w = embed_watermark(a2, 16000, bits, emb)
torch.cuda.synchronize(); print(f"new length: embed {1e3*(time.time()-t0):.1f} ms")
