"""Launch one clip-aligned conv block shape repeatedly (profiling aid).
usage: python3 tools/x3_one.py N K epi mode [B] [Tp] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aware_amd import runtime as rt

N, K, epi, mode = (int(v) for v in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 64
Tp = int(sys.argv[6]) if len(sys.argv) > 6 else 94
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
RP = 32 * ((Tp + 31) // 32)
g = torch.Generator().manual_seed(1)
a = torch.randn(B * RP, K, generator=g).cuda()
w = torch.randn(N, K, generator=g) * (1.0 / K ** 0.5)
bias = torch.randn(N, generator=g).cuda()
act = torch.randn(B * RP, N, generator=g).cuda()
rstd = (torch.rand(B, N, generator=g) + 0.5).cuda()
pk = rt.x3_pack(w) if mode == 1 else None
wd = w.cuda()
for _ in range(reps):
    rt.gemm_clip(a, wd, bias if epi != 2 else None, B, Tp, epi, rstd, act if epi == 2 else None, mode, pk)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    rt.gemm_clip(a, wd, bias if epi != 2 else None, B, Tp, epi, rstd, act if epi == 2 else None, mode, pk)
e1.record()
torch.cuda.synchronize()
print(f"N={N} K={K} epi={epi} mode={mode} B={B}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us")
