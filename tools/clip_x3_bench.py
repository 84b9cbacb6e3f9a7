"""Clip-aligned conv block: f32-MFMA kernel (mode 0) vs bf16 three-way split kernel (mode 1).
usage: python tools/clip_x3_bench.py [B] [Tp]   -- accuracy vs fp64 and time per launch for the detector's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aware_amd import runtime as rt

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Tp = int(sys.argv[2]) if len(sys.argv) > 2 else 94
RP = 32 * ((Tp + 31) // 32)
g = torch.Generator().manual_seed(1)


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (N, K, epi) in [(512, 128, 1), (1024, 512, 1), (1024, 1024, 1), (1024, 40, 2), (1024, 1024, 2), (512, 1024, 2), (1024, 1024, 0)]:
    a = torch.randn(B * RP, K, generator=g)
    a.view(B, RP, K)[:, Tp:] = 0
    w = torch.randn(N, K, generator=g) * (1.0 / K ** 0.5)
    bias = torch.randn(N, generator=g) * 0.1 if epi != 2 else None
    act = torch.randn(B * RP, N, generator=g) if epi == 2 else None
    rstd = (torch.rand(B, N, generator=g) + 0.5) if epi == 2 else None
    ad, wd = a.cuda(), w.cuda()
    bd = bias.cuda() if bias is not None else None
    actd = act.cuda() if act is not None else None
    pk = {0: None, 1: rt.x3_pack(w)}
    outs = []
    modes = (0, 1) if K % 64 == 0 else (0,)
    for mode in modes:
        r = rstd.cuda() if rstd is not None else None
        c, rs = rt.gemm_clip(ad, wd, bd, B, Tp, epi, r, actd, mode, pk[mode])
        outs.append(c.cpu().double())
    # fp64 reference
    z = a.double() @ w.double().T
    z = z.view(B, RP, N)[:, :Tp]
    if epi == 0:
        ref = z + bias.double()
    elif epi == 1:
        z = z + bias.double()
        u = (z - z.mean(1, keepdim=True)) / torch.sqrt(z.var(1, unbiased=False, keepdim=True) + 1e-5)
        ref = torch.where(u > 0, u, 0.2 * u)
    else:
        av = act.double().view(B, RP, N)[:, :Tp]
        u = torch.where(av > 0, av, av * 5.0)
        du = z * torch.where(av > 0, 1.0, 0.2)
        ref = rstd.double()[:, None, :] * (du - du.mean(1, keepdim=True) - u * (du * u).mean(1, keepdim=True))
    errs = [(o.view(B, RP, N)[:, :Tp] - ref).abs().max().item() / ref.abs().max().item() for o in outs]
    pad = [o.view(B, RP, N)[:, Tp:].abs().max().item() if RP > Tp else 0.0 for o in outs]
    t = []
    for mode in modes:
        r = rstd.cuda() if rstd is not None else torch.zeros(B, N, device="cuda")
        t.append(bench(lambda: rt.gemm_clip(ad, wd, bd, B, Tp, epi, r, actd, mode, pk[mode])))
    fl = 2.0 * B * Tp * N * K
    print(f"N={N:5d} K={K:5d} epi={epi}  relerr " + " ".join(f"{e:.2e}" for e in errs) + f"  pad {max(pad)}  time us (TF): "
          + "  ".join(f"m{m} {tt:6.1f} ({fl / tt / 1e6:5.1f})" for m, tt in zip(modes, t)), flush=True)
