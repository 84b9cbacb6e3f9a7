"""Clip-aligned conv block: f32-MFMA (mode 0), bf16 three-term / six-product (mode 1), f16 two-term / three-product (h2).
usage: python tools/clip_h2_bench.py [B] [Tp]  -- column-relative error vs fp64 and microseconds per launch, detector shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from aware_amd import runtime as rt
from aware_amd.runtime import _ptr, _stream, check, load_library

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Tp = int(sys.argv[2]) if len(sys.argv) > 2 else 94
RP = 32 * ((Tp + 31) // 32)
g = torch.Generator().manual_seed(1)
lib = load_library()


def bench(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (N, K, epi) in [(512, 128, 1), (1024, 512, 1), (1024, 1024, 1), (1024, 1024, 2), (512, 1024, 2), (1024, 1024, 0)]:
    a = torch.randn(B * RP, K, generator=g)
    a.view(B, RP, K)[:, Tp:] = 0
    w = torch.randn(N, K, generator=g) * torch.exp2(torch.randint(-6, 4, (N, 1), generator=g).float()) / K ** 0.5
    bias = torch.randn(N, generator=g) * 0.1 if epi != 2 else None
    act = torch.randn(B * RP, N, generator=g) if epi == 2 else None
    rstd = (torch.rand(B, N, generator=g) + 0.5) if epi == 2 else None
    ad, wd = a.cuda(), w.cuda()
    bd = bias.cuda() if bias is not None else None
    actd = act.cuda() if act is not None else None
    rd = rstd.cuda() if rstd is not None else torch.zeros(B, N, device="cuda")
    pk = rt.x3_pack(w)
    z = (a.double() @ w.double().T).view(B, RP, N)[:, :Tp]
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
    scale = ref.abs().amax(dim=(0, 1), keepdim=True).clamp_min(1e-30)
    outs = [rt.gemm_clip(ad, wd, bd, B, Tp, epi, rd.clone(), actd, m, pk)[0] for m in (0, 1)]
    outs.append(rt.gemm_clip_h2(ad, wd, bd, B, Tp, epi, rd.clone(), actd)[0])
    errs = [((o.cpu().view(B, RP, N)[:, :Tp].double() - ref).abs() / scale).max().item() for o in outs]
    rms = [(((o.cpu().view(B, RP, N)[:, :Tp].double() - ref) / scale) ** 2).mean().sqrt().item() for o in outs]
    # timing: h2 with the weights packed once and the maxima computed once, as in the embed loop
    nbytes = int(lib.aware_gemm_clip_h2_workspace_bytes(B, N, K))
    t = [bench(lambda: rt.gemm_clip(ad, wd, bd, B, Tp, epi, rd, actd, m, pk)) for m in (0, 1)]
    t.append(bench(lambda: rt.gemm_clip_h2(ad, wd, bd, B, Tp, epi, rd, actd)))
    fl = 2.0 * B * Tp * N * K
    print(f"N={N:5d} K={K:5d} epi={epi}  max err f32 {errs[0]:.2e} x3 {errs[1]:.2e} h2 {errs[2]:.2e} | rms {rms[0]:.2e} {rms[1]:.2e} {rms[2]:.2e}"
          f" | us (TF f32-eq): f32 {t[0]:6.1f} ({fl / t[0] / 1e6:5.1f})  x3 {t[1]:6.1f} ({fl / t[1] / 1e6:5.1f})  "
          f"h2 incl. pack+amax+sync {t[2]:6.1f}", flush=True)
