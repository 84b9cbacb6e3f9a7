import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from aware_amd import runtime as rt
from test_gpu_kernels import _block_reference
for (B, Tp, N, K, CL) in [(40, 94, 512, 256, 32), (40, 94, 512, 256, 40), (40, 94, 1024, 256, 32), (40,94,512,512,32), (40,94,512,128,32)]:
    RP = 32 * ((Tp + 31) // 32)
    g = torch.Generator().manual_seed(B + Tp + N + K + CL)
    a = torch.randn(B * RP, K, generator=g); a.view(B, RP, K)[:, Tp:] = 0
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g) * 0.1
    wl = torch.randn(CL, N, generator=g) / N ** 0.5
    ref = _block_reference(a, w, bias, None, None, B, RP, Tp, 1)
    c, rs, zp = rt.gemm_clip_last(a.cuda(), w, bias.cuda(), wl, B, Tp)
    c1, _ = rt.gemm_clip(a.cuda(), w.cuda(), bias.cuda(), B, Tp, 1, None, None, 1)
    c = c.cpu().view(B, RP, N)[:, :Tp].double(); c1 = c1.cpu().view(B, RP, N)[:, :Tp].double()
    e = (c - ref).abs(); e1 = (c1 - ref).abs()
    print((B,Tp,N,K,CL), "last err", e.max().item(), "epi1 err", e1.max().item())
    bad = (e > 1e-3).nonzero()
    if len(bad):
        print(" bad count", len(bad), "clips", bad[:,0].unique()[:10].tolist(), "rows", bad[:,1].unique()[:10].tolist(), "cols", bad[:,2].unique()[:20].tolist(), bad[:,2].max().item())
