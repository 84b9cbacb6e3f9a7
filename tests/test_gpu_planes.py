"""gemm_h2p.hip -- the conv blocks of a uniform batch with the operands PRE-SPLIT between the layers (binary16 (h, l) fragment
images + one power-of-two scale per (clip, 128-column slab)) -- against fp64, beside the f32-input MFMA kernel on the same inputs,
through the C-ABI (aware_gemm_clip_h2p).  Reference semantics: detection/modules/conv1d.py:38-42 and its backward."""
import numpy as np
import pytest
import torch

from test_gpu_kernels import _block_reference

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rt():
    from aware_amd import runtime
    return runtime


# (B, Tp, N, K, epi, a_planes, out_planes): the five launches of the bench's iteration (conv0: f32 A split on the fly -> planes;
# conv1: planes -> planes; data gradients: planes -> planes / planes -> f32 rows), every row-group count (Tp = 31, 63, 94, 128),
# grids that are / are not a multiple of 8 workgroups (plain and slab-group-major tile walk), the plain epilogue
CASES = [(40, 94, 512, 128, 1, 0, 1), (256, 94, 512, 128, 1, 0, 1), (24, 94, 1024, 512, 1, 1, 1), (64, 94, 1024, 512, 1, 1, 1),
         (16, 94, 1024, 1024, 2, 1, 1), (64, 94, 1024, 1024, 2, 1, 1), (64, 94, 512, 1024, 2, 1, 0), (21, 94, 1024, 1024, 1, 1, 1),
         (21, 94, 1024, 1024, 2, 1, 0), (40, 94, 512, 128, 0, 1, 1), (72, 31, 512, 128, 1, 0, 1), (32, 63, 1024, 512, 2, 1, 1),
         (32, 128, 1024, 256, 1, 1, 1), (19, 128, 1024, 256, 2, 1, 0), (64, 94, 1024, 1024, 1, 1, 0), (5, 94, 512, 128, 1, 0, 0),
         (3, 31, 128, 1024, 1, 1, 1)]


@pytest.mark.parametrize("B,Tp,N,K,epi,apl,opl", CASES)
def test_gemm_clip_h2p(rt, B, Tp, N, K, epi, apl, opl):
    RP = 32 * ((Tp + 31) // 32)
    g = torch.Generator().manual_seed(B * 1000 + Tp + N + K + epi + 7)
    a = torch.randn(B * RP, K, generator=g)
    # slabs of very different magnitude inside a clip: the consumer rescales its accumulators between them
    # (the first slab stays at unit size or above so that the block's InstanceNorm is not a difference of nearly equal numbers)
    mag = torch.exp2(torch.randint(-12, 6, (1, K // 128, 1), generator=g).float())
    mag[0, 0, 0] = 2.0 ** float(torch.randint(0, 4, (1,), generator=g))
    a = (a.view(B * RP, K // 128, 128) * mag).reshape(B * RP, K)
    a.view(B, RP, K)[:, Tp:] = 0
    w = torch.randn(N, K, generator=g) * torch.exp2(torch.randint(-6, 4, (N, 1), generator=g).float()) / K ** 0.5
    bias = torch.randn(N, generator=g) * 0.1 if epi != 2 else None
    act = torch.randn(B * RP, N, generator=g) if epi == 2 else None
    if act is not None:
        act.view(B, RP, N)[:, Tp:] = 0
    rstd = torch.rand(B, N, generator=g) + 0.5 if epi == 2 else None
    ref = _block_reference(a, w, bias, act, rstd, B, RP, Tp, epi)
    cu = lambda t: None if t is None else t.cuda()
    c0, _ = rt.gemm_clip(a.cuda(), w.cuda(), cu(bias), B, Tp, epi, cu(rstd), cu(act), 0)              # f32-input MFMA
    c, rs, so = rt.gemm_clip_h2p(a.cuda(), w, cu(bias), B, Tp, epi, cu(rstd), cu(act), a_planes=bool(apl), out_planes=bool(opl))
    c = c.cpu().view(B, RP, N)
    assert RP == Tp or c[:, Tp:].abs().max().item() == 0.0                                            # padding rows are zero
    scale = ref.abs().amax(dim=(0, 1), keepdim=True).clamp_min(1e-30)
    e32 = ((c0.cpu().view(B, RP, N)[:, :Tp].double() - ref).abs() / scale).max().item()
    eh = ((c[:, :Tp].double() - ref).abs() / scale).max().item()
    print(f"max column-relative error: f32 MFMA {e32:.2e}, planes {eh:.2e}")
    tol = 4e-6 * max(1.0, K / 256)
    assert eh < tol and eh < 2.0 * e32 + 4e-7, (e32, eh)
    if epi == 1:
        z = (a.double() @ w.double().T).view(B, RP, N)[:, :Tp] + bias.double()
        ref_rs = 1.0 / torch.sqrt(z.var(1, unbiased=False) + 1e-5)
        assert ((rs.cpu().double() - ref_rs).abs() / ref_rs).max().item() < 2e-5
    if opl:
        # the slab scales: a power of two that brings the slab's largest magnitude into [2^13, 2^14)
        so = so.cpu()[:, : N // 128].double()
        amax = c.abs().view(B, RP, N // 128, 128).amax(dim=(1, 3)).double()
        assert torch.all(torch.log2(so) == torch.round(torch.log2(so)))
        ok = (amax * so >= 2.0 ** 13 * (1 - 1e-6)) & (amax * so < 2.0 ** 14 * (1 + 1e-6))
        assert bool(torch.all(ok | (amax < 1e-30)))


@pytest.mark.parametrize("B,Tp,N,K,CL", [(16, 94, 1024, 1024, 40), (21, 94, 1024, 1024, 40), (64, 94, 1024, 1024, 40),
                                         (32, 63, 1024, 512, 40), (40, 94, 512, 256, 32), (128, 31, 1024, 128, 16)])
def test_gemm_clip_h2p_last_partials(rt, B, Tp, N, K, CL):
    """gemm_clip_h2p_kernel<RG, X3_FWD_LAST>: block 2 of the embed loop from a planes operand -- f32 rows out + the split-K
    partials of the skinny last conv (multibit_detector_net.py:58-70) -- against fp64."""
    RP = 32 * ((Tp + 31) // 32)
    g = torch.Generator().manual_seed(B + Tp + N + K + CL + 3)
    a = torch.randn(B * RP, K, generator=g)
    a.view(B, RP, K)[:, Tp:] = 0
    w = torch.randn(N, K, generator=g) * torch.exp2(torch.randint(-6, 4, (N, 1), generator=g).float()) / K ** 0.5
    bias = torch.randn(N, generator=g) * 0.1
    wl = torch.randn(CL, N, generator=g) * torch.exp2(torch.randint(-4, 3, (CL, 1), generator=g).float()) / N ** 0.5
    ref = _block_reference(a, w, bias, None, None, B, RP, Tp, 1)
    c, rs, _, zp = rt.gemm_clip_h2p(a.cuda(), w, bias.cuda(), B, Tp, 1, None, None, w_last=wl, a_planes=True, out_planes=False)
    c = c.cpu().view(B, RP, N)
    assert RP == Tp or c[:, Tp:].abs().max().item() == 0.0
    scale = ref.abs().amax(dim=(0, 1), keepdim=True).clamp_min(1e-30)
    err = ((c[:, :Tp].double() - ref).abs() / scale).max().item()
    assert err < 4e-6 * max(1.0, K / 256), err
    z = (a.double() @ w.double().T).view(B, RP, N)[:, :Tp] + bias.double()
    ref_rs = 1.0 / torch.sqrt(z.var(1, unbiased=False) + 1e-5)
    assert ((rs.cpu().double() - ref_rs).abs() / ref_rs).max().item() < 2e-5
    zsum = zp.cpu().double().sum(0).view(B, RP, CL)
    zref = ref @ wl.double().T
    zs = zref.abs().amax(dim=(0, 1), keepdim=True).clamp_min(1e-30)
    ez = ((zsum[:, :Tp] - zref).abs() / zs).max().item()
    print(f"block output err {err:.2e}; last-conv partial-sum err {ez:.2e}")
    assert ez < 4e-6 * max(1.0, N / 256) + 8 * err, ez
    assert RP == Tp or zsum[:, Tp:].abs().max().item() == 0.0


def test_planes_scale_spread_is_safe(rt):
    """Slabs of one clip 2^100 apart (a dead slab next to a live one, both orders): the accumulator rescaling is capped, nothing
    overflows, and the result equals the live slabs' contribution to f32 level."""
    B, Tp, N, K = 8, 94, 256, 512
    RP = 96
    g = torch.Generator().manual_seed(11)
    a = torch.randn(B * RP, K, generator=g)
    mag = torch.ones(B, 1, K // 128, 1)
    mag[0::2, 0, 1] = 2.0 ** -100          # clip 0, 2, ..: slab 1 dead after a live slab 0
    mag[1::2, 0, 0] = 2.0 ** -100          # clip 1, 3, ..: slab 0 dead before live slabs
    mag[:, 0, 3] = 0.0                     # an all-zero slab
    a = (a.view(B, RP, K // 128, 128) * mag).reshape(B * RP, K)
    a.view(B, RP, K)[:, Tp:] = 0
    w = torch.randn(N, K, generator=g) / K ** 0.5
    ref = _block_reference(a, w, torch.zeros(N), None, None, B, RP, Tp, 0)
    c, _, _ = rt.gemm_clip_h2p(a.cuda(), w, None, B, Tp, 0, None, None, a_planes=True, out_planes=True)
    c = c.cpu().view(B, RP, N)[:, :Tp].double()
    assert bool(torch.isfinite(c).all())
    scale = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
    assert ((c - ref).abs() / scale).max().item() < 4e-6
