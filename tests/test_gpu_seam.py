"""The differentiable plug-in seam: the reference's loop written with its own object shapes -- ordered
plug-in lists, autograd, an optimiser step, a clamp (embedding/multibit_embedder.py:40-41,49-67,95-122) -- runs on
torch.autograd.Function wrappers whose forward AND backward are C-ABI calls, and lands on the same numbers as the fused
loop (aware_embed_iterate).  Plus: each backward against torch autograd on the oracle's restatement of the op, and the
push_extremes + L1 objective (EXTENSION) against autograd on the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import make_clip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from aware_amd._lib import require_gpu
    require_gpu()
    from aware_amd import runtime as rt
    from aware_amd.utils.models import load
    from oracle import aware_oracle as O
    emb, det = load()
    return rt, emb, det, O


@pytest.mark.parametrize("lengths", [[48000], [513], [16000, 48000, 23456, 777, 160000], [256 * 40, 256 * 40 + 255, 256 * 41 - 1]])
def test_stft_backward_any_length(env, lengths):
    """aware_stft_bwd for every input aware_stft accepts: clips of any length > 512 in a ragged batch (the reference's
    torch.stft(center=True) under autograd, utils/audio/stft.py:27-28), against torch autograd on the oracle's explicit
    restatement (reflect pad, unfold, window, rfft) with a random complex cotangent."""
    rt, emb, det, O = env
    from aware_amd.utils.audio import default_plan
    plan = default_plan()
    g = torch.Generator().manual_seed(sum(lengths))
    clips = [0.1 * torch.randn(n, generator=g) for n in lengths]
    batch = rt.Batch(lengths)
    G = torch.zeros((batch.total_frames, rt.FULL_STRIDE), dtype=torch.complex64)
    G[:, :513] = torch.complex(torch.randn(batch.total_frames, 513, generator=g), torch.randn(batch.total_frames, 513, generator=g))
    ga = rt.stft_bwd(plan, batch, G.cuda()).cpu()
    assert ga.shape[0] == sum(lengths)
    for i, c in enumerate(clips):
        x = c.clone().requires_grad_(True)
        S = O.stft(x[None])[0]                                                   # [513, T]
        Gi = G[batch.frame_offsets[i]: batch.frame_offsets[i + 1], :513].T
        (S.real * Gi.real + S.imag * Gi.imag).sum().backward()
        mine = ga[batch.in_offsets[i]: batch.in_offsets[i] + lengths[i]]
        rel = float((mine - x.grad).norm() / x.grad.norm())
        print(f"clip {i} (n = {lengths[i]}, T = {batch.frames[i]}): rel L2 {rel:.2e}")
        assert rel < 2e-6, (i, rel)
    # and through the plug-in object (single clip), as a torch op
    from aware_amd.utils.audio import STFT
    xd = clips[0].cuda().requires_grad_(True)
    S = STFT(1024, 256, "hann", 1024)(xd)
    Gi = G[: batch.frames[0], :513].T.cuda()
    (S.real * Gi.real + S.imag * Gi.imag).sum().backward()
    assert float((xd.grad.cpu() - ga[: lengths[0]]).abs().max()) == 0.0


def test_transform_backward_matches_autograd(env):
    """STFT / ISTFT / normaliser / decomposer / assembler: gradients through the C ABI vs torch autograd on the oracle's
    explicit restatement (reflect pad, unfold, rfft, overlap-add), random cotangents."""
    rt, emb, det, O = env
    from aware_amd.utils.audio import STFT, ISTFT, WaveformNormalizer, STFTDecomposer, STFTAssembler
    g = torch.Generator().manual_seed(3)
    n = 256 * 40
    x = (0.1 * torch.randn(n, generator=g)).requires_grad_(True)
    xd = x.detach().cuda().requires_grad_(True)
    # normalise -> STFT -> (|.|, angle) -> assemble -> ISTFT -> normalise, random cotangent
    pre = [WaveformNormalizer(), STFT(1024, 256, "hann", 1024), STFTDecomposer()]
    post = [STFTAssembler(), ISTFT(1024, 256, "hann", 1024), WaveformNormalizer()]
    v = xd
    for p in pre:
        v = p(v)
    mag, ph = v
    y = post[2](post[1](post[0](mag * 1.5, ph)))
    w = torch.randn(y.shape, generator=g)
    (y * w.cuda()).sum().backward()
    # oracle chain under autograd
    xn = x / torch.amax(torch.abs(x) + 1e-8)
    S = O.stft(xn[None])[0]
    m0, p0 = torch.abs(S), torch.angle(S)
    yr = O.istft(((m0 * 1.5) * torch.exp(1j * p0))[None])[0]
    yr = yr / torch.amax(torch.abs(yr) + 1e-8)
    (yr * w).sum().backward()
    assert y.shape == yr.shape
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), atol=3e-6)
    rel = float((xd.grad.cpu() - x.grad).norm() / x.grad.norm())
    print("chain gradient rel L2 vs autograd:", rel)
    assert rel < 2e-5, rel


def test_detector_net_backward_matches_autograd(env):
    rt, emb, det, O = env
    g = torch.Generator().manual_seed(5)
    B, T = 2, 63
    mag = torch.rand(B, 513, T, generator=g) * 3
    oe = O.Embedder()
    mag[:, oe.nonband] = 0
    cot = torch.randn(B, 20, generator=g)
    md = mag.cuda().requires_grad_(True)
    out = emb.detection_net(md)
    assert out.shape == (B, 20, 1)
    (out[:, :, 0] * cot.cuda()).sum().backward()
    mr = mag.clone().requires_grad_(True)
    ref = oe.det.forward(mr)
    (ref * cot).sum().backward()
    np.testing.assert_allclose(out[:, :, 0].detach().cpu().numpy(), ref.detach().numpy(), atol=5e-5)
    gd, gr = md.grad.cpu()[:, oe.band], mr.grad[:, oe.band]
    rel = float((gd - gr).norm() / gr.norm())
    print("detector backward rel L2 vs autograd:", rel)
    assert rel < 1e-4, rel
    assert float(md.grad.cpu()[:, oe.nonband].abs().max()) == 0.0       # out-of-band bins never reach the network


def test_reference_shaped_loop_matches_fused_loop(env):
    """Five iterations of AWAREEmbedder._optimize written as the reference writes it (plug-in lists + loss.backward() +
    optimiser step + clamp) on the plug-in seam, against aware_embed_iterate on the same clip: first-iteration loss and
    gradient to f32 rounding, per-step losses within the measured few-step drift, coefficients after five steps equal
    for all but the handful whose tiny gradient changes sign with rounding (NAdam's first steps move by lr * sign(g))."""
    rt, emb, det, O = env
    from aware_amd.utils.audio import get_plan
    from aware_amd.embedding.losses import get_loss_fn
    audio, bits = make_clip(1, 16000)
    target = torch.from_numpy(O.bits_to_bipolar(bits).astype(np.float32)).cuda()
    pre, post = emb.audio_preprocess_pipeline, emb.audio_postprocess_pipeline
    x = torch.from_numpy(audio).cuda()
    v = x
    for p in pre:
        v = p(v)
    magnitude, phase = v                                             # [513, T]
    fi, nfi = emb._get_embedding_frequency_indices(16000, 1024)
    fi_t, nfi_t = torch.from_numpy(fi).cuda(), torch.from_numpy(nfi).cuda()
    c0 = magnitude[fi_t].flatten().detach().clone()
    delta = c0 * 10 ** (-emb.tolerance_db / 20)
    lo, hi = torch.clamp(c0 - delta, min=0), c0 + delta
    coeffs = c0.clone().requires_grad_(True)
    opt = rt.NAdamClamp(coeffs.data, lr=0.1)
    loss_fn = get_loss_fn("push_extremes")
    losses, grad1 = [], None
    for it in range(5):
        coeffs.grad = None
        wm = magnitude.detach().clone()
        wm[fi_t] = coeffs.reshape(len(fi), -1)
        d = (wm, phase.detach())                                     # _recompute_watermarked_magnitude :49-67
        for p in post:
            d = p(*d) if isinstance(d, tuple) else p(d)
        for p in pre:
            d = p(*d) if isinstance(d, tuple) else p(d)
        wmag = d[0].clone()
        wmag[nfi_t] = 0.0
        pred = emb.detection_net(wmag.unsqueeze(0)).squeeze()
        loss = loss_fn(pred, target)
        loss.backward()
        if it == 0:
            grad1 = coeffs.grad.detach().clone()
        opt.step(coeffs.grad, lo, hi)                                # optimizer.step() + torch.clamp(coeffs, lo, hi)
        losses.append(float(loss))
    # the fused loop
    plan = get_plan()
    batch = rt.Batch([16000])
    sess = emb.start_session(batch, 16000)
    sess.begin(batch.pack([audio]), target[None])
    gf = sess.gradient()[:, :225].T.flatten()
    fused = []
    for it in range(5):
        sess.iterate(1)
        fused.append(float(sess.loss.cpu()[0]))
    assert abs(losses[0] - fused[0]) < 5e-6, (losses[0], fused[0])
    rel = float((grad1 - gf).norm() / gf.norm())
    print("first gradient, plug-in seam vs fused loop, rel L2:", rel, "| losses", losses, fused)
    assert rel < 5e-5, rel
    assert np.max(np.abs(np.asarray(losses) - np.asarray(fused))) < 1e-3
    cf = sess.coef[:, :225].T.flatten()
    frac = float(((coeffs.detach() - cf).abs() <= 1e-3 * (1 + cf.abs())).float().mean())
    print("coefficients equal after 5 steps:", frac)
    assert frac > 0.995
    assert bool(((coeffs.detach() >= lo) & (coeffs.detach() <= hi)).all())


def test_push_extremes_l1_extension(env):
    """EXTENSION (BASELINE config 3 'BER + L1 loss'; not in the reference -- parity unpinned, the oracle is the spec):
    loss id 6 = push_extremes + l1_weight * mean|c - c0|.  After a few plain steps (so that c != c0) the loss and the
    gradient of the L1 objective against torch autograd on the oracle; with weight 0 it is push_extremes exactly; a full
    L1 run stays closer to the host signal than the plain one and still decodes."""
    rt, emb, det, O = env
    from aware_amd.utils.audio import get_plan
    plan = get_plan()
    dw = emb.detection_net.device_weights(plan)
    audio, bits = make_clip(31, 16000)
    wm = O.bits_to_bipolar(bits).astype(np.float32)
    batch = rt.Batch([16000])
    lam = 0.5
    s1 = rt.EmbedSession(plan, dw, batch, loss="push_extremes_l1", l1_weight=lam, use_graph=False, num_iterations=60)
    s1.begin(batch.pack([audio]), torch.from_numpy(wm[None]).cuda())
    s1.iterate(3)
    g = s1.gradient().cpu()[:, :225].T
    l_hip = float(s1.loss.cpu()[0])
    c_now = s1.coef.cpu()[:, :225].T.contiguous()
    oe = O.Embedder(loss="push_extremes_l1", l1_weight=lam)
    mag0, phase = oe.analyse(torch.from_numpy(audio)[None])
    c = c_now[None].clone().requires_grad_(True)
    l, _ = oe.forward_loss(c, mag0, phase, torch.from_numpy(wm)[None])
    l.sum().backward()
    assert abs(l_hip - float(l)) < 2e-5, (l_hip, float(l))
    rel = float((g - c.grad[0]).norm() / c.grad[0].norm())
    print("L1 objective: gradient rel L2 vs autograd on the oracle", rel)
    assert rel < 2e-4, rel
    # weight 0 == push_extremes
    s0 = rt.EmbedSession(plan, dw, batch, loss="push_extremes_l1", l1_weight=0.0, use_graph=False)
    sp = rt.EmbedSession(plan, dw, batch, loss="push_extremes", use_graph=False)
    for s in (s0, sp):
        s.begin(batch.pack([audio]), torch.from_numpy(wm[None]).cuda())
        s.iterate(4)
    assert torch.equal(s0.coef, sp.coef) and torch.equal(s0.loss, sp.loss)
    # full runs: the L1 penalty keeps the coefficients closer to the original ones
    outs = {}
    for name, kw in (("plain", dict(loss="push_extremes")), ("l1", dict(loss="push_extremes_l1", l1_weight=lam))):
        s = rt.EmbedSession(plan, dw, batch, use_graph=True, **kw)
        s.begin(batch.pack([audio]), torch.from_numpy(wm[None]).cuda())
        c0 = s.coef.clone()
        s.iterate(400)
        dist = float((s.best_coef - c0).abs().mean())
        out = s.finish(torch.tensor([float(audio.max())], device="cuda"))
        vals = rt.detect(plan, dw, rt.Batch([out.numel()]), out).cpu().numpy()[0]
        outs[name] = (dist, vals)
    print("mean |c - c0|: plain", outs["plain"][0], "with L1", outs["l1"][0])
    assert outs["l1"][0] < outs["plain"][0]
    np.testing.assert_array_equal(O.decode_bits(outs["plain"][1]), bits)
    # staged DSP kernels do not carry the L1 term: refused, not ignored
    with pytest.raises(Exception):
        rt.EmbedSession(plan, dw, batch, loss="push_extremes_l1", l1_weight=lam, dsp_path="staged")


def test_detector_weight_gradients_extension(env):
    """EXTENSION (detector training; no reference counterpart, parity unpinned -- the oracle's Detector under torch
    autograd is the specification): dL/dW and dL/db of every conv block and dL/dmag for a random upstream gradient, on a
    ragged batch; then one DetectorTrainer step changes the device weights consistently with the host copy."""
    rt, emb, det, O = env
    from aware_amd.utils.audio import get_plan
    plan = get_plan()
    dw = emb.detection_net.device_weights(plan)
    g = torch.Generator().manual_seed(9)
    lengths = [16000, 24000]
    clips = [make_clip(300 + i, n)[0] for i, n in enumerate(lengths)]
    batch = rt.Batch(lengths)
    x = batch.pack(clips)
    mag, _ = rt.stft_band(plan, batch, x, normalize=True)
    cot = torch.randn(2, 20, generator=g)
    vals, gmag, gw, gb = rt.detector_weight_gradients(plan, dw, batch, mag, cot.cuda())
    torch.cuda.synchronize()
    od = O.Detector()
    ws = [w.clone().requires_grad_(True) for w in od.ws]
    bs = [b.clone().requires_grad_(True) for b in od.bs]
    od.ws, od.bs = ws, bs
    oe = O.Embedder()
    total = 0.0
    for i, c in enumerate(clips):
        a = torch.from_numpy(c)[None]
        m = torch.abs(O.stft(a / torch.amax(torch.abs(a) + 1e-8))).clone()
        m[:, oe.nonband] = 0.0
        out = od.forward(m)
        np.testing.assert_allclose(vals[i].cpu().numpy(), out[0].detach().numpy(), atol=5e-5)
        total = total + (out[0] * cot[i]).sum()
    total.backward()
    for l in range(4):
        ref = ws[l].grad
        rel = float((gw[l].cpu() - ref).norm() / ref.norm())
        print(f"layer {l}: dL/dW rel L2 vs autograd {rel:.2e}; |dL/db| max {float(gb[l].abs().max()):.1e} (autograd {float(bs[l].grad.abs().max()):.1e})")
        assert rel < 2e-4, (l, rel)
        # the InstanceNorm behind every convolution removes per-channel constants: bias gradients vanish up to rounding
        assert float(gb[l].abs().max()) < 1e-4 * float(ref.abs().max()) * ref.shape[1]
    # Training steps, everything on the device (one forward + backward with the loss inside, Adam on the flat bucket, the packed
    # images rebuilt by kernels): against torch autograd + torch.optim.Adam on the oracle's detector with the same objective
    # (mean over clips of mse - 0.1 mean|raw|, losses.py:38-42) -- losses of three steps and the weights after them.
    from aware_amd.training import DetectorTrainer
    from aware_amd.utils.models import load
    emb2, det2 = load()
    w_before = [w.copy() for w in det2.detection_net.weights]
    bits = torch.randint(0, 2, (2, 20), generator=g).to(torch.int32).cuda()
    tr = DetectorTrainer(det2, lr=1e-3)
    mine = [tr.step(rt.Ragged(x, lengths), bits)[0] for _ in range(3)]
    od2 = O.Detector()
    pw = [w.clone().requires_grad_(True) for w in od2.ws]
    pb = [b_.clone().requires_grad_(True) for b_ in od2.bs]
    od2.ws, od2.bs = pw, pb
    adam = torch.optim.Adam(pw + pb, lr=1e-3)
    tg = (2 * bits.cpu().float() - 1)
    ref_losses = []
    for _ in range(3):
        adam.zero_grad()
        per_clip = []
        for i, c in enumerate(clips):
            a = torch.from_numpy(c)[None]
            m = torch.abs(O.stft(a / torch.amax(torch.abs(a) + 1e-8))).clone()
            m[:, oe.nonband] = 0.0
            p = od2.forward(m)[0]
            per_clip.append(((p - tg[i]) ** 2).mean() - 0.1 * p.abs().mean())
        loss = torch.stack(per_clip).mean()
        loss.backward()
        adam.step()
        ref_losses.append(float(loss))
    print("trainer losses", mine, "oracle", ref_losses)
    np.testing.assert_allclose(mine, ref_losses, atol=2e-5)
    assert mine[2] < mine[0]                                            # Adam steps on the same batch reduce its loss
    tr.sync_host()
    for l in range(4):
        rel = float(np.linalg.norm(det2.detection_net.weights[l] - pw[l].detach().numpy()) / np.linalg.norm(pw[l].detach().numpy()))
        moved = float(np.abs(det2.detection_net.weights[l] - w_before[l]).max())
        print(f"layer {l}: weights after 3 steps rel L2 vs oracle {rel:.2e}, moved by up to {moved:.1e}")
        assert rel < 2e-5 and moved > 1e-4
    v = det2.detect_batch(clips, 16000)
    assert bool(torch.isfinite(v).all())
    # the refreshed device images (f32, transposed, bf16x3 and f16x2 packs) are those of the new weights: a detector built
    # from the synced host copy gives the same outputs
    from aware_amd.runtime import DetectorWeights
    fresh = DetectorWeights(plan, det2.detection_net.mel_basis, det2.detection_net.weights, det2.detection_net.biases)
    np.testing.assert_allclose(v.cpu().numpy(), rt.detect(plan, fresh, batch, x).cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(v.cpu().numpy(), rt.detect(plan, det2.detection_net.device_weights(plan), batch, x).cpu().numpy(), atol=1e-6)
