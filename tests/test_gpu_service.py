"""GPU tests through the reference-shaped API: load(), embed_watermark, detect_watermark,
plug-in classes, and the batched pipeline."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, make_clip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def models():
    from aware_amd._lib import require_gpu
    require_gpu()
    from aware_amd.utils.models import load
    return load()


def test_readme_flow_mono_1s(models):
    """README flow at 16 kHz on the golden 1 s clip: clean path is exact (20/20 bits)."""
    from aware_amd.service import embed_watermark, detect_watermark
    from aware_amd.metrics import BER
    emb, det = models
    e = np.load(os.path.join(GOLDEN, "embed_1s.npz"))
    audio, bits = make_clip(1, 16000)
    wm_audio = embed_watermark(audio, 16000, bits, emb)
    assert wm_audio.shape == (int(e["out_len"]),) and wm_audio.dtype == np.float32
    got = detect_watermark(wm_audio, 16000, det)
    assert got.dtype == np.int32
    np.testing.assert_array_equal(got, bits)
    np.testing.assert_array_equal(got, e["det_bits"])
    assert BER()(bits, got) == 0.0
    # detecting the REFERENCE's own watermarked audio gives the reference's raw values
    raw = det.detect(e["out_sample"], 16000)
    np.testing.assert_allclose(raw, e["raw_marked"], atol=1e-4)
    # unmarked audio does not decode to the pattern
    raw_un = det.detect(audio, 16000)
    np.testing.assert_allclose(raw_un, e["raw_unmarked"], atol=5e-5)


def test_config1_44k_front_end(models):
    """BASELINE config 1: 3 s @44.1 kHz -> 16 kHz -> embed -> detect, BER 0."""
    from aware_amd.service import embed_watermark, detect_watermark
    from aware_amd.attacks import resample_poly_batch
    from aware_amd import runtime as rt
    emb, det = models
    c = np.load(os.path.join(GOLDEN, "config1_44k.npz"))
    rng = np.random.default_rng(0)
    a441 = (0.1 * rng.standard_normal(132300)).astype(np.float32)
    bits = rng.integers(0, 2, 20).astype(np.int32)
    a16 = resample_poly_batch(rt.Ragged.from_list([a441]), 16000, 44100).to_list()[0]
    wm_audio = embed_watermark(a16, 16000, bits, emb)
    assert wm_audio.shape[0] == int(c["out_len"])
    got = detect_watermark(wm_audio, 16000, det)
    np.testing.assert_array_equal(got, bits)
    np.testing.assert_array_equal(got, c["det_bits"])


def test_stereo_and_ragged_batch(models):
    from aware_amd.service import embed_watermark, detect_watermark, embed_watermark_batch, detect_watermark_batch
    emb, det = models
    l, bits = make_clip(11, 16000)
    r, _ = make_clip(12, 16000)
    st = np.column_stack([l, r])
    out = embed_watermark(st, 16000, bits, emb)
    assert out.shape == (15872, 2)
    np.testing.assert_array_equal(detect_watermark(out, 16000, det), bits)
    # each channel equals the mono call on that channel (two independent problems, embed.py:52-53)
    mono = embed_watermark(l, 16000, bits, emb)
    np.testing.assert_allclose(out[:, 0], mono, atol=1e-6)
    lens = [16000, 24000, 9000]
    clips, bl = zip(*[make_clip(30 + i, n) for i, n in enumerate(lens)])
    outs = embed_watermark_batch(list(clips), 16000, list(bl), emb)
    assert [o.shape[0] for o in outs] == [256 * (n // 256) for n in lens]
    got = detect_watermark_batch(outs, 16000, det)
    for g, b in zip(got, bl):
        np.testing.assert_array_equal(g, b)
    # no cross-talk between clips: the same clip inside a different ragged batch (same kernels,
    # per-clip segmented reductions) gives the identical waveform
    outs2 = embed_watermark_batch([clips[1], clips[2][:8000]], 16000, [bl[1], bl[2]], emb)
    np.testing.assert_array_equal(outs2[0], outs[1])
    # ragged batches run the generic kernels, a uniform batch the clip-aligned fused ones: the two
    # paths agree to rounding after a few iterations (400-step trajectories drift apart in fp32)
    from aware_amd import runtime as rt
    import torch
    short = []
    for lens_ in ([24000], [24000, 9000]):
        b_ = rt.Batch(lens_)
        s_ = emb.start_session(b_, 16000)
        wm_ = torch.tensor(np.stack([2 * bl[1] - 1] + ([2 * bl[2] - 1] if len(lens_) > 1 else [])), dtype=torch.float32, device="cuda")
        s_.begin(b_.pack([clips[1]] + ([clips[2]] if len(lens_) > 1 else [])), wm_)
        s_.iterate(5)
        short.append(b_.unpack_out(s_.finish(None))[0].cpu().numpy())
    # (NAdam's first steps are sign-like: a coefficient whose tiny gradient flips sign between the
    # two summation orders moves by 2*lr, so compare in the L2 sense)
    assert np.linalg.norm(short[0] - short[1]) / np.linalg.norm(short[0]) < 1e-3


def test_plugins_match_torch(models):
    from aware_amd.utils.audio import STFT, ISTFT, WaveformNormalizer, STFTDecomposer, STFTAssembler
    x = torch.from_numpy(make_clip(2, 20000)[0])
    xn = WaveformNormalizer()(x)
    ref = x / torch.max(torch.abs(x) + 1e-8)
    assert float((xn.cpu() - ref).abs().max()) == 0.0
    S = STFT(1024, 256, "hann", 1024)(xn)
    St = torch.stft(ref, 1024, 256, window=torch.hann_window(1024), center=True, return_complex=True)
    assert S.shape == St.shape
    assert float((S.cpu() - St).abs().max()) < 1e-5 * float(St.abs().max())
    mag, ph = STFTDecomposer()(S)
    y = ISTFT(1024, 256, "hann", 1024)(STFTAssembler()(mag, ph))
    yt = torch.istft(St, 1024, 256, window=torch.hann_window(1024), center=True)
    assert y.shape == yt.shape and float((y.cpu() - yt).abs().max()) < 2e-6
    with pytest.raises(ValueError, match="Invalid window type"):
        STFT(1024, 256, "blackman", 1024)
    Sh = STFT(1024, 256, "hamming", 1024)(xn)
    Sht = torch.stft(ref, 1024, 256, window=torch.hamming_window(1024), center=True, return_complex=True)
    assert float((Sh.cpu() - Sht).abs().max()) < 1e-5 * float(Sht.abs().max())


def test_detector_net_forward_shape_and_parity(models):
    from oracle import aware_oracle as O
    emb, det = models
    x = torch.from_numpy(np.stack([make_clip(40, 16000)[0], make_clip(41, 16000)[0]]))
    oe = O.Embedder()
    mag = torch.abs(O.stft(x / x.abs().amax(dim=1, keepdim=True)))
    mag[:, oe.nonband] = 0
    out = emb.detection_net(mag)
    assert out.shape == (2, 20, 1)
    np.testing.assert_allclose(out[:, :, 0].cpu().numpy(), oe.det.forward(mag).numpy(), atol=5e-5)


def test_pipeline_attack_chain_and_each(models):
    from aware_amd.pipeline import WatermarkPipeline, synthetic_clips
    from aware_amd.attacks import config3_attack_stack, PCMBitDepthConversion, LowPassFilter, Resample
    emb, det = models
    audio, bits = synthetic_clips(4, 1.0, 44100, first_seed=0)
    clean = WatermarkPipeline(emb, det, [], 16000).run(audio, bits, input_rate=44100)
    assert int(clean.bit_errors) == 0 and abs(clean.seconds - 4.0) < 1e-9
    np.testing.assert_array_equal(clean.bits.cpu().numpy(), bits.cpu().numpy())
    chain = WatermarkPipeline(emb, det, config3_attack_stack(), 16000, "chain").run(audio, bits, input_rate=44100)
    each = WatermarkPipeline(emb, det, [PCMBitDepthConversion(16), LowPassFilter(), Resample()], 16000, "each").run(
        audio, bits, input_rate=44100)
    assert set(each.per_attack_errors) == {"pcm_16", "low_pass", "resample_16000"}
    assert int(each.per_attack_errors["pcm_16"]) == 0
    # robustness is not asserted bit-exactly under attacks; BER must stay far below chance (50 %)
    assert int(chain.bit_errors) <= 0.25 * bits.numel()


def test_config5_ragged_clips_with_per_clip_chains(models):
    """BASELINE config 5 in miniature: mixed-length clips, a randomly drawn attack chain per clip,
    clips grouped by chain id."""
    import random
    from aware_amd import runtime as rt
    from aware_amd.pipeline import WatermarkPipeline
    from aware_amd.attacks import PCMBitDepthConversion, LowPassFilter, GaussianNoise, Resample, SampleSupression
    emb, det = models
    rng = np.random.default_rng(5)
    lens = [16000, 24000, 32000, 20000, 16000, 40000]
    pairs = [make_clip(60 + i, n) for i, n in enumerate(lens)]
    audio = rt.Ragged.from_list([p[0] for p in pairs])
    bits = torch.tensor(np.stack([p[1] for p in pairs]), dtype=torch.int32, device="cuda")
    chains = [[], [PCMBitDepthConversion(16)], [LowPassFilter(), GaussianNoise(30.0)], [Resample(), PCMBitDepthConversion(8)]]
    random.seed(3)
    chain_of_clip = [random.randrange(len(chains)) for _ in lens]
    res = WatermarkPipeline(emb, det, [], 16000).run(audio, bits, chains=chains, chain_of_clip=chain_of_clip)
    assert int(res.clean_bit_errors) == 0
    assert res.bits.shape == bits.shape
    # un-attacked and 16-bit PCM clips decode exactly; the rest stay far below chance
    for i, c in enumerate(chain_of_clip):
        wrong = int((res.bits[i] != bits[i]).sum())
        assert wrong == 0 if c in (0, 1) else wrong <= 6, (i, c, wrong)


def test_losses_and_windows_first_gradient(models):
    """Other registered losses (mse, hinge, sign) and the hamming window: first-iteration loss and
    gradient against torch autograd on the oracle."""
    from oracle import aware_oracle as O
    from aware_amd import runtime as rt
    from aware_amd.utils.audio import get_plan
    emb, det = models
    audio, bits = make_clip(21, 16000)
    wm = O.bits_to_bipolar(bits).astype(np.float32)
    plan = get_plan()
    dw = emb.detection_net.device_weights(plan)
    batch = rt.Batch([16000])
    for loss in ("mse", "hinge", "sign", "push_sigmoid", "ber"):
        sess = rt.EmbedSession(plan, dw, batch, loss=loss, use_graph=False)
        sess.begin(batch.pack([audio]), torch.from_numpy(wm[None]).cuda())
        g = sess.gradient().cpu()[:, :225].T
        oe = O.Embedder(loss=loss)
        a = torch.from_numpy(audio)[None]
        mag0, phase = oe.analyse(a)
        c0 = mag0[:, oe.band].clone().requires_grad_(True)
        l, _ = oe.forward_loss(c0, mag0, phase, torch.from_numpy(wm)[None])
        l.sum().backward()
        assert abs(float(sess.loss.cpu()[0]) - float(l)) < 2e-5
        ref = c0.grad[0]
        if float(ref.norm()) > 0:
            assert float((g - ref).norm() / ref.norm()) < 3e-3, loss
        else:
            assert float(g.abs().max()) == 0.0


def test_edge_cases(models):
    from aware_amd import runtime as rt
    from aware_amd.service import embed_watermark, detect_watermark
    emb, det = models
    with pytest.raises(ValueError):
        rt.Batch([512])                                     # torch.stft's reflect padding needs n > 512
    # shortest legal clip (3 frames) and a 10 s clip (T = 626) in one ragged batch
    a_short, b_short = make_clip(70, 513)
    a_long, b_long = make_clip(71, 160000)
    vals = det.detect_batch([a_short, a_long], 16000)
    assert vals.shape == (2, 20) and bool(torch.isfinite(vals).all())
    from oracle import aware_oracle as O
    ref = O.Embedder().detect_raw(a_long[None])[0].numpy()
    np.testing.assert_allclose(vals[1].cpu().numpy(), ref, atol=5e-5)
    # all-zero audio is finite end to end (the reference would stop it at the VAD gate)
    z = det.detect(np.zeros(16000, dtype=np.float32), 16000)
    assert np.isfinite(z).all()
    # 10 s clip through the generic (unfused) embed path, few iterations
    emb2_iters = emb.num_iterations
    try:
        emb.num_iterations = 3
        out = emb.embed(a_long, 16000, O.bits_to_bipolar(b_long))
        assert out.shape == (160000,) and np.isfinite(out).all() and abs(np.abs(out).max() - 1.0) < 1e-6
    finally:
        emb.num_iterations = emb2_iters


def test_full_size_config2_properties(models):
    """BASELINE config 2 at full size (64 x 3 s clips from 44.1 kHz): size-independent properties --
    every clip decodes exactly (BER 0), the best loss improved on the first iteration's for every
    clip, the watermark stays inside the +-6 dB box, outputs are unit-peak times the input maximum,
    and a second run is bit-identical (no atomics anywhere on the path)."""
    from aware_amd.pipeline import WatermarkPipeline, synthetic_clips
    emb, det = models
    audio, bits = synthetic_clips(64, 3.0, 44100, first_seed=1000)
    pipe = WatermarkPipeline(emb, det, [], 16000)
    r1 = pipe.run(audio, bits, input_rate=44100)
    assert int(r1.bit_errors) == 0 and abs(r1.seconds - 192.0) < 1e-9
    key = next(k for k in pipe._sessions if not isinstance(k[0], str))
    batch, sess = pipe._sessions[key]
    assert batch.B == 64 and batch.frames[0] == 188
    best = sess.best_loss.cpu().numpy()
    assert np.all(best < 0.65) and np.all(np.isfinite(best))
    lo, hi = sess.bounds
    bc = sess.best_coef
    assert bool(((bc >= lo) & (bc <= hi)).all())
    assert bool((bc[:, 225:] == 0).all())                     # row padding stays zero
    assert int(sess.step.cpu()[0]) == 400
    w1 = r1.watermarked.data.clone()
    r2 = pipe.run(audio, bits, input_rate=44100)
    assert torch.equal(w1, r2.watermarked.data)
    assert torch.equal(r1.values, r2.values)
    # detector margin as in the reference (|raw| ~ 0.28-0.30 on marked audio)
    assert float(r1.values.abs().min()) > 0.15


def test_full_size_config3_properties(models):
    """BASELINE config 3 at full size (256 x 3 s clips from 44.1 kHz, attack stack resample 16k<->44.1k + low-pass +
    Gaussian noise 20 dB + PCM 16): the clean read-out of every clip is exact, the attacked read-out stays far from
    chance, SNR of the watermark is finite and in the range the +-6 dB box allows, and a second run is bit-identical
    up to the detected bits (the noise attack is keyed per clip, so it repeats too)."""
    from aware_amd.pipeline import WatermarkPipeline, synthetic_clips
    from aware_amd.attacks import config3_attack_stack
    emb, det = models
    audio, bits = synthetic_clips(256, 3.0, 44100, first_seed=5000)
    pipe = WatermarkPipeline(emb, det, config3_attack_stack(), 16000, "chain")
    r1 = pipe.run(audio, bits, input_rate=44100, report_snr=True)
    assert abs(r1.seconds - 768.0) < 1e-9
    assert int(r1.clean_bit_errors) == 0
    assert int(r1.bit_errors) <= 0.05 * bits.numel()
    snr = r1.snr_db.cpu().numpy()
    assert snr.shape == (256,) and np.all(np.isfinite(snr)) and np.all(snr > 0.0) and np.all(snr < 40.0)
    r2 = pipe.run(audio, bits, input_rate=44100)
    assert torch.equal(r1.watermarked.data, r2.watermarked.data)
    assert torch.equal(r1.bits, r2.bits)


def test_full_size_config5_properties(models):
    """BASELINE config 5 at one GPU's share, bench.py's own batch (256 clips of seeded 1..10 s from 44.1 kHz, 400 iterations, a seeded chain
    of 1..3 attacks per clip out of {pcm16, resample, lowpass, bandstop, cut 10 %, gaussian 20 dB}, bench.py's own
    plan): size-independent properties -- every clip's clean read-out is exact, the attacked read-out stays far from
    chance overall and exact for the chains made of benign attacks only, every clip's optimiser ran 400 steps inside
    its box, output lengths follow 256*(T-1), and the embed is bit-identical on a second run (ragged path: no atomics)."""
    import random
    import bench
    from aware_amd.pipeline import WatermarkPipeline, synthetic_ragged_clips
    emb, det = models
    secs, chains = bench.config5_plan(256, 1)
    assert min(secs) == 1 and max(secs) == 10 and len(set(map(tuple, chains))) > 20
    order = sorted(range(256), key=lambda i: (-secs[i], i))
    secs, chains = [secs[i] for i in order], [chains[i] for i in order]
    audio, bits = synthetic_ragged_clips(secs, 44100, seeds=order)
    pipe = WatermarkPipeline(emb, det, [], 16000)
    random.seed(11)
    np.random.seed(11)
    r1 = pipe.run(audio, bits, input_rate=44100, chains_by_kind=(chains, bench.make_attack_of_kind))
    assert abs(r1.seconds - float(sum(secs))) < 1e-6
    assert int(r1.clean_bit_errors) == 0
    wrong = (r1.bits != bits).sum(dim=1).cpu().numpy()
    assert wrong.sum() <= 0.05 * bits.numel(), wrong
    benign = [i for i, c in enumerate(chains) if set(c) <= {"pcm", "resample", "lowpass"}]
    assert len(benign) >= 10 and int(wrong[benign].sum()) == 0
    key = next(k for k in pipe._sessions if not isinstance(k[0], str))
    batch, sess = pipe._sessions[key]
    assert batch.B == 256 and max(batch.frames) == 626 and min(batch.frames) == 63
    assert r1.watermarked.lengths == [256 * (t - 1) for t in batch.frames]
    assert int(sess.step.cpu()[0]) == 400
    best = sess.best_loss.cpu().numpy()
    assert np.all(np.isfinite(best)) and np.all(best < 0.7)
    lo, hi = sess.bounds
    bc = sess.best_coef
    assert bool(((bc >= lo) & (bc <= hi)).all())
    w1 = r1.watermarked.data.clone()
    random.seed(11)
    np.random.seed(11)
    r2 = pipe.run(audio, bits, input_rate=44100, chains_by_kind=(chains, bench.make_attack_of_kind))
    assert torch.equal(w1, r2.watermarked.data)
    assert torch.equal(r1.bits, r2.bits)                      # same host draws -> same attacked audio -> same bits


def test_degenerate_inputs_stay_finite(models):
    """Silence and the shortest legal clip go through a few optimiser iterations without NaN/Inf
    (zero variance in every InstanceNorm, zero gradients into NAdam, a single pooled frame)."""
    from aware_amd import runtime as rt
    emb, det = models
    lens = [16000, 513, 16000]
    clips = [np.zeros(16000, dtype=np.float32), make_clip(80, 513)[0], make_clip(81, 16000)[0]]
    batch = rt.Batch(lens)
    sess = emb.start_session(batch, 16000)
    wm = torch.ones((3, 20), device="cuda")
    sess.begin(batch.pack(clips), wm)
    sess.iterate(5)
    out = sess.finish(None)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())
    assert bool(torch.isfinite(sess.loss).all()) and bool(torch.isfinite(sess.coef).all())
    o = batch.unpack_out(out)
    assert float(o[0].abs().max()) == 0.0                 # silence stays silence
    assert [x.numel() for x in o] == [15872, 512, 15872]


def test_pipeline_snr_and_44k_back_end(models):
    """the imperceptibility metric on the device and the 16 kHz -> 44.1 kHz back end.
    SNR: the +-6 dB box around every band magnitude bounds the distortion, so the watermarked clip stays within a
    few dB of the host (the oracle's own embed of the 1 s golden clip gives the reference value);
    back end: polyphase 441/160 of the watermarked clip = scipy.signal.resample_poly on the same samples."""
    from scipy.signal import resample_poly
    from aware_amd import runtime as rt
    from aware_amd.pipeline import WatermarkPipeline
    from oracle import aware_oracle as O
    emb, det = models
    clips = [make_clip(s, n) for s, n in zip((1, 7), (16000, 20000))]
    audio = rt.Ragged(torch.from_numpy(np.concatenate([c[0] for c in clips])).cuda(), [16000, 20000])
    bits = torch.from_numpy(np.stack([c[1] for c in clips])).cuda()
    pipe = WatermarkPipeline(emb, det)
    res = pipe.run(audio, bits, report_snr=True, output_rate=44100)
    assert int(res.bit_errors) == 0
    snr = res.snr_db.cpu().numpy()
    wm = res.watermarked
    for i, (a, _) in enumerate(clips):
        w = wm.data[wm.offsets[i]: wm.offsets[i] + wm.lengths[i]].cpu().numpy()
        n = min(len(w), len(a))
        ref = 10 * np.log10(np.mean(w[:n].astype(np.float64) ** 2) / np.mean((w[:n].astype(np.float64) - a[:n]) ** 2))
        assert abs(snr[i] - ref) < 1e-6
        assert 0.0 < snr[i] < 40.0
        up = res.watermarked_out
        got = up.data[up.offsets[i]: up.offsets[i] + up.lengths[i]].cpu().numpy()
        want = resample_poly(w.astype(np.float32), 441, 160)
        assert got.shape == want.shape
        assert np.max(np.abs(got - want)) < 3e-6 * max(1.0, np.max(np.abs(want)))
    # the oracle's embed of the golden 1 s clip lands at the same SNR (both trajectories stay in the same box)
    e = np.load(os.path.join(GOLDEN, "embed_1s.npz"))
    ref_out = e["out_sample"]
    a = clips[0][0]
    n = min(len(ref_out), len(a))
    ref_snr = 10 * np.log10(np.mean(ref_out[:n].astype(np.float64) ** 2) / np.mean((ref_out[:n].astype(np.float64) - a[:n]) ** 2))
    assert abs(snr[0] - ref_snr) < 1.0, (snr[0], ref_snr)


def test_run_folder_harness(tmp_path):
    """The reference's harness loop over a folder (scripts/test.py:52-106) on WAV files: two 16 kHz clips (one PCM16, one
    float), one 44.1 kHz clip (front-end resampling), one file that is too short; BER per file clean and per attack."""
    from aware_amd.utils.models import load
    from aware_amd.utils.audio import io
    from aware_amd import attacks as A
    from aware_amd.pipeline import run_folder
    rng = np.random.default_rng(11)
    io.write_wav(tmp_path / "a.wav", (0.1 * rng.standard_normal(16000)).astype(np.float32), 16000)
    io.write_wav(tmp_path / "b.wav", (0.1 * rng.standard_normal(24000)).astype(np.float32), 16000, subtype="FLOAT")
    io.write_wav(tmp_path / "c.wav", (0.1 * rng.standard_normal(44100)).astype(np.float32), 44100, subtype="FLOAT")
    io.write_wav(tmp_path / "short.wav", np.zeros(300, dtype=np.float32), 16000)
    emb, det = load()
    rec = run_folder(tmp_path, emb, det, attacks=[A.PCMBitDepthConversion(16), A.LowPassFilter()], seed=3)
    assert sorted(rec["files"]) == ["a.wav", "b.wav", "c.wav"]
    assert [n for n, _ in rec["skipped"]] == ["short.wav"]
    assert rec["orig"] == [0.0, 0.0, 0.0]                                   # BER in percent, clean path: bit-exact
    assert rec["pcm_16"] == [0.0, 0.0, 0.0] and rec["low_pass"] == [0.0, 0.0, 0.0]
    assert all(np.isfinite(v) and v > 10.0 for v in rec["snr_db"])
