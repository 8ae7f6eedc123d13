"""GPU parity of the device-resident glue against the reference's own f0 functions (golden fixture
glue_f0.npz, produced by rvc/f0/rmvpe.py, rvc/f0/f0.py and rvc/f0/gen.py themselves) and the oracle restatement.

Bars: the f0 chain is fp64 end to end like numpy: pitchf equal to 1e-6 relative (device pow/log may differ from glibc in the
last ulp before the float32 cast), coarse pitch bins identical; feature expansion / protect mix / int16 scaling bit-exact (on top of the blend's own 1e-5 bar when an index is used)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import glue_oracle, ivf_oracle, synth

pytestmark = pytest.mark.gpu


def _cases(d):
    return sorted({k.split("::")[0] for k in d})


def test_rmvpe_f0_chain_matches_reference_golden(gpu):
    import rvc_amd

    d = load_golden("glue_f0")
    for c in _cases(d):
        n, p_len, key = (int(v) for v in d[c + "::meta"])
        pitch, pitchf = rvc_amd.glue.rmvpe_f0(torch.from_numpy(d[c + "::salience"]).to(gpu), p_len, key, 0.03)
        pitch, pitchf = pitch[0].cpu().numpy(), pitchf[0].cpu().numpy()
        ref_p, ref_f = d[c + "::pitch"], d[c + "::pitchf"]
        assert pitch.shape == ref_p.shape and pitch.dtype == np.int64
        assert np.allclose(pitchf, ref_f, rtol=1e-6, atol=0), "%s: pitchf differs (max rel %.2e)" % (
            c, np.max(np.abs(pitchf - ref_f) / np.maximum(np.abs(ref_f), 1e-9)))
        assert np.array_equal(pitchf == 0, ref_f == 0), c
        assert np.array_equal(pitch, ref_p), "%s: %d coarse bins differ" % (c, int((pitch != ref_p).sum()))


def test_f0_post_only(gpu):
    import rvc_amd

    rng = np.random.default_rng(5)
    f0 = rng.uniform(40, 1300, 500)
    f0[rng.random(500) < 0.2] = 0.0
    for key in (-12, 0, 5):
        ref_c, ref_f = glue_oracle.post_process(f0.copy(), key)
        pitch, pitchf = rvc_amd.glue.f0_post(torch.from_numpy(f0).to(gpu), key)
        assert np.array_equal(pitch[0].cpu().numpy(), ref_c.astype(np.int64))
        assert np.allclose(pitchf[0].cpu().numpy(), ref_f.astype(np.float32), rtol=1e-6, atol=0)


def test_f0_chain_has_no_length_limit(gpu):
    """The reference computes f0 ONCE for the whole file (pipeline.py:260-266, 100 frames/s): a 6-minute file is 36 000
    frames.  Beyond 20 480 frames the sequential pass keeps its work arrays in global memory instead of LDS."""
    import rvc_amd

    rng = np.random.default_rng(9)
    n, p_len = 36000, 36011
    sal = (rng.random((n, 360), dtype=np.float32) * 0.02).astype(np.float32)
    centre = (150 + 100 * np.sin(np.arange(n) / 40.0)).astype(int)
    voiced = (np.arange(n) % 300) >= 45
    voiced[-200:] = False  # trailing unvoiced run (the interpolation's tail branch)
    for w_ in range(-4, 5):
        sal[np.arange(n)[voiced], centre[voiced] + w_] += np.float32(0.9 * np.exp(-0.5 * (w_ / 2.0) ** 2))
    ref_p, ref_f = glue_oracle.rmvpe_f0(sal, p_len, 3, 0.03)
    pitch, pitchf = rvc_amd.glue.rmvpe_f0(torch.from_numpy(sal).to(gpu), p_len, 3, 0.03)
    assert np.allclose(pitchf[0].cpu().numpy(), ref_f, rtol=1e-6, atol=0)
    assert np.array_equal(pitch[0].cpu().numpy(), ref_p)
    f0 = rng.uniform(40, 1300, 30000)
    f0[rng.random(30000) < 0.2] = 0.0
    ref_c, ref_f2 = glue_oracle.post_process(f0.copy(), -4)
    pitch2, pitchf2 = rvc_amd.glue.f0_post(torch.from_numpy(f0).to(gpu), -4)
    assert np.array_equal(pitch2[0].cpu().numpy(), ref_c.astype(np.int64))
    assert np.allclose(pitchf2[0].cpu().numpy(), ref_f2.astype(np.float32), rtol=1e-6, atol=0)


@pytest.mark.parametrize("use_index,protect,p_len", [(True, 0.33, 117), (True, 0.5, 120), (False, 0.2, 120), (False, 0.5, 101), (True, 0.0, 1)])
def test_retrieve_blend_expand_matches_pipeline_expression(use_index, protect, p_len, gpu):
    """search + blend + x2 + protect mix in one pass == the pipeline.py:118-159 expression evaluated with torch-CPU on the
    oracle's retrieval result."""
    import rvc_amd

    nq, d_ = 60, 256
    idx = synth.make_ivf(3000, d_, seed=11)
    feats = 0.5 * torch.randn(1, nq, d_, generator=torch.Generator().manual_seed(2))
    pitchf = synth.make_f0(1, 2 * nq)
    pitchf[0, 5] = 0.5  # 0 < pitchf < 1 -> protected too (pitchff[pitchf < 1] = protect)
    hip = rvc_amd.IVFFlatHIP.from_arrays(idx["centroids"], idx["list_offsets"], idx["ids"], idx["vecs"], device=gpu) if use_index else None
    got = rvc_amd.glue.retrieve_blend_expand(feats.to(gpu), hip, 0.75, pitchf.to(gpu), protect, p_len).cpu()
    blended = feats
    if use_index:
        npy = ivf_oracle.search_blend(idx, feats[0].numpy(), 0.75)
        blended = torch.from_numpy(npy).unsqueeze(0)
    ref = glue_oracle.expand_protect(blended, feats, pitchf, protect, p_len)
    assert got.shape == ref.shape
    if use_index:  # the blend itself carries the documented <= 1e-5 bar (fp32 weight normalisation order); the rest is exact
        assert (got - ref).abs().max() <= 2e-6, "max abs diff %.3e" % (got - ref).abs().max()
    else:
        assert torch.equal(got, ref), "max abs diff %.3e" % (got - ref).abs().max()


@pytest.mark.parametrize("skip_rows,protect", [(20, 0.33), (45, 0.5), (0, 0.2)])
def test_realtime_partial_blend_expand_protect(skip_rows, protect, gpu):
    """rtrvc.py:167-185 + 221-233: only feats[0][skip_head // 2:] is searched and blended; x2 and the protect mix cover all
    rows (feats0 = the un-blended clone)."""
    import rvc_amd

    nq, d_, p_len = 60, 256, 117
    idx = synth.make_ivf(3000, d_, seed=11)
    feats = 0.5 * torch.randn(1, nq, d_, generator=torch.Generator().manual_seed(3))
    pitchf = synth.make_f0(1, 2 * nq)
    hip = rvc_amd.IVFFlatHIP.from_arrays(idx["centroids"], idx["list_offsets"], idx["ids"], idx["vecs"], device=gpu)
    got = rvc_amd.glue.retrieve_blend_expand(feats.to(gpu), hip, 0.75, pitchf.to(gpu), protect, p_len, realtime_guard=True,
                                             skip_rows=skip_rows).cpu()
    blended = feats.clone()
    blended[0, skip_rows:] = torch.from_numpy(ivf_oracle.search_blend(idx, feats[0, skip_rows:].numpy(), 0.75))
    ref = glue_oracle.expand_protect(blended, feats, pitchf, protect, p_len)
    assert got.shape == ref.shape
    assert (got - ref).abs().max() <= 2e-6, "max abs diff %.3e" % (got - ref).abs().max()
    assert torch.equal(got[0, :2 * skip_rows], glue_oracle.expand_protect(feats, feats, pitchf, protect, p_len)[0, :2 * skip_rows])


def test_realtime_vc_block_loop(gpu):
    """RealtimeVC (the rtrvc.py RVC.infer mirror) over three blocks of a rolling window: the phone / pitch / pitchf it hands
    to net_g.infer are the reference's expressions (rtrvc.py:163, 189-191, 209-233) evaluated literally."""
    import rvc_amd

    d_, win = 256, 160
    idx = synth.make_ivf(3000, d_, seed=11)
    hip = rvc_amd.IVFFlatHIP.from_arrays(idx["centroids"], idx["list_offsets"], idx["ids"], idx["vecs"], device=gpu)
    seen = {}

    class Net:  # records what infer receives, returns a dummy block
        def infer(self, phone, lengths, sid, pitch=None, pitchf=None, skip_head=None, return_length=None, return_length2=None):
            seen.update(phone=phone.cpu(), lengths=lengths.cpu(), pitch=pitch.cpu(), pitchf=pitchf.cpu(), skip_head=skip_head,
                        return_length=return_length, return_length2=return_length2)
            n = (return_length2 or return_length) * 480
            return (0.4 * torch.sin(torch.arange(n, device=phone.device, dtype=torch.float32) * 0.013)).reshape(1, 1, n)

    rt = rvc_amd.RealtimeVC(Net(), index=hip, index_rate=0.75, device=gpu, tgt_sr=48000)
    n_samples, block16k, skip_head, ret_len = 160 * 118, 4000, 40, 25
    p_len = n_samples // win
    cp = np.zeros(1024, dtype=np.int64)
    cf = np.zeros(1024, dtype=np.float32)
    g = torch.Generator().manual_seed(5)
    for blk in range(3):
        feats = 0.5 * torch.randn(1, 59, d_, generator=g)
        m = p_len
        pitchf = synth.make_f0(1, m + 7)[0, blk:blk + m].contiguous()
        pitch = synth.make_pitch(pitchf[None])[0]
        wav = rt.infer(feats.to(gpu), n_samples, block16k, skip_head, ret_len, pitch=pitch.to(gpu), pitchf=pitchf.to(gpu), protect=0.33)
        assert wav.shape == (ret_len * 480,)
        # literal rtrvc.py:209-219
        shift = block16k // win
        cp[:-shift] = cp[shift:].copy()
        cf[:-shift] = cf[shift:].copy()
        cp[4 - m:] = pitch.numpy()[3:-1]
        cf[4 - m:] = pitchf.numpy()[3:-1]
        assert np.array_equal(seen["pitch"][0].numpy(), cp[-p_len:]), "pitch window"
        assert np.allclose(seen["pitchf"][0].numpy(), cf[-p_len:] * ret_len / ret_len, rtol=1e-6, atol=0), "pitchf window"  # (x * rl2) / rl on the device
        assert seen["skip_head"] == skip_head and seen["return_length"] == ret_len and seen["return_length2"] == ret_len
        assert int(seen["lengths"][0]) == p_len
        f2 = torch.cat((feats, feats[:, -1:]), 1)
        blended = f2.clone()
        blended[0, skip_head // 2:] = torch.from_numpy(ivf_oracle.search_blend(idx, f2[0, skip_head // 2:].numpy(), 0.75))
        ref = glue_oracle.expand_protect(blended, f2, pitchf[None], 0.33, p_len)
        assert seen["phone"].shape == ref.shape, (seen["phone"].shape, ref.shape)
        assert (seen["phone"] - ref).abs().max() <= 2e-6, "phone: max abs diff %.3e" % (seen["phone"] - ref).abs().max()
    with pytest.raises(RuntimeError, match="exactly p_len"):  # the reference's broadcast (rtrvc.py:229) fails for m != p_len as well
        rt.infer(feats.to(gpu), n_samples, block16k, skip_head, ret_len, pitch=pitch[:-3].to(gpu), pitchf=pitchf[:-3].to(gpu), protect=0.33)
    rt.set_formant(2.0)  # rtrvc.py:190-191, 218-219: the decoder is asked for ceil(return_length * 2^(2/12)) frames ...
    wav = rt.infer(feats.to(gpu), n_samples, block16k, skip_head, ret_len, pitch=pitch.to(gpu), pitchf=pitchf.to(gpu))
    # ... and its output is resampled from upp_res to tgt_sr / 100 samples per 10 ms (rtrvc.py:248-259) by the HIP polyphase kernel
    factor = 2 ** (2.0 / 12)
    upp_res = int(np.floor(factor * 48000 // 100))
    rl2_ = int(np.ceil(ret_len * factor))
    sig = (0.4 * torch.sin(torch.arange(rl2_ * 480, dtype=torch.float32) * 0.013)).numpy()[: ret_len * upp_res]
    want = glue_oracle.sinc_resample(sig, upp_res, 480)
    assert wav.shape == want.shape and np.abs(wav.cpu().numpy() - want).max() <= 2e-6
    cp[:-shift] = cp[shift:].copy()
    cf[:-shift] = cf[shift:].copy()
    cp[4 - m:] = pitch.numpy()[3:-1]
    cf[4 - m:] = pitchf.numpy()[3:-1]
    rl2 = int(np.ceil(ret_len * 2 ** (2.0 / 12)))
    assert seen["return_length2"] == rl2
    assert np.allclose(seen["pitchf"][0].numpy(), cf[-p_len:] * rl2 / ret_len, rtol=1e-6)


def test_sinc_resample_kernel_matches_the_oracle(gpu):
    """rvcmi_glue_resample_poly + the product's filter table against the independent numpy restatement (oracle unpinned vs
    torchaudio, see glue_oracle.sinc_resample): up and down ratios of the realtime formant shift, ragged lengths, batch rows."""
    import rvc_amd

    rng = np.random.default_rng(4)
    for orig, new, n in ((423, 400, 12345), (538, 480, 13450), (357, 400, 8000), (400, 400, 100), (3, 2, 17)):
        x = rng.standard_normal((2, n)).astype(np.float32)
        rs = rvc_amd.SincResample(orig, new, gpu)
        got = rs(torch.from_numpy(x).to(gpu)).cpu().numpy()
        for r in range(2):
            want = glue_oracle.sinc_resample(x[r], orig, new) if orig != new else x[r]
            assert got[r].shape == want.shape and np.abs(got[r] - want).max() <= 5e-6 * max(1.0, np.abs(want).max()), (orig, new)
    with pytest.raises(rvc_amd.RvcmiError):
        rvc_amd.SincResample(423, 400, gpu)(torch.zeros(10))


def test_scale_int16_range(gpu):
    import rvc_amd

    rng = np.random.default_rng(9)
    for amp in (0.3, 0.99, 2.5):
        a = (amp * rng.standard_normal(100_003)).astype(np.float32)
        ref = glue_oracle.scale_int16_range(a)
        got = rvc_amd.glue.scale_int16_range(torch.from_numpy(a.copy()).to(gpu)).cpu().numpy()
        assert np.array_equal(got, ref)


def test_glue_rejects_cpu_tensors():
    import rvc_amd

    with pytest.raises(rvc_amd.RvcmiError):
        rvc_amd.glue.f0_post(torch.zeros(4, dtype=torch.float64))
    with pytest.raises(rvc_amd.RvcmiError):
        rvc_amd.glue.retrieve_blend_expand(torch.zeros(1, 4, 8), None, 0.0)


@pytest.mark.parametrize("true_off", [0, 137, 400])
def test_sola_matches_gui_expression(true_off, gpu):
    """gui.py geometry at 40 kHz: zc 400, block 26 zc, cross-fade buffer 4 zc, search 1 zc, chunk 31 zc."""
    import rvc_amd

    zc, blk, Lb, Ls = 400, 26 * 400, 4 * 400, 400
    gen = torch.Generator().manual_seed(true_off)
    t = torch.arange(31 * zc, dtype=torch.float32)
    wav = 0.4 * torch.sin(2 * np.pi * 220.0 * t / 40000) + 0.2 * torch.sin(2 * np.pi * 523.0 * t / 40000) + 0.02 * torch.randn(31 * zc, generator=gen)
    buf = wav[true_off: true_off + Lb].clone() * 0.9 + 0.01 * torch.randn(Lb, generator=gen)  # the previous tail: best match at true_off
    fade_in = torch.sin(0.5 * np.pi * torch.linspace(0.0, 1.0, Lb)) ** 2  # gui.py:841-855
    fade_out = 1 - fade_in
    ref_out, ref_buf, ref_off = glue_oracle.sola(wav, buf, fade_in, fade_out, blk, Ls)
    assert ref_off == true_off
    dbuf = buf.to(gpu).clone()
    out, off = rvc_amd.glue.sola(wav.to(gpu), dbuf, fade_in.to(gpu), fade_out.to(gpu), blk, Ls, return_offset=True)
    assert int(off.item()) == ref_off
    assert torch.equal(out.cpu(), ref_out) and torch.equal(dbuf.cpu(), ref_buf)


@pytest.mark.parametrize("rate", [0.25, 0.0, 0.8])
def test_change_rms_matches_the_pipeline_expression(rate, gpu):
    """rms-mix (pipeline.py:26-46,351; WebUI default rms_mix_rate = 0.25): device result against the oracle restatement
    (torch's own F.interpolate / pow; the librosa frame RMS is restated and UNPINNED, see oracle/glue_oracle.py:frame_rms).
    Bar: 2e-6 relative (device powf vs torch's, fp64 vs float32 frame sums)."""
    import rvc_amd

    rng = np.random.default_rng(3)
    n1 = 16000 * 7 + 1234
    t = np.arange(n1) / 16000.0
    env = (0.05 + 0.9 * (np.sin(2 * np.pi * 0.3 * t) ** 2)).astype(np.float32)
    data1 = (env * rng.standard_normal(n1).astype(np.float32) * 0.3).astype(np.float32)
    data1[20000:36000] = 0.0  # a silent second: rms1 -> 0
    n2 = int(n1 * 3)  # 48 kHz output
    data2 = (0.2 * rng.standard_normal(n2)).astype(np.float32)
    data2[100000:180000] = 0.0  # silent output stretch: the 1e-6 floor of rms2
    ref = glue_oracle.change_rms(data1, 16000, data2.copy(), 48000, rate)
    d2 = torch.from_numpy(data2.copy()).to(gpu)
    out = rvc_amd.glue.change_rms(torch.from_numpy(data1).to(gpu), 16000, d2, 48000, rate)
    assert out.data_ptr() == d2.data_ptr()
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.allclose(got, ref, rtol=2e-6, atol=1e-9), "max rel %.2e" % np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-9))


def test_realtime_entry_keeps_rmvpe_f0_on_the_device(gpu):
    """``rvc_infer_hip`` (the rebound ``RVC.infer``, rtrvc.py:134-260) with f0method "rmvpe": the f0 window goes waveform -> mel -> network ->
    ``rvcmi_glue_rmvpe_f0`` on the device (what the estimator's host path -- ``_get_f0``, which fails this test if called -- computes, see the
    golden test of the decode chain above); a fractional key (formant slider) takes the reference's own method."""
    import types

    from rvc_amd import glue
    from rvc_amd.realtime import f0_extractor_frame, rvc_infer_hip

    seen, host_calls = {}, []

    class Net:
        def infer(self, phone, lengths, sid, pitch=None, pitchf=None, skip_head=None, return_length=None, return_length2=None):
            seen.update(pitch=pitch.cpu(), pitchf=pitchf.cpu())
            n = (return_length2 or return_length) * 480
            return torch.zeros(1, 1, n, device=phone.device)

    block16k, win = 4096, 160
    n_in = 160 * 118
    fake = synth.FakeRMVPE(gpu, 77)

    def host_f0(x, key, method="rmvpe"):
        host_calls.append(float(key))
        m = int(x.shape[0]) // win
        pf = synth.make_f0(1, m)[0]
        return synth.make_pitch(pf[None])[0].to(gpu), pf.to(gpu)

    me = types.SimpleNamespace(index=None, net_g=Net(), index_rate=0.0, device=gpu, if_f0=1, tgt_sr=48000, f0_up_key=2, formant_shift=0, window=win,
                               is_half=False, version="v2", hubert=synth.FakeHubert(768, 5), f0_gen=types.SimpleNamespace(rmvpe=fake, is_half=False, device=gpu),
                               _get_f0=host_f0)
    wav_in = torch.from_numpy(synth.make_audio16k(n_in, 3)).to(gpu)
    out = rvc_infer_hip(me, wav_in, block16k, 40, 25, "rmvpe", 1.0)
    assert out.shape == (25 * 480,) and host_calls == [] and fake.model.calls == 1
    n = f0_extractor_frame(block16k, "rmvpe", win)
    m = n // win
    with torch.no_grad():
        hid = fake._mel2hidden(fake.mel_extractor(wav_in[-n:].unsqueeze(0), center=True))
    pitch, pitchf = glue.rmvpe_f0(hid.squeeze(0).float(), m, 2, 0.03)
    p_len = n_in // win
    # the cache window the decoder saw ends with the estimator's frames 3 .. m - 2 (rtrvc.py:213-217)
    assert torch.equal(seen["pitch"][0, p_len - (m - 4):], pitch[0, 3:-1].cpu())
    assert torch.allclose(seen["pitchf"][0, p_len - (m - 4):], pitchf[0, 3:-1].cpu(), rtol=1e-6, atol=0)  # ((x * rl2) / rl on the device, rtrvc.py:218-219)
    assert float(pitchf.max()) > 0
    me.formant_shift = 0.5  # key 1.5: not the C ABI's integer -> the object's own _get_f0
    rvc_infer_hip(me, wav_in, block16k, 40, 25, "rmvpe", 1.0)
    assert host_calls == [1.5]
