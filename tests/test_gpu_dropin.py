"""The literal drop-in entry points executed on the GPU: ``rvc_amd.get_synthesizer`` /
``accelerate_synthesizer`` / ``from_reference`` on a synthesizer built by an RVC-shaped loader, and ``rvc_amd.install()``
rebinding ``rvc.synthesizer.get_synthesizer`` and ``faiss`` behind unmodified callers.  The RVC tree used here is the
compute-free skeleton of tests/skeleton (every forward raises), so any waveform that comes out was made by the HIP path.
Expected values: the golden fixtures produced by the REAL reference's ``net_g.infer`` (oracle/make_golden.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, rms
from oracle import ivf_oracle, nsf_oracle, synth
from oracle.front_oracle import FrontConfig

pytestmark = pytest.mark.gpu
SKEL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "skeleton")
_OURS = ("rvc", "infer", "faiss")


@pytest.fixture()
def rvc_tree():
    """Put the skeleton checkout on sys.path (as if the process ran inside an RVC checkout) and clean up afterwards."""
    import rvc_amd

    def purge():
        for m in [m for m in sys.modules if m.split(".")[0] in _OURS]:
            del sys.modules[m]

    purge()
    sys.path.insert(0, SKEL)
    yield SKEL
    rvc_amd.uninstall()
    sys.path.remove(SKEL)
    purge()


def make_cpt(seed=1234, f0=1):
    from oracle.nsf_oracle import GenConfig

    cfg, fcfg = nsf_oracle.CONFIGS["v2_48k"], FrontConfig()
    if not f0:
        cfg = GenConfig(**{**vars(cfg), "use_f0": False})
    wd, wf = synth.make_dec_weights(cfg, seed), synth.make_front_weights(fcfg, seed)
    if not f0:
        wf = {k: v for k, v in wf.items() if k != "enc_p.emb_pitch.weight"}
    weight = dict(wf)
    weight.update({"dec." + k: v for k, v in wd.items()})
    config = [1025, 32, 192, 192, 768, 2, 6, 3, 0, "1", cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes, cfg.upsample_rates,
              cfg.upsample_initial_channel, cfg.upsample_kernel_sizes, 109, cfg.gin_channels, cfg.sr]
    return dict(weight=weight, config=config, f0=f0, version="v2", info="synthetic", sr="48k")


def golden_args(d, gpu, dtype=torch.float32):
    T = d["phone"].shape[1]
    a = (torch.from_numpy(d["phone"]).to(gpu, dtype), torch.tensor([T], device=gpu), torch.from_numpy(d["sid"]).to(gpu),
         torch.from_numpy(d["pitch"]).to(gpu), torch.from_numpy(d["pitchf"]).to(gpu, dtype))
    kw = dict(noise_zp=torch.from_numpy(d["noise_zp"]).to(gpu), noise_dec=torch.from_numpy(d["noise_dec"]).to(gpu))
    return a, kw


def test_get_synthesizer_returns_a_net_whose_infer_and_dec_run_on_hip(rvc_tree, gpu):
    import rvc.layers.nsf as ref_nsf

    import rvc_amd

    d = load_golden("infer_full_v2_48k_T40")
    net_g, cpt = rvc_amd.get_synthesizer(make_cpt(int(d["seed"])), gpu)
    assert cpt["config"][-3] == 109  # the loader's in-place config fix-up survived (rvc/synthesizer.py:11)
    # type identity the reference's own infer dispatches on, and the HIP class, at once
    assert isinstance(net_g.dec, ref_nsf.NSFGenerator) and isinstance(net_g.dec, rvc_amd.NSFGeneratorHIP)
    a, kw = golden_args(d, gpu)
    out = net_g.infer(*a, **kw)
    assert out.shape == d["out"].shape and out.dtype == torch.float32
    e = rms(out.cpu(), d["out"])
    assert e <= 1e-3, "net_g.infer through the drop-in: RMS %.3e vs the reference" % e
    # the module call with the extra keyword (the narrowed reference __call__ must not be in the way) and without it
    g = net_g.emb_g(a[2]).unsqueeze(-1)
    z = net_g._rvcmi_front(a[0], a[3], a[1], g, 0, noise=kw["noise_zp"])
    o2 = net_g.dec(z, a[4], g=g, noise=kw["noise_dec"])
    assert torch.equal(o2, out)
    torch.manual_seed(7)
    o3 = net_g.dec(z, a[4], g=g)
    torch.manual_seed(7)
    assert torch.equal(o3, net_g.dec(z, a[4], g)) and not torch.equal(o3, out)
    # seeded RNG path of the whole infer: reproducible, and it advances the generator
    torch.manual_seed(114514)
    r1 = net_g.infer(*a)
    torch.manual_seed(114514)
    r2 = net_g.infer(*a)
    assert torch.equal(r1, r2) and not torch.equal(r1, net_g.infer(*a))
    # realtime arguments (skip_head / return_length / return_length2) through the same entry
    drt = load_golden("infer_full_v2_48k_rt")
    art, kwrt = golden_args(drt, gpu)
    ort = net_g.infer(*art, int(drt["skip_head"]), int(drt["return_length"]), int(drt["return_length2"]), **kwrt)
    assert rms(ort.cpu(), drt["out"]) <= 1e-3
    # f0 model called without pitchf: the reference's error, not a crash inside the kernels
    with pytest.raises(KeyError, match="unknown dec type"):
        net_g.infer(a[0], a[1], a[2], a[3], None)


def test_half_after_the_swap_and_workspace_growth(rvc_tree, gpu):
    """``net_g.half()`` (infer/modules/vc/modules.py:94-95) after the swap, and a clip longer than the default max_T = 256
    (the handle is re-created with a larger workspace on first use)."""
    import rvc_amd

    d = load_golden("infer_full_v2_48k_T40")
    net_g, _ = rvc_amd.get_synthesizer(make_cpt(int(d["seed"])), gpu)
    net_g = net_g.half()
    assert net_g.emb_g.weight.dtype == torch.float16
    a, kw = golden_args(d, gpu, torch.float16)
    out = net_g.infer(*a, **kw)
    assert out.dtype == torch.float16 and torch.isfinite(out).all()
    assert rms(out.float().cpu(), d["out"]) <= 2e-3  # fp16 phone / g / pitchf / output rounding on top of the 1e-3 class
    net_g = net_g.float()
    T = 300
    cfg = nsf_oracle.CONFIGS["v2_48k"]
    phone, pitchf = synth.make_phone(1, T, 768, 5), synth.make_f0(1, T)
    pitch = synth.make_pitch(pitchf)
    nz_zp = torch.randn(1, 192, T, generator=torch.Generator().manual_seed(3))
    nz_dec = nsf_oracle.reference_noise(1, T, cfg.upp, 4)
    ws0 = net_g.dec.workspace_bytes
    out = net_g.infer(phone.to(gpu), torch.tensor([T], device=gpu), torch.tensor([3], device=gpu), pitch.to(gpu), pitchf.to(gpu),
                      noise_zp=nz_zp.to(gpu), noise_dec=nz_dec.to(gpu))
    assert net_g.dec.workspace_bytes > ws0 and out.shape == (1, 1, T * cfg.upp)
    from oracle import front_oracle

    fcfg = FrontConfig()
    wd, wf = synth.make_dec_weights(cfg, int(d["seed"])), synth.make_front_weights(fcfg, int(d["seed"]))
    with torch.no_grad():
        z, m1, g = front_oracle.infer_front(fcfg, wf, phone, pitch, torch.tensor([T]), torch.tensor([3]), nz_zp)
        ref = nsf_oracle.generator_forward(cfg, wd, z * m1, pitchf, g, nz_dec)
    assert rms(out.cpu(), ref) <= 1e-3


def test_no_f0_model_through_the_loader(rvc_tree, gpu):
    import rvc.layers.generators as ref_gen

    import rvc_amd

    net_g, _ = rvc_amd.get_synthesizer(make_cpt(77, f0=0), gpu)
    assert isinstance(net_g.dec, ref_gen.Generator) and isinstance(net_g.dec, rvc_amd.GeneratorHIP)
    T = 24
    phone = synth.make_phone(1, T, 768, 9).to(gpu)
    nz = torch.randn(1, 192, T, generator=torch.Generator().manual_seed(1)).to(gpu)
    out = net_g.infer(phone, torch.tensor([T], device=gpu), torch.tensor([0], device=gpu), noise_zp=nz)
    cfg = nsf_oracle.CONFIGS["v2_48k"]
    assert out.shape == (1, 1, T * cfg.upp) and torch.isfinite(out).all()
    from oracle import front_oracle
    from oracle.nsf_oracle import GenConfig

    fcfg = FrontConfig(use_f0=False)
    cfg0 = GenConfig(**{**vars(cfg), "use_f0": False})
    wd, wf = synth.make_dec_weights(cfg0, 77), synth.make_front_weights(FrontConfig(), 77)
    with torch.no_grad():
        z, m1, g = front_oracle.infer_front(fcfg, wf, phone.cpu(), None, torch.tensor([T]), torch.tensor([0]), nz.cpu())
        ref = nsf_oracle.generator_forward(cfg0, wd, z * m1, None, g, None)
    assert rms(out.cpu(), ref) <= 1e-3


def test_install_rebinds_loader_and_faiss_behind_unmodified_callers(rvc_tree, gpu, tmp_path):
    """``rvc_amd.install()``: a caller module that bound ``load_synthesizer`` at import time BEFORE install (the WebUI's VC
    class does), one imported AFTER it, and the module-level ``import faiss`` of the conversion pipeline."""
    import infer.modules.vc.modules as vc_modules  # before install: holds the original loader functions
    import rvc.synthesizer as rs

    import rvc_amd

    orig_get = rs.get_synthesizer
    d = load_golden("infer_full_v2_48k_T40")
    pth = str(tmp_path / "model.pth")
    torch.save(make_cpt(int(d["seed"])), pth)
    rvc_amd.install(device=gpu)
    rvc_amd.install(device=gpu)  # idempotent
    assert rs.get_synthesizer is not orig_get and vc_modules.get_synthesizer is rs.get_synthesizer
    vc = vc_modules.VC(gpu, is_half=True)  # the reference's get_vc: load_synthesizer(...) then net_g.half()
    net_g = vc.get_vc(pth)
    assert isinstance(net_g.dec, rvc_amd.NSFGeneratorHIP)
    a, kw = golden_args(d, gpu, torch.float16)
    assert rms(net_g.infer(*a, **kw).float().cpu(), d["out"]) <= 2e-3
    # retrieval: an index file on disk, read through `faiss.read_index` inside the (unmodified) pipeline module
    idx = synth.make_ivf(3000, 768, seed=11)
    rvc_amd.write_index(rvc_amd.IVFFlatHIP.from_arrays(idx["centroids"], idx["list_offsets"], idx["ids"], idx["vecs"], device=gpu),
                        str(tmp_path / "added.index"))
    import infer.modules.vc.pipeline as pl  # after install: `import faiss` resolves to the shim even though faiss is absent

    index, big_npy = pl.load_index(str(tmp_path / "added.index"))
    assert isinstance(index, rvc_amd.IVFFlatHIP) and index.ntotal == 3000 and np.array_equal(big_npy, idx["xb"])
    q = np.random.default_rng(5).standard_normal((50, 768), dtype=np.float32)
    got, ix = pl.blend(index, big_npy, q, 0.75)
    Dr, Ir = ivf_oracle.search(idx, q, 8)
    assert np.array_equal(ix, Ir)
    assert np.allclose(got, ivf_oracle.blend(q, Dr, Ir, idx["xb"], 0.75), rtol=1e-5, atol=1e-6)
    import infer.lib.rtrvc as rt

    r = rt.RVC(pth, str(tmp_path / "added.index"), gpu)
    assert isinstance(r.index, rvc_amd.IVFFlatHIP) and isinstance(r.net_g.dec, rvc_amd.NSFGeneratorHIP)
    with pytest.raises(AttributeError, match="faiss is not installed"):
        pl.faiss.index_factory(768, "IVF16,Flat")
    rvc_amd.uninstall()
    assert rs.get_synthesizer is orig_get and vc_modules.get_synthesizer is orig_get and "faiss" not in sys.modules


# ---- Pipeline.vc / Pipeline.pipeline rebound by install(): device resident behind unmodified callers ------------------------

class _CpuSpy:
    """Counts Tensor.cpu() calls on CUDA tensors (what ``.cpu().numpy()`` hops of the reference's vc would be)."""

    def __init__(self, monkeypatch):
        self.calls = []
        orig = torch.Tensor.cpu

        def spy(t, *a, **k):
            if t.is_cuda:
                self.calls.append(tuple(t.shape))
            return orig(t, *a, **k)

        monkeypatch.setattr(torch.Tensor, "cpu", spy)


def _bind_reference_noise(net_g, noise, gpu):
    """Test-only shim: the reference draws its noise from the CPU generator inside net_g.infer; hand the same draws, call by
    call, to the HIP infer.  ``net_g.infer.rewind()`` starts over at the first call's draws."""
    real_infer = net_g.infer
    pos = [0]

    def infer_with_reference_noise(*a, **k):
        nz, nd = noise[pos[0]]
        pos[0] += 1
        return real_infer(*a, noise_zp=nz.to(gpu), noise_dec=nd.to(gpu), **k)

    infer_with_reference_noise.rewind = lambda: pos.__setitem__(0, 0)
    net_g.infer = infer_with_reference_noise


def _pipeline_fixture(gpu, rvc_tree, tmp_path):
    import types

    import rvc_amd

    d = load_golden("pipeline_v2_48k_3seg")
    seed = int(d["seed"])
    cfg = nsf_oracle.CONFIGS["v2_48k"]
    assert synth.weights_sha256(synth.make_dec_weights(cfg, seed)) == str(d["dec_sha256"])
    rvc_amd.install(device=gpu, operand="fp16")
    import infer.modules.vc.pipeline as pl
    import rvc.synthesizer as rs

    net_g, _ = rs.get_synthesizer(make_cpt(seed), gpu)
    noise = synth.infer_noise([int(x) for x in d["seg_frames"]], cfg.upp)
    import hashlib

    h = hashlib.sha256()
    for nz, nd in noise:
        h.update(nz.numpy().tobytes())
        h.update(nd.numpy().tobytes())
    assert h.hexdigest() == str(d["noise_sha256"]), "torch's CPU generator changed: the stored noise hash no longer matches"
    _bind_reference_noise(net_g, noise, gpu)
    config = types.SimpleNamespace(device=gpu, **{k[4:]: (bool(d[k]) if k == "cfg_is_half" else int(d[k])) for k in d if k.startswith("cfg_")})
    pipe = pl.Pipeline(cfg.sr, config)
    audio = synth.make_audio16k(int(d["n_audio"]), seed)
    p_all = (audio.shape[0] + 2 * pipe.t_pad) // pipe.window
    pitchf = synth.make_f0(1, p_all)[0]
    return d, cfg, pl, pipe, net_g, audio, synth.make_pitch(pitchf).numpy(), pitchf.numpy()


def test_unmodified_pipeline_entry_runs_device_resident_and_matches_the_reference(rvc_tree, gpu, tmp_path, monkeypatch):
    """``Pipeline.pipeline`` of the RVC-shaped skeleton (its own methods raise) after ``install()``: the waveform equals what the
    REAL reference ``Pipeline.pipeline`` returned for the same inputs (fixture pipeline_v2_48k_3seg: three segments, protect mix,
    int16-range scaling), and between HuBERT's output and the returned array exactly ONE tensor crosses to the host."""
    import rvc_amd

    d, cfg, pl, pipe, net_g, audio, pitch, pitchf = _pipeline_fixture(gpu, rvc_tree, tmp_path)
    assert pl.Pipeline.vc is rvc_amd.pipeline.vc_hip and pl.Pipeline.pipeline is rvc_amd.pipeline.pipeline_hip
    spy = _CpuSpy(monkeypatch)
    hub = synth.FakeHubert(768, int(d["seed"]))
    times = [0, 0, 0]
    out = pipe.pipeline(hub, net_g, int(d["sid"]), audio.copy(), times, 0, (pitch, pitchf), "", 0.75, 2, 3, cfg.sr, 0, 1, "v2",
                        float(d["protect"]))
    assert hub.calls == 3 and isinstance(out, np.ndarray) and out.shape == d["out"].shape
    assert spy.calls == [tuple(d["out"].shape)], "host hops between HuBERT and the returned audio: %s" % spy.calls
    e = rms(out / 32768.0, d["out"] / 32768.0)
    assert e <= 1e-3, "Pipeline.pipeline through the drop-in: RMS %.3e (int16 range / 32768) vs the reference" % e
    assert times[0] > 0 and times[2] > 0
    rvc_amd.uninstall()
    with pytest.raises(NotImplementedError):
        pipe.pipeline(hub, net_g, 3, audio.copy(), times, 0, (pitch, pitchf), "", 0.75, 2, 3, cfg.sr, 0, 1, "v2", 0.33)


def test_unmodified_vc_caller_without_and_with_an_index(rvc_tree, gpu, tmp_path, monkeypatch):
    """``Pipeline.vc`` (numpy in, numpy out, the reference's signature) on the first segment of the same fixture.  Without an
    index it reproduces the reference's segment.  WITH an ``IVFFlatHIP`` index it equals the numpy expressions of
    pipeline.py:118-159 (oracle search + inverse-square blend, x2, protect mix) fed through the same ``net_g.infer`` with the
    same noise -- one host hop each."""
    import rvc_amd
    from oracle import glue_oracle

    d, cfg, pl, pipe, net_g, audio, pitch, pitchf = _pipeline_fixture(gpu, rvc_tree, tmp_path)
    from scipy import signal as sg

    a = sg.filtfilt(pl.bh, pl.ah, audio)
    apad = np.pad(a, (pipe.t_pad, pipe.t_pad), mode="reflect")
    n0 = int(d["seg0_raw_len"]) // cfg.upp  # frames of the first segment as the reference cut it
    seg = apad[: (n0 + 1) * pipe.window]  # t + t_pad2 + window samples: one frame more than HuBERT's 2 * nq
    T0 = int(d["seg_frames"][0])
    pt = torch.from_numpy(pitch)[None, : T0 + 1].to(gpu)
    pf = torch.from_numpy(pitchf)[None, : T0 + 1].to(gpu)
    sid = torch.tensor([int(d["sid"])], device=gpu)
    spy = _CpuSpy(monkeypatch)
    hub = synth.FakeHubert(768, int(d["seed"]))
    o = pipe.vc(hub, net_g, sid, seg, pt, pf, [0, 0, 0], None, None, 0.75, "v2", float(d["protect"]))
    assert len(spy.calls) == 1 and o.dtype == np.float32
    ref0 = d["out"][: o.shape[0] - 2 * pipe.t_pad_tgt] / float(d["scale"])
    assert rms(o[pipe.t_pad_tgt: o.shape[0] - pipe.t_pad_tgt], ref0) <= 1e-3
    # ---- with an index: the retrieval branch of pipeline.py:113-138 ----
    idx = synth.make_ivf(3000, 768, seed=11)
    index = rvc_amd.IVFFlatHIP.from_arrays(idx["centroids"], idx["list_offsets"], idx["ids"], idx["vecs"], device=gpu)
    big_npy = idx["xb"]  # what index.reconstruct_n(0, ntotal) returns (pipeline.py:215); the HIP vc only checks it is not None
    net_g.infer.rewind()
    del spy.calls[:]
    oi = pipe.vc(hub, net_g, sid, seg, pt, pf, [0, 0, 0], index, big_npy, 0.75, "v2", float(d["protect"]))
    assert len(spy.calls) == 1 and oi.shape == o.shape
    assert rms(oi, o) > 1e-2, "the index changed nothing: the retrieval branch did not run"
    # expected: the reference's numpy lines on the CPU oracle, then the SAME HIP net_g.infer with the same noise
    feats = hub.extract_features(torch.zeros(1, seg.shape[0]), torch.zeros(1, seg.shape[0], dtype=torch.bool), 12)[0]
    hub.calls -= 1
    npy = feats[0].numpy()
    blended = torch.from_numpy(ivf_oracle.search_blend(idx, npy, 0.75, 8)).unsqueeze(0)
    exp_feats = glue_oracle.expand_protect(blended, feats, pf.cpu(), float(d["protect"]), seg.shape[0] // pipe.window)
    assert exp_feats.shape[1] == T0
    net_g.infer.rewind()
    with torch.no_grad():
        exp = net_g.infer(exp_feats.to(gpu), torch.tensor([T0], device=gpu), sid, pitch=pt[:, :T0], pitchf=pf[:, :T0])[0, 0]
    e = rms(oi, exp.cpu())
    assert e <= 2e-5, "Pipeline.vc with an index vs oracle retrieval + the same infer: RMS %.3e" % e
    # index_rate 0 and a missing big_npy are the reference's "no retrieval" guards (pipeline.py:113-117)
    for args in ((index, big_npy, 0), (index, None, 0.75)):
        net_g.infer.rewind()
        assert np.array_equal(pipe.vc(hub, net_g, sid, seg, pt, pf, [0, 0, 0], args[0], args[1], args[2], "v2", float(d["protect"])), o)


def test_webui_defaults_through_the_rebound_pipeline(rvc_tree, gpu, tmp_path, monkeypatch):
    """The WebUI's single-inference defaults (web.py:756-802) through the rebound ``Pipeline.pipeline``: f0_method "rmvpe" with
    the UI's filter_radius 3 (NOT a voicing threshold: rvc/f0/gen.py:113 passes the constant 0.03), an index FILE with
    index_rate 0.75, rms_mix_rate 0.25, protect 0.33, if_f0 1.  Expected waveform: fixture pipeline_v2_48k_webui, returned by the
    REAL reference ``Pipeline.pipeline`` (its own read_index / reconstruct_n / search+blend / Generator.calculate /
    RMVPE.compute_f0 / change_rms lines; stand-ins only for HuBERT, the RMVPE network, the faiss object = CPU oracle, and
    ``librosa.feature.rms``).  Executes ``pipeline_hip``'s index branch, ``_rmvpe_on_device`` and ``glue.change_rms``."""
    import hashlib
    import types

    import rvc_amd
    from oracle import ivf_oracle as io

    d = load_golden("pipeline_v2_48k_webui")
    seed = int(d["seed"])
    cfg = nsf_oracle.CONFIGS["v2_48k"]
    assert synth.weights_sha256(synth.make_dec_weights(cfg, seed)) == str(d["dec_sha256"])
    rvc_amd.install(device=gpu, operand="fp16")
    import infer.modules.vc.pipeline as pl
    import rvc.synthesizer as rs

    net_g, _ = rs.get_synthesizer(make_cpt(seed), gpu)
    noise = synth.infer_noise([int(x) for x in d["seg_frames"]], cfg.upp)
    h = hashlib.sha256()
    for nz, nd in noise:
        h.update(nz.numpy().tobytes())
        h.update(nd.numpy().tobytes())
    assert h.hexdigest() == str(d["noise_sha256"])
    _bind_reference_noise(net_g, noise, gpu)
    config = types.SimpleNamespace(device=gpu, **{k[4:]: (bool(d[k]) if k == "cfg_is_half" else int(d[k])) for k in d if k.startswith("cfg_")})
    pipe = pl.Pipeline(cfg.sr, config)
    fake = synth.FakeRMVPE(gpu, seed)
    pipe.f0_gen = types.SimpleNamespace(rmvpe=fake, is_half=False, device=gpu)  # no ``calculate``: a host fallback would raise
    audio = synth.make_audio16k(int(d["n_audio"]), seed)
    path = str(tmp_path / "added.index")
    io.write_index(synth.make_ivf(int(d["index_n"]), int(d["index_d"]), seed=int(d["index_seed"])), path)
    f0_calls = []
    real_rmvpe_f0 = rvc_amd.glue.rmvpe_f0

    def rmvpe_f0_spy(sal, p_len, key, thred):
        r = real_rmvpe_f0(sal, p_len, key, thred)
        f0_calls.append((tuple(sal.shape), p_len, key, thred, r[0].clone(), r[1].clone()))
        return r

    monkeypatch.setattr(rvc_amd.glue, "rmvpe_f0", rmvpe_f0_spy)
    rms_calls = []
    real_change_rms = rvc_amd.glue.change_rms
    monkeypatch.setattr(rvc_amd.glue, "change_rms", lambda *a: (rms_calls.append(a[4]), real_change_rms(*a))[1])
    spy = _CpuSpy(monkeypatch)
    hub = synth.FakeHubert(768, seed)
    times = [0, 0, 0]
    out = pipe.pipeline(hub, net_g, int(d["sid"]), audio.copy(), times, int(d["f0_up_key"]), "rmvpe", path, float(d["index_rate"]), 1,
                        int(d["filter_radius"]), cfg.sr, 0, float(d["rms_mix_rate"]), "v2", float(d["protect"]))
    assert hub.calls == 3 and fake.model.calls == 1 and rms_calls == [0.25]
    assert isinstance(out, np.ndarray) and out.shape == d["out"].shape
    assert spy.calls == [tuple(d["out"].shape)], "host hops between HuBERT and the returned audio: %s" % spy.calls
    # the f0 track: threshold 0.03 whatever filter_radius says; bins identical and Hz to the last ulps vs the reference's track
    (_, p_len, key, thred, pitch, pitchf), = f0_calls
    assert thred == 0.03 and key == int(d["f0_up_key"]) and p_len == d["pitch"].shape[0]
    assert int((pitchf > 0).sum()) == int(d["voiced_frames"]) > 0
    assert np.array_equal(pitch[0].cpu().numpy(), d["pitch"])
    assert np.allclose(pitchf[0].cpu().numpy(), d["pitchf"], rtol=2e-6, atol=0)
    e = rms(out / 32768.0, d["out"] / 32768.0)
    assert e <= 1e-3, "WebUI defaults through the drop-in: RMS %.3e (int16 range / 32768) vs the reference" % e
    assert times[0] > 0 and times[1] > 0 and times[2] > 0
    # an index on another kind of device object is refused loudly, not searched on the wrong GPU
    with pytest.raises(rvc_amd.RvcmiError, match="no CPU fallback"):
        rvc_amd.read_index(path, device="cpu")


def test_segments_of_a_file_in_one_batched_call_equal_the_sequential_pipeline(rvc_tree, gpu, tmp_path, monkeypatch):
    """``Pipeline.pipeline`` on the three-segment fixture with every segment in ONE ``net_g.infer`` call (ragged batch,
    ``rvc_amd.pipeline.infer_segments``) against the call-per-segment order of the reference (RVCMI_PIPELINE_BATCH=0): bit-equal
    waveforms with the ResBlock kernel family pinned, both within 1e-3 of what the REAL reference pipeline returned, and the same
    number of draws from the seeded generator (the noise of item b is drawn as the b-th sequential call would draw it)."""
    import rvc_amd

    d, cfg, pl, pipe, net_g, audio, pitch, pitchf = _pipeline_fixture(gpu, rvc_tree, tmp_path)
    noise = synth.infer_noise([int(x) for x in d["seg_frames"]], cfg.upp)
    # un-shim: the fixture bound a per-call noise injector; this test needs the HIP partial itself plus a batch-aware injector
    import functools

    from rvc_amd.front import infer_hip

    base = functools.partial(infer_hip, net_g, net_g._rvcmi_front)
    calls = []

    def infer_with_reference_noise(phone, lengths, sid, *a, **k):
        B = phone.shape[0]
        lens = [int(x) for x in lengths.tolist()]
        calls.append(lens)
        first = sum(len(c) for c in calls[:-1])
        Tm = phone.shape[1]
        nz = torch.zeros(B, 192, Tm)
        nd = torch.zeros(B, Tm * cfg.upp)
        for b, n in enumerate(lens):
            z_b, d_b = noise[first + b]
            assert z_b.shape[2] == n
            nz[b, :, :n], nd[b, :n * cfg.upp] = z_b[0], d_b[0]
        k["noise_zp"], k["noise_dec"] = nz.to(gpu), nd.to(gpu)
        return base(phone, lengths, sid, *a, **k)

    infer_with_reference_noise._rvcmi_ragged = True
    net_g.infer = infer_with_reference_noise
    for key, val in (("RB_STREAM", 0), ("NO_RB_SPLIT", 1)):  # same kernel family for B = 3 and B = 1
        net_g.dec.set_option(key, val)
    hub = synth.FakeHubert(768, int(d["seed"]))
    args = (hub, net_g, int(d["sid"]), None, None, 0, (pitch, pitchf), "", 0.75, 2, 3, cfg.sr, 0, 1, "v2", float(d["protect"]))

    def run():
        a = list(args)
        a[3], a[4] = audio.copy(), [0, 0, 0]
        del calls[:]
        return pipe.pipeline(*a)

    monkeypatch.setenv("RVCMI_PIPELINE_BATCH", "1")
    out_b = run()
    assert calls == [[int(x) for x in d["seg_frames"]]], calls  # ONE call, three ragged items
    monkeypatch.setenv("RVCMI_PIPELINE_BATCH", "0")
    out_s = run()
    assert calls == [[int(x)] for x in d["seg_frames"]]
    assert np.array_equal(out_b, out_s), "batched vs sequential segments: max diff %g" % float(np.abs(out_b - out_s).max())
    for o in (out_b, out_s):
        assert rms(o / 32768.0, d["out"] / 32768.0) <= 1e-3
    # without injected noise: the batched path draws item b's noise as the b-th sequential call would -- same seed, same waveform,
    # same generator state afterwards
    net_g.infer = base
    res = []
    for mode in ("1", "0"):
        monkeypatch.setenv("RVCMI_PIPELINE_BATCH", mode)
        torch.manual_seed(5)
        o = run()
        res.append((o, torch.cuda.get_rng_state(gpu).clone()))
    assert np.array_equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert not np.array_equal(res[0][0], out_b)  # (different noise than the reference's CPU draws)


def test_convert_files_batches_several_files_and_equals_the_per_file_pipeline(rvc_tree, gpu, tmp_path, monkeypatch):
    """``Pipeline.convert_files`` (bound by ``install()``; ``rvc_amd.pipeline.convert_files``) = the loop body of ``VC.vc_multi``
    (infer/modules/vc/modules.py:201-266) for several inputs at once: five 16 kHz inputs of different lengths (one, two and three
    segments; the WebUI-default configuration with an index FILE, rmvpe, rms_mix_rate 0.25, protect 0.33).

      * every waveform is BIT-equal to what ``Pipeline.pipeline`` returns for that file alone (same seed: the noise of every segment
        is drawn in the order of the sequential loop; ResBlock kernel family pinned, since a batch and a single clip may otherwise
        pick different families per stage);
      * the HuBERT frames of ALL segments of ALL files go through ONE retrieval call, the segments through ragged ``net_g.infer``
        batches, and exactly one tensor per file crosses to the host;
      * the fixture file (first in the list, reference noise injected for its three segments) is within 1e-3 of the waveform the REAL
        reference ``Pipeline.pipeline`` returned (golden pipeline_v2_48k_webui)."""
    import functools
    import types

    import rvc_amd
    from oracle import ivf_oracle as io
    from rvc_amd.front import infer_hip

    d = load_golden("pipeline_v2_48k_webui")
    seed = int(d["seed"])
    cfg = nsf_oracle.CONFIGS["v2_48k"]
    rvc_amd.install(device=gpu, operand="fp16")
    import infer.modules.vc.pipeline as pl
    import rvc.synthesizer as rs

    assert pl.Pipeline.convert_files is rvc_amd.pipeline.convert_files
    net_g, _ = rs.get_synthesizer(make_cpt(seed), gpu)
    for key, val in (("RB_STREAM", 0), ("NO_RB_SPLIT", 1)):  # same kernel family for every batch size
        net_g.dec.set_option(key, val)
    for key, val in (("FR_NJ", 1), ("FR_FFN_SPLIT", 1), ("FR_WN_SPLIT", 1)):  # ... in the front too (tile height, the split FFN and the tap-split WN gate follow the tile count)
        net_g._rvcmi_front.set_option(key, val)
    config = types.SimpleNamespace(device=gpu, **{k[4:]: (bool(d[k]) if k == "cfg_is_half" else int(d[k])) for k in d if k.startswith("cfg_")})
    pipe = pl.Pipeline(cfg.sr, config)
    fake = synth.FakeRMVPE(gpu, seed)
    pipe.f0_gen = types.SimpleNamespace(rmvpe=fake, is_half=False, device=gpu)  # no ``calculate``: a host fallback would raise
    path = str(tmp_path / "added.index")
    io.write_index(synth.make_ivf(int(d["index_n"]), int(d["index_d"]), seed=int(d["index_seed"])), path)
    n0 = int(d["n_audio"])
    audios = [synth.make_audio16k(n0, seed), synth.make_audio16k(16000 * 2 + 77, seed + 1), synth.make_audio16k(n0 // 2 + 4321, seed + 2),
              synth.make_audio16k(16000, seed + 3), synth.make_audio16k(50000, seed + 4)]  # t_max = 16000 samples: 3, 3, 2, 1, 4 segments
    hub = synth.FakeHubert(768, seed)
    tail = (int(d["f0_up_key"]), "rmvpe", path, float(d["index_rate"]), 1, int(d["filter_radius"]), cfg.sr, 0, float(d["rms_mix_rate"]),
            "v2", float(d["protect"]))
    # --- per file, the rebound Pipeline.pipeline
    torch.manual_seed(5)
    seq = [pipe.pipeline(hub, net_g, int(d["sid"]), a.copy(), [0, 0, 0], *tail) for a in audios]
    state_seq = torch.cuda.get_rng_state(gpu).clone()
    nseg = hub.calls
    assert nseg > len(audios) and fake.model.calls == len(audios)  # (some files have several segments)
    # --- all files at once
    searches, infers = [], []
    real_blend = rvc_amd.glue._blend_expand_into
    monkeypatch.setattr(rvc_amd.glue, "_blend_expand_into", lambda out, f, index, *a: (searches.append((int(f.shape[0]), index is not None)), real_blend(out, f, index, *a))[1])
    base = net_g.infer

    def counting_infer(phone, lengths, *a, **k):
        infers.append([int(x) for x in lengths.tolist()])
        return base(phone, lengths, *a, **k)

    counting_infer._rvcmi_ragged = True
    net_g.infer = counting_infer
    spy = _CpuSpy(monkeypatch)
    hub.calls = 0
    times = [0, 0, 0]
    torch.manual_seed(5)
    bat = pipe.convert_files(hub, net_g, int(d["sid"]), [a.copy() for a in audios], times, *tail)
    assert torch.equal(torch.cuda.get_rng_state(gpu), state_seq)  # the same draws from the seeded generator
    assert hub.calls == nseg and len(bat) == len(audios)
    assert len(searches) == 1 and searches[0][1], searches          # ONE retrieval call for every segment of every file
    assert sum(len(c) for c in infers) == nseg and len(infers) == 1, infers  # one ragged batch (well under MAX_BATCH_FRAMES)
    assert spy.calls == [tuple(o.shape) for o in bat], "host hops: %s" % spy.calls  # one per file: the finished audio
    for i, (o, r) in enumerate(zip(bat, seq)):
        assert o.shape == r.shape and np.array_equal(o, r), "file %d: batched vs per-file max diff %g" % (i, float(np.abs(o - r).max()))
    assert times[0] > 0 and times[1] > 0 and times[2] > 0
    # --- bounded batches: the same result in several ragged calls / several retrieval calls
    monkeypatch.setattr(rvc_amd.pipeline, "MAX_BATCH_FRAMES", 2 * max(max(c) for c in infers))
    monkeypatch.setattr(rvc_amd.pipeline, "MAX_BATCH_QUERIES", 400)
    del searches[:], infers[:]
    torch.manual_seed(5)
    bat2 = pipe.convert_files(hub, net_g, int(d["sid"]), [a.copy() for a in audios], [0, 0, 0], *tail)
    assert len(infers) > 1 and len(searches) > 1
    for o, r in zip(bat2, seq):
        assert np.array_equal(o, r)
    # --- the fixture file against the REAL reference's output: its three segments get the reference's noise draws
    noise = synth.infer_noise([int(x) for x in d["seg_frames"]], cfg.upp)
    raw_infer = functools.partial(infer_hip, net_g, net_g._rvcmi_front)

    def infer_with_reference_noise(phone, lengths, sid, *a, **k):
        nz, nd = k["noise_zp"].clone(), k["noise_dec"].clone()  # (infer_segments drew them; items 0..2 are the fixture file's segments)
        for b, (z_b, d_b) in enumerate(noise):
            n = z_b.shape[2]
            assert int(lengths[b]) == n
            nz[b, :, :n], nd[b, :n * cfg.upp] = z_b[0].to(gpu), d_b[0].to(gpu)
        k["noise_zp"], k["noise_dec"] = nz, nd
        return raw_infer(phone, lengths, sid, *a, **k)

    infer_with_reference_noise._rvcmi_ragged = True
    net_g.infer = infer_with_reference_noise
    monkeypatch.setattr(rvc_amd.pipeline, "MAX_BATCH_FRAMES", 32768)
    out = pipe.convert_files(hub, net_g, int(d["sid"]), [a.copy() for a in audios], [0, 0, 0], *tail)[0]
    e = rms(out / 32768.0, d["out"] / 32768.0)
    assert out.shape == d["out"].shape and e <= 1e-3, "fixture file inside a five-file batch: RMS %.3e vs the reference" % e
    rvc_amd.uninstall()
    assert not hasattr(pl.Pipeline, "convert_files")


def test_convert_files_with_the_launcher_defaults_meets_the_reference_golden(rvc_tree, gpu, tmp_path, monkeypatch, capsys):
    """What PRODUCTION runs: ``Pipeline.convert_files`` on five files with NO option pinned -- the 13-segment ragged batch picks the
    kernel families, tile heights and the FFN realisation the launchers choose for that shape (not the ones a single file's call picks,
    which is what the bit-equality test above pins).  The fixture file (reference noise injected for its three segments) must be within
    1e-3 RMS of the waveform the REAL reference ``Pipeline.pipeline`` returned (golden pipeline_v2_48k_webui); the same for the file
    converted alone (the other set of launcher choices).  The distance between the two realisations is stated as an RMS (the round-5
    note "1.3e-3 of full scale apart" was a max-abs figure) and bounded: a file's waveform may depend on what it was batched with only
    far inside the parity bar."""
    import functools
    import types

    import rvc_amd
    from oracle import ivf_oracle as io
    from rvc_amd.front import infer_hip

    d = load_golden("pipeline_v2_48k_webui")
    seed = int(d["seed"])
    cfg = nsf_oracle.CONFIGS["v2_48k"]
    rvc_amd.install(device=gpu, operand="fp16")
    import infer.modules.vc.pipeline as pl
    import rvc.synthesizer as rs

    net_g, _ = rs.get_synthesizer(make_cpt(seed), gpu)  # launcher defaults: nothing pinned
    config = types.SimpleNamespace(device=gpu, **{k[4:]: (bool(d[k]) if k == "cfg_is_half" else int(d[k])) for k in d if k.startswith("cfg_")})
    pipe = pl.Pipeline(cfg.sr, config)
    pipe.f0_gen = types.SimpleNamespace(rmvpe=synth.FakeRMVPE(gpu, seed), is_half=False, device=gpu)
    path = str(tmp_path / "added.index")
    io.write_index(synth.make_ivf(int(d["index_n"]), int(d["index_d"]), seed=int(d["index_seed"])), path)
    n0 = int(d["n_audio"])
    audios = [synth.make_audio16k(n0, seed), synth.make_audio16k(16000 * 2 + 77, seed + 1), synth.make_audio16k(n0 // 2 + 4321, seed + 2),
              synth.make_audio16k(16000, seed + 3), synth.make_audio16k(50000, seed + 4)]
    hub = synth.FakeHubert(768, seed)
    tail = (int(d["f0_up_key"]), "rmvpe", path, float(d["index_rate"]), 1, int(d["filter_radius"]), cfg.sr, 0, float(d["rms_mix_rate"]),
            "v2", float(d["protect"]))
    noise = synth.infer_noise([int(x) for x in d["seg_frames"]], cfg.upp)
    raw_infer = functools.partial(infer_hip, net_g, net_g._rvcmi_front)
    shapes = []

    def infer_with_reference_noise(phone, lengths, sid, *a, **k):
        shapes.append([int(x) for x in lengths.tolist()])
        nz, nd = k["noise_zp"].clone(), k["noise_dec"].clone()
        for b, (z_b, d_b) in enumerate(noise):  # items 0..2 = the fixture file's segments
            n = z_b.shape[2]
            assert int(lengths[b]) == n
            nz[b, :, :n], nd[b, :n * cfg.upp] = z_b[0].to(gpu), d_b[0].to(gpu)
        k["noise_zp"], k["noise_dec"] = nz, nd
        return raw_infer(phone, lengths, sid, *a, **k)

    infer_with_reference_noise._rvcmi_ragged = True
    net_g.infer = infer_with_reference_noise
    ref = d["out"] / 32768.0

    def convert(files):
        torch.manual_seed(5)
        return pipe.convert_files(hub, net_g, int(d["sid"]), [a.copy() for a in files], [0, 0, 0], *tail)[0] / 32768.0

    in_batch = convert(audios)
    assert len(shapes) == 1 and len(shapes[0]) == 13, shapes  # ONE 13-segment ragged batch
    alone = convert(audios[:1])
    assert shapes[1] == [int(x) for x in d["seg_frames"]]
    e_batch, e_alone, apart = rms(in_batch, ref), rms(alone, ref), rms(in_batch, alone)
    # the two FFN realisations on the SAME batch (everything else as the launcher chose): the front's contribution alone
    net_g._rvcmi_front.set_option("FR_FFN_SPLIT", 1)
    split = convert(audios)
    net_g._rvcmi_front.set_option("FR_FFN_SPLIT", 0)
    fused = convert(audios)
    net_g._rvcmi_front.set_option("FR_FFN_SPLIT", None)
    ffn_rms, ffn_max = rms(split, fused), float(np.abs(split - fused).max())
    with capsys.disabled():
        print("\n[convert_files, launcher defaults] fixture file vs the reference golden: in the 13-segment batch RMS %.3e, alone %.3e; "
              "batch vs alone RMS %.3e (max abs %.3e); split-FFN vs fused-FFN on the same batch RMS %.3e (max abs %.3e); waveform RMS %.3f"
              % (e_batch, e_alone, apart, float(np.abs(in_batch - alone).max()), ffn_rms, ffn_max, float(np.sqrt((ref ** 2).mean()))))
    assert in_batch.shape == ref.shape and e_batch <= 1e-3, "fixture file inside the un-pinned five-file batch: RMS %.3e vs the reference" % e_batch
    assert e_alone <= 1e-3, "fixture file alone, un-pinned: RMS %.3e vs the reference" % e_alone
    assert apart <= 5e-4 and ffn_rms <= 5e-4, "a file's waveform moves by RMS %.3e with its batch (FFN realisations %.3e apart)" % (apart, ffn_rms)
    rvc_amd.uninstall()


def test_front_at_benchmark_size_matches_the_reference_modules(gpu):
    """enc_p + z_p + flow^-1 at T = 1198 (global attention over the whole 10 s clip) against the REFERENCE modules' own output
    (fixture bigfront_v2_B1_T1198_z), not only against the oracle restatement."""
    import rvc_amd

    d = load_golden("bigfront_v2_B1_T1198_z")
    seed, T = int(d["seed"]), int(d["T"])
    fcfg = FrontConfig()
    wf = synth.make_front_weights(fcfg, seed)
    assert synth.weights_sha256(wf) == str(d["weights_sha256"])
    front = rvc_amd.FrontHIP(vars(fcfg), wf, device=gpu, operand="fp16", max_B=1, max_T=T)
    phone = synth.make_phone(1, T, 768, seed).to(gpu)
    pitch = synth.make_pitch(synth.make_f0(1, T)).to(gpu)
    noise = torch.randn(1, 192, T, generator=torch.Generator().manual_seed(seed + 9)).to(gpu)
    g = wf["emb_g.weight"][int(d["sid"][0])].reshape(1, -1, 1).to(gpu)
    z = front(phone, pitch, torch.tensor([T], device=gpu), g, 0, noise=noise)
    e = rms(z.cpu(), d["z"])
    assert e <= 5e-3, "front z at T=1198 vs the reference modules: RMS %.3e (z RMS 1.5; same bar as the short goldens)" % e
