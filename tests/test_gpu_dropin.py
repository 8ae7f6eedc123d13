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
