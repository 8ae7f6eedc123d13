"""TEST INFRASTRUCTURE ONLY -- pins the oracle to the REFERENCE ITSELF and writes tests/golden/*.npz.

Runs only where /root/reference is mounted (the build container); the GPU box and the tests read the
committed fixtures, never the reference.  For every case below the reference's own modules
(``rvc.layers.nsf.NSFGenerator``, ``rvc.layers.generators.Generator``, and for one case the whole
``SynthesizerTrnMsNSFsid.infer``) are executed on torch-CPU fp32 with seeded synthetic weights, the
oracle restatement (oracle/nsf_oracle.py) is checked against them to <= 2e-6, and inputs + injected
noise + outputs are stored.  Weights are NOT stored (63 MB): tests regenerate them from the seed with
oracle/synth.py and verify the sha256 recorded here.

    python -m oracle.make_golden            # regenerate every fixture
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("RVC_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")

from oracle import front_oracle, nsf_oracle, synth  # noqa: E402
from oracle.front_oracle import FrontConfig  # noqa: E402
from oracle.nsf_oracle import CONFIGS, GenConfig  # noqa: E402


def build_reference_dec(cfg: GenConfig, w):
    sys.path.insert(0, REF)
    from rvc.layers.generators import Generator
    from rvc.layers.nsf import NSFGenerator

    if cfg.use_f0:
        net = NSFGenerator(cfg.inter_channels, "1", cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes, cfg.upsample_rates,
                           cfg.upsample_initial_channel, cfg.upsample_kernel_sizes, cfg.gin_channels, cfg.sr)
    else:
        net = Generator(cfg.inter_channels, "1", cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes, cfg.upsample_rates,
                        cfg.upsample_initial_channel, cfg.upsample_kernel_sizes, cfg.gin_channels)
    net.eval()
    net.remove_weight_norm()
    net.load_state_dict(w, strict=True)
    return net


def dec_case(name: str, cfg_name: str, B: int, T: int, seed: int = 1234, use_f0: bool = True, n_res=None, noise_seed: int = 114514):
    cfg = CONFIGS[cfg_name]
    if not use_f0:
        cfg = GenConfig(**{**vars(cfg), "use_f0": False})
    w = synth.make_dec_weights(cfg, seed)
    net = build_reference_dec(cfg, w)
    z, f0, g = synth.make_dec_inputs(cfg, B, T, seed)
    torch.manual_seed(noise_seed)  # the reference draws rand(1,1,1) then randn_like([B,T*upp,1]) from the global CPU generator
    with torch.no_grad():
        ref = net(z, f0, g=g, n_res=n_res) if use_f0 else net(z, g=g, n_res=n_res)
    noise = nsf_oracle.reference_noise(B, T, cfg.upp, noise_seed) if use_f0 else None
    taps = {}
    with torch.no_grad():
        ora = nsf_oracle.generator_forward(cfg, w, z, f0, g, noise, n_res=n_res, taps=taps)
    err = (ora - ref).abs().max().item()
    assert err < 2e-6, "%s: oracle restatement deviates from the reference by %g" % (name, err)
    out = dict(cfg_name=cfg_name, use_f0=use_f0, seed=seed, noise_seed=noise_seed, weights_sha256=synth.weights_sha256(w),
               z=z.numpy(), g=g.numpy(), out=ref.numpy(), n_res=-1 if n_res is None else int(n_res),
               oracle_max_abs_dev=err)
    if use_f0:
        out.update(f0=f0.numpy(), noise=noise.numpy(), har=taps["har"].numpy())
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("%-28s ref rms %.3f  oracle-vs-reference max dev %.2e  (%d KB)" % (
        name, ref.pow(2).mean().sqrt().item(), err, os.path.getsize(os.path.join(GOLD, name + ".npz")) // 1024))


def dec_full_case(name: str = "full_v2_48k_T1198_voiced", cfg_name: str = "v2_48k", T: int = 1198, seed: int = 1234,
                  noise_seed: int = 114514):
    """BASELINE configs[1] at full size: one 10 s clip (T = 1198 frames), VOICED f0 (80 % of the frames,
    the inputs bench.py times).  Only the reference's output waveform is stored (2.3 MB); inputs and noise are regenerated
    from the seeds and verified by sha256.  Pins ``NSFGenerator.forward`` (rvc/layers/nsf.py:145-191) -- in particular the
    cumulative harmonic phase over 1198 frames (generators.py:175-194) -- at the size the benchmark runs."""
    import hashlib

    cfg = CONFIGS[cfg_name]
    w = synth.make_dec_weights(cfg, seed)
    net = build_reference_dec(cfg, w)
    z, f0, g = synth.make_dec_inputs(cfg, 1, T, seed)
    assert float((f0 > 0).float().mean()) > 0.5, "the full-size case must be voiced"
    torch.manual_seed(noise_seed)
    with torch.no_grad():
        ref = net(z, f0, g=g)
    noise = nsf_oracle.reference_noise(1, T, cfg.upp, noise_seed)
    with torch.no_grad():
        ora = nsf_oracle.generator_forward(cfg, w, z, f0, g, noise)
    err = (ora - ref).abs().max().item()
    assert err < 2e-6, "%s: oracle restatement deviates from the reference by %g" % (name, err)
    sha = lambda t: hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), cfg_name=cfg_name, seed=seed, noise_seed=noise_seed, T=T,
                        weights_sha256=synth.weights_sha256(w), inputs_sha256=sha(z) + sha(f0) + sha(g) + sha(noise),
                        out=ref.numpy(), oracle_max_abs_dev=err)
    print("%-28s ref rms %.3f  oracle-vs-reference max dev %.2e  (%d KB)" % (
        name, ref.pow(2).mean().sqrt().item(), err, os.path.getsize(os.path.join(GOLD, name + ".npz")) // 1024))


def infer_case(name: str = "infer_v2_48k_T20", T: int = 20, seed: int = 1234):
    """The whole ``SynthesizerTrnMsNSFsid.infer`` (rvc/layers/synthesizers.py:160-203) of the reference, loaded
    through the reference's own ``get_synthesizer`` from a synthetic fp16 / legacy-weight-norm checkpoint.  What the
    decoder receives inside ``infer`` (z*x_mask, pitchf, g) and what it returns are captured with a forward hook, so
    the fixture pins the drop-in boundary *in situ*, including the RNG draw order (randn_like for z_p first)."""
    sys.path.insert(0, REF)
    from rvc.layers.synthesizers import SynthesizerTrnMsNSFsid
    from rvc.synthesizer import get_synthesizer

    cfg = CONFIGS["v2_48k"]
    cpt, expect = synth.make_legacy_checkpoint(cfg, "v2", seed)
    torch.manual_seed(seed)
    donor = SynthesizerTrnMsNSFsid(*cpt["config"], encoder_dim=768, use_f0=True)
    for k, v in donor.state_dict().items():
        if k.startswith(("enc_q.", "dec.")):
            continue
        k2 = k.replace("parametrizations.weight.original0", "weight_g").replace("parametrizations.weight.original1", "weight_v")
        cpt["weight"][k2] = v.half()
    net_g, _ = get_synthesizer(cpt, torch.device("cpu"))
    dec_sd = {k: v.clone() for k, v in net_g.dec.state_dict().items()}
    for k, v in expect.items():  # the loader's folded fp32 weights are what the HIP side must receive
        assert torch.allclose(dec_sd[k], v, rtol=0, atol=2e-7 * max(1.0, v.abs().max().item())), k
    cap = {}

    def hook(mod, args, kwargs, output):
        cap["z"], cap["f0"] = args[0].detach().clone(), args[1].detach().clone()
        cap["g"] = kwargs["g"].detach().clone()
        cap["out"] = output.detach().clone()

    hd = net_g.dec.register_forward_hook(hook, with_kwargs=True)
    phone = synth.make_phone(1, T, 768, seed)
    pitchf = synth.make_f0(1, T)
    pitch = torch.clamp((1127 * torch.log(1 + pitchf / 700) - 1127 * np.log(1 + 50 / 700)) * 254 / (1127 * np.log(1 + 1100 / 700) - 1127 * np.log(1 + 50 / 700)) + 1, 1, 255).round().long()
    torch.manual_seed(114514)
    with torch.no_grad():
        o = net_g.infer(phone, torch.tensor([T]), torch.tensor([0]), pitch, pitchf)
    hd.remove()
    gen = torch.Generator().manual_seed(114514)
    torch.randn(1, 192, T, generator=gen)  # z_p draw (synthesizers.py:188)
    torch.rand(1, 1, 1, generator=gen)  # rand_ini (generators.py:164)
    noise = torch.randn(1, T * cfg.upp, 1, generator=gen).squeeze(-1)
    with torch.no_grad():
        ora = nsf_oracle.generator_forward(cfg, dec_sd, cap["z"], cap["f0"], cap["g"], noise)
    err = (ora - o).abs().max().item()
    assert err < 2e-6, "infer case: oracle deviates by %g" % err
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), cfg_name="v2_48k", seed=seed, z=cap["z"].numpy(), f0=cap["f0"].numpy(),
                        g=cap["g"].numpy(), noise=noise.numpy(), out=o.numpy(), oracle_max_abs_dev=err,
                        weights_sha256=synth.weights_sha256(dec_sd), **{"w::" + k: v.numpy() for k, v in dec_sd.items()
                                                                        if v.numel() <= 4096})
    print("%-28s ref rms %.3f  oracle-vs-reference max dev %.2e  (loader-folded weights verified)" % (name, o.pow(2).mean().sqrt().item(), err))
    return dec_sd


def build_reference_net(cfg: GenConfig, fcfg: FrontConfig, w_dec, w_front):
    """The reference's whole synthesizer (rvc/layers/synthesizers.py:23-113) carrying the seeded weights."""
    sys.path.insert(0, REF)
    from rvc.layers.synthesizers import SynthesizerTrnMsNSFsid

    cl = [1025, 32, fcfg.inter_channels, fcfg.hidden_channels, fcfg.filter_channels, fcfg.n_heads, fcfg.n_layers, fcfg.kernel_size, 0, "1",
          cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes, cfg.upsample_rates, cfg.upsample_initial_channel,
          cfg.upsample_kernel_sizes, fcfg.spk_embed_dim, cfg.gin_channels, cfg.sr]
    net = SynthesizerTrnMsNSFsid(*cl, encoder_dim=fcfg.in_channels, use_f0=True)
    del net.enc_q
    net.eval()
    net.remove_weight_norm()
    sd = net.state_dict()
    for k, v in w_front.items():
        assert k in sd and sd[k].shape == v.shape, k
    sd.update(w_front)
    sd.update({"dec." + k: v for k, v in w_dec.items()})
    net.load_state_dict(sd, strict=True)
    return net


def front_case(name: str, B: int, T: int, lengths, flow_head=None, in_channels: int = 768, seed: int = 1234):
    """enc_p + z_p + flow^-1 of the REFERENCE modules (encoders.py:134-159, synthesizers.py:182-183, residuals.py:319-321)
    on seeded weights; the oracle restatement (oracle/front_oracle.py) is checked against them."""
    cfg, fcfg = CONFIGS["v2_48k"], FrontConfig(in_channels=in_channels)
    wf = synth.make_front_weights(fcfg, seed)
    net = build_reference_net(cfg, fcfg, synth.make_dec_weights(cfg, seed), wf)
    phone = synth.make_phone(B, T, in_channels, seed)
    pitchf = synth.make_f0(B, T)
    pitch = synth.make_pitch(pitchf)
    lengths = torch.tensor(lengths, dtype=torch.long)
    sid = torch.arange(B, dtype=torch.long) * 5
    with torch.no_grad():
        g = net.emb_g(sid).unsqueeze(-1)
        m_p, logs_p, x_mask = net.enc_p(phone, pitch, lengths, flow_head)
        gen = torch.Generator().manual_seed(seed + 9)
        noise = torch.randn(m_p.shape, generator=gen)
        z_p = (m_p + torch.exp(logs_p) * noise * 0.66666) * x_mask
        z = net.flow(z_p, x_mask, g=g, reverse=True)
        taps = {}
        z2, m1, g2 = front_oracle.infer_front(fcfg, wf, phone, pitch, lengths, sid, noise, flow_head, taps)
    err = max((z2 - z).abs().max().item(), (taps["z_p"] - z_p).abs().max().item())
    assert err < 2e-5, "%s: front oracle deviates from the reference by %g" % (name, err)
    cl = lambda t: t.transpose(1, 2).contiguous().numpy()  # channels-last, the layout of the HIP debug taps
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), in_channels=in_channels, seed=seed, weights_sha256=synth.weights_sha256(wf),
                        phone=phone.numpy(), pitch=pitch.numpy(), lengths=lengths.numpy(), sid=sid.numpy(), g=g.numpy(), noise=noise.numpy(),
                        flow_head=-1 if flow_head is None else int(flow_head), z=(z * x_mask).numpy(), z_p=cl(z_p),
                        emb=cl(taps["emb"]), attn0=cl(taps["attn0"]), layer0=cl(taps["layer0"]),
                        layer5=cl(taps["layer%d" % (fcfg.n_layers - 1)]), oracle_max_abs_dev=err)
    print("%-28s z rms %.3f  oracle-vs-reference max dev %.2e  (%d KB)" % (
        name, z.pow(2).mean().sqrt().item(), err, os.path.getsize(os.path.join(GOLD, name + ".npz")) // 1024))


def infer_full_case(name: str, T: int, skip_head=None, return_length=None, return_length2=None, seed: int = 1234):
    """The reference's whole ``net_g.infer`` (synthesizers.py:160-203) on seeded enc_p / flow / dec weights, RNG seeded with
    the reference's own 114514; the three draws (z_p noise, rand_ini, generator noise) are re-derived and stored."""
    cfg, fcfg = CONFIGS["v2_48k"], FrontConfig()
    wd, wf = synth.make_dec_weights(cfg, seed), synth.make_front_weights(fcfg, seed)
    net = build_reference_net(cfg, fcfg, wd, wf)
    phone = synth.make_phone(1, T, 768, seed)
    pitchf = synth.make_f0(1, T)
    pitch = synth.make_pitch(pitchf)
    lengths, sid = torch.tensor([T]), torch.tensor([3])
    torch.manual_seed(114514)
    with torch.no_grad():
        o = net.infer(phone, lengths, sid, pitch, pitchf, skip_head, return_length, return_length2)
    fh = 0 if skip_head is None else max(int(skip_head) - 24, 0)
    Td = T if skip_head is None else int(return_length)
    Te = Td if return_length2 is None else int(return_length2)
    gen = torch.Generator().manual_seed(114514)
    nz_zp = torch.randn(1, 192, T - fh, generator=gen)
    torch.rand(1, 1, 1, generator=gen)
    nz_dec = torch.randn(1, Td * cfg.upp, 1, generator=gen).squeeze(-1)  # drawn at the decoder's T, before the n_res interpolation (nsf.py:155-162)
    with torch.no_grad():  # oracle front + oracle generator must reproduce the reference end to end
        z, m1, g = front_oracle.infer_front(fcfg, wf, phone, pitch, lengths, sid, nz_zp, fh if skip_head is not None else None)
        z = z * m1
        pf = pitchf
        if skip_head is not None:
            dh = int(skip_head) - fh
            z = z[:, :, dh:dh + Td]
            pf = pitchf[:, int(skip_head):int(skip_head) + Td]
        ora = nsf_oracle.generator_forward(cfg, wd, z, pf, g, nz_dec, n_res=return_length2)
    err = (ora - o).abs().max().item()
    assert err < 2e-5, "%s: oracle deviates from the reference infer by %g" % (name, err)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), seed=seed, phone=phone.numpy(), pitch=pitch.numpy(), pitchf=pitchf.numpy(),
                        sid=sid.numpy(), noise_zp=nz_zp.numpy(), noise_dec=nz_dec.numpy(), out=o.numpy(),
                        skip_head=-1 if skip_head is None else int(skip_head), return_length=-1 if return_length is None else int(return_length),
                        return_length2=-1 if return_length2 is None else int(return_length2), oracle_max_abs_dev=err,
                        front_sha256=synth.weights_sha256(wf), dec_sha256=synth.weights_sha256(wd))
    print("%-28s out rms %.3f  oracle-vs-reference max dev %.2e  (%d KB)" % (
        name, o.pow(2).mean().sqrt().item(), err, os.path.getsize(os.path.join(GOLD, name + ".npz")) // 1024))


def front_full_case(name: str = "bigfront_v2_B1_T1198_z", T: int = 1198, seed: int = 1234):
    """enc_p + z_p + flow^-1 of the REFERENCE modules at the benchmark size (one 10 s clip: T = 1198, global attention over all
    frames): only ``z * mask`` is stored (0.9 MB); phone / pitch / noise are regenerated from the seeds by the test."""
    cfg, fcfg = CONFIGS["v2_48k"], FrontConfig()
    wf = synth.make_front_weights(fcfg, seed)
    net = build_reference_net(cfg, fcfg, synth.make_dec_weights(cfg, seed), wf)
    phone = synth.make_phone(1, T, 768, seed)
    pitch = synth.make_pitch(synth.make_f0(1, T))
    lengths, sid = torch.tensor([T]), torch.tensor([3])
    with torch.no_grad():
        g = net.emb_g(sid).unsqueeze(-1)
        m_p, logs_p, x_mask = net.enc_p(phone, pitch, lengths, None)
        noise = torch.randn(m_p.shape, generator=torch.Generator().manual_seed(seed + 9))
        z = net.flow((m_p + torch.exp(logs_p) * noise * 0.66666) * x_mask, x_mask, g=g, reverse=True)
        z2, m1, _ = front_oracle.infer_front(fcfg, wf, phone, pitch, lengths, sid, noise, None, {})
    err = (z2 - z).abs().max().item()
    assert err < 5e-5, "%s: front oracle deviates from the reference by %g" % (name, err)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), seed=seed, T=T, sid=sid.numpy(), weights_sha256=synth.weights_sha256(wf),
                        z=(z * x_mask).numpy(), oracle_max_abs_dev=err)
    print("%-28s z rms %.3f  oracle-vs-reference max dev %.2e  (%d KB)" % (
        name, z.pow(2).mean().sqrt().item(), err, os.path.getsize(os.path.join(GOLD, name + ".npz")) // 1024))


class NumpyFaissIndex:
    """The faiss index OBJECT as the reference's pipeline uses it (pipeline.py:214-215,126: ``ntotal``, ``reconstruct_n``,
    ``search``), answered by the CPU oracle (oracle/ivf_oracle.py) on a file in the IwFl layout.  faiss itself is not
    installable offline: what this pins is every line of the reference AROUND the faiss calls, not faiss' arithmetic."""

    def __init__(self, path):
        from oracle import ivf_oracle

        self._ix = ivf_oracle.read_index(path)
        self.ntotal = int(self._ix["ids"].shape[0])
        self.searches = 0

    def reconstruct_n(self, i0, n):
        from oracle import ivf_oracle

        return ivf_oracle.reconstruct_n(self._ix, i0, n)

    def search(self, x, k):
        from oracle import ivf_oracle

        assert x.dtype == np.float32 and x.flags.c_contiguous
        self.searches += 1
        return ivf_oracle.search(self._ix, x, k)


def import_reference_pipeline():
    """The REAL ``infer.modules.vc.pipeline`` of the reference.  Its module-level imports of faiss / librosa and numba's ``jit``
    (rvc/f0/gen.py) are not installable offline; stand-ins carry exactly the four names the exercised lines call:
    ``faiss.read_index`` -> ``NumpyFaissIndex`` (CPU oracle), ``librosa.feature.rms`` -> ``glue_oracle.frame_rms`` (restatement
    of librosa's documented behaviour, unpinned), ``librosa.filters.mel`` / ``librosa.util.pad_center`` (import-time names of
    rvc/f0/mel.py and stft.py, never called: the mel front end of the fake RMVPE is a stand-in), ``numba.jit`` -> identity."""
    import types

    from oracle import glue_oracle

    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)
    fa = types.ModuleType("faiss")
    fa.read_index = NumpyFaissIndex
    lb = types.ModuleType("librosa")
    lb.__path__ = []
    for sub, names in (("filters", ["mel"]), ("util", ["pad_center"]), ("feature", [])):
        m = types.ModuleType("librosa." + sub)
        for n in names:
            setattr(m, n, None)
        setattr(lb, sub, m)
        sys.modules["librosa." + sub] = m
    lb.feature.rms = lambda y, frame_length, hop_length: glue_oracle.frame_rms(y, frame_length, hop_length)[None]
    sys.modules.update(numba=nb, faiss=fa, librosa=lb)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.environ.setdefault("rmvpe_root", "/nonexistent")
    import importlib.util

    # by file: the package's __init__ would pull in the WebUI's audio I/O stack (av, ffmpeg), which this path never touches
    spec = importlib.util.spec_from_file_location("rvc_reference_pipeline", os.path.join(REF, "infer", "modules", "vc", "pipeline.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


PIPE_CFG = dict(x_pad=1, x_query=1, x_center=1, x_max=1, is_half=False)


def pipeline_case(name: str = "pipeline_v2_48k_3seg", n_audio: int = 38400, seed: int = 1234, protect: float = 0.33):
    """The reference's own ``Pipeline.pipeline`` (pipeline.py:186-360) -- high-pass, cut points, three ``Pipeline.vc`` calls
    (x2 interpolation, protect mix, ``net_g.infer`` of the reference synthesizer with seeded weights), concatenation, int16-range
    scaling -- with a seeded stand-in for HuBERT (oracle/synth.py FakeHubert), a precomputed f0 track (if_f0 = 2) and no
    index.  Stored: the returned waveform, the per-call frame counts, the sha256 of the injected noise and of the weights."""
    import hashlib
    import types

    pl = import_reference_pipeline()
    cfg, fcfg = CONFIGS["v2_48k"], FrontConfig()
    wd, wf = synth.make_dec_weights(cfg, seed), synth.make_front_weights(fcfg, seed)
    net = build_reference_net(cfg, fcfg, wd, wf)
    config = types.SimpleNamespace(device=torch.device("cpu"), **PIPE_CFG)
    pipe = pl.Pipeline(cfg.sr, config)
    audio = synth.make_audio16k(n_audio, seed)
    p_len_all = (n_audio + 2 * pipe.t_pad) // pipe.window
    pitchf = synth.make_f0(1, p_len_all)[0]
    pitch = synth.make_pitch(pitchf)
    lens, raw = [], []
    orig_infer, orig_vc = net.infer, pl.Pipeline.vc

    def infer_spy(phone, lengths, *a, **k):
        lens.append(int(phone.shape[1]))
        return orig_infer(phone, lengths, *a, **k)

    def vc_spy(self, *a, **k):
        o = orig_vc(self, *a, **k)
        raw.append(o.copy())
        return o

    net.infer = infer_spy
    pl.Pipeline.vc = vc_spy
    try:
        torch.manual_seed(114514)
        times = [0, 0, 0]
        out = pipe.pipeline(synth.FakeHubert(768, seed), net, 3, audio.copy(), times, 0, (pitch.numpy(), pitchf.numpy()), "", 0.75, 2, 3,
                            cfg.sr, 0, 1, "v2", protect)
    finally:
        pl.Pipeline.vc = orig_vc
    assert len(lens) == 3, lens
    h = hashlib.sha256()
    for nz, nd in synth.infer_noise(lens, cfg.upp):
        h.update(nz.numpy().tobytes())
        h.update(nd.numpy().tobytes())
    cat = np.concatenate([r[pipe.t_pad_tgt:-pipe.t_pad_tgt] for r in raw])
    scale = 32768.0 / max(1.0, float(np.abs(cat).max()) / 0.99)
    assert np.allclose(cat * scale, out, rtol=1e-6, atol=1e-3)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), seed=seed, n_audio=n_audio, protect=protect, sid=3, out=out.astype(np.float32),
                        seg_frames=np.array(lens), seg0_raw_len=raw[0].shape[0], scale=scale, noise_sha256=h.hexdigest(),
                        front_sha256=synth.weights_sha256(wf), dec_sha256=synth.weights_sha256(wd),
                        **{"cfg_" + k: v for k, v in PIPE_CFG.items()})
    print("%-28s out rms %.1f (int16 range), segments %s, scale %.1f  (%d KB)" % (
        name, float(np.sqrt(np.mean(out.astype(np.float64) ** 2))), lens, scale, os.path.getsize(os.path.join(GOLD, name + ".npz")) // 1024))


def reference_rmvpe(seed: int):
    """A REAL ``rvc.f0.rmvpe.RMVPE`` object without its constructor (which loads the checkpoint and builds librosa's mel filter
    bank): ``compute_f0``, ``_mel2hidden``, ``_decode``, ``_to_local_average_cents`` and the ``F0Predictor`` resize /
    interpolate methods are the reference's own; only the mel front end and the network are the seeded stand-ins of
    oracle/synth.py."""
    from rvc.f0.rmvpe import RMVPE

    r = RMVPE.__new__(RMVPE)
    r.hop_length, r.f0_min, r.f0_max, r.sampling_rate = 160, 30, 8000, 16000
    r.device, r.is_half = torch.device("cpu"), False
    r.cents_mapping = np.pad(20 * np.arange(360) + 1997.3794084376191, (4, 4))  # rmvpe.py:62-63
    r.mel_extractor, r.model = synth.FakeMel(), synth.FakeRMVPEModel(seed)
    return r


WEBUI = dict(f0_method="rmvpe", index_rate=0.75, filter_radius=3, resample_sr=0, rms_mix_rate=0.25, protect=0.33, f0_up_key=0)
WEBUI_INDEX = dict(n=3000, d=768, seed=11)


def pipeline_webui_case(name: str = "pipeline_v2_48k_webui", n_audio: int = 38400, seed: int = 1234, f0_up_key: int = 0):
    """The reference's own ``Pipeline.pipeline`` with the WebUI's single-inference defaults (web.py:756-802: rmvpe,
    index_rate 0.75 WITH an index file, filter_radius 3, rms_mix_rate 0.25, protect 0.33, if_f0 1): ``faiss.read_index`` +
    ``reconstruct_n`` (pipeline.py:214-215), ``Generator.calculate`` -> ``RMVPE.compute_f0`` -> ``post_process`` (gen.py:103-137),
    three ``Pipeline.vc`` calls WITH the retrieval branch (pipeline.py:113-138), ``change_rms`` (pipeline.py:26-46), int16 scaling.
    Stand-ins: HuBERT, the RMVPE mel + network, the faiss object (CPU oracle), ``librosa.feature.rms`` (restatement)."""
    import hashlib
    import tempfile
    import types

    from oracle import ivf_oracle

    pl = import_reference_pipeline()
    cfg, fcfg = CONFIGS["v2_48k"], FrontConfig()
    wd, wf = synth.make_dec_weights(cfg, seed), synth.make_front_weights(fcfg, seed)
    net = build_reference_net(cfg, fcfg, wd, wf)
    config = types.SimpleNamespace(device=torch.device("cpu"), **PIPE_CFG)
    pipe = pl.Pipeline(cfg.sr, config)
    pipe.f0_gen.rmvpe = reference_rmvpe(seed)
    audio = synth.make_audio16k(n_audio, seed)
    idx = synth.make_ivf(WEBUI_INDEX["n"], WEBUI_INDEX["d"], seed=WEBUI_INDEX["seed"])
    lens, raw, pitches, found = [], [], [], []
    orig_infer, orig_vc, orig_read = net.infer, pl.Pipeline.vc, pl.faiss.read_index

    def infer_spy(phone, lengths, *a, **k):
        lens.append(int(phone.shape[1]))
        return orig_infer(phone, lengths, *a, **k)

    def vc_spy(self, model, net_g, sid, audio0, pitch, pitchf, times, index, big_npy, *a, **k):
        assert isinstance(index, NumpyFaissIndex) and big_npy.shape == (WEBUI_INDEX["n"], WEBUI_INDEX["d"])
        found.append(index)
        pitches.append((pitch[0].numpy().copy(), pitchf[0].numpy().copy()))
        o = orig_vc(self, model, net_g, sid, audio0, pitch, pitchf, times, index, big_npy, *a, **k)
        raw.append(o.copy())
        return o

    net.infer = infer_spy
    pl.Pipeline.vc = vc_spy
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "added.index")
        ivf_oracle.write_index(idx, path)
        try:
            torch.manual_seed(114514)
            times = [0, 0, 0]
            out = pipe.pipeline(synth.FakeHubert(768, seed), net, 3, audio.copy(), times, f0_up_key, WEBUI["f0_method"], path,
                                WEBUI["index_rate"], 1, WEBUI["filter_radius"], cfg.sr, WEBUI["resample_sr"], WEBUI["rms_mix_rate"],
                                "v2", WEBUI["protect"])
        finally:
            pl.Pipeline.vc = orig_vc
    assert len(lens) == 3 and found[0].searches == 3 and pipe.f0_gen.rmvpe.model.calls == 1, (lens, found[0].searches)
    h = hashlib.sha256()
    for nz, nd in synth.infer_noise(lens, cfg.upp):
        h.update(nz.numpy().tobytes())
        h.update(nd.numpy().tobytes())
    # the f0 track the reference computed for the whole padded input (its three vc calls see overlapping slices of it)
    p_len_all = (n_audio + 2 * pipe.t_pad) // pipe.window
    sal = pipe.f0_gen.rmvpe._mel2hidden(synth.FakeMel()(torch.zeros(1, n_audio + 2 * pipe.t_pad))).squeeze(0).numpy()
    from oracle import glue_oracle

    o_pitch, o_pitchf = glue_oracle.rmvpe_f0(sal, p_len_all, f0_up_key, 0.03)
    assert np.array_equal(pitches[0][0], o_pitch[: len(pitches[0][0])]) and np.array_equal(pitches[0][1], o_pitchf[: len(pitches[0][1])])
    # what the oracle restatements make of the same three raw segments: change_rms + scaling (pins the glue oracle end to end)
    cat = np.concatenate([r[pipe.t_pad_tgt:-pipe.t_pad_tgt] for r in raw])
    from scipy import signal as sg

    a_hp = sg.filtfilt(pl.bh, pl.ah, audio)
    ora = glue_oracle.scale_int16_range(glue_oracle.change_rms(a_hp, 16000, cat, cfg.sr, WEBUI["rms_mix_rate"]))
    assert np.allclose(ora, out, rtol=1e-6, atol=1e-3)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), seed=seed, n_audio=n_audio, sid=3, out=out.astype(np.float32),
                        seg_frames=np.array(lens), pitch=o_pitch, pitchf=o_pitchf, noise_sha256=h.hexdigest(),
                        index_n=WEBUI_INDEX["n"], index_d=WEBUI_INDEX["d"], index_seed=WEBUI_INDEX["seed"],
                        index_rate=WEBUI["index_rate"], filter_radius=WEBUI["filter_radius"], rms_mix_rate=WEBUI["rms_mix_rate"],
                        protect=WEBUI["protect"], f0_up_key=f0_up_key, voiced_frames=int((o_pitchf > 0).sum()),
                        front_sha256=synth.weights_sha256(wf), dec_sha256=synth.weights_sha256(wd),
                        **{"cfg_" + k: v for k, v in PIPE_CFG.items()})
    print("%-28s out rms %.1f (int16 range), segments %s, %d of %d frames voiced  (%d KB)" % (
        name, float(np.sqrt(np.mean(out.astype(np.float64) ** 2))), lens, int((o_pitchf > 0).sum()), p_len_all,
        os.path.getsize(os.path.join(GOLD, name + ".npz")) // 1024))


def glue_case(name: str = "glue_f0"):
    """The f0 chain of the REFERENCE itself -- RMVPE._to_local_average_cents/_decode (rvc/f0/rmvpe.py:119-164),
    F0Predictor._resize_f0/_interpolate_f0 (rvc/f0/f0.py:31-78), post_process (rvc/f0/gen.py:10-41, numba stubbed to a
    pass-through) -- on seeded salience maps with voiced/unvoiced runs, edge layouts included.  The oracle restatement is
    checked against them, inputs and outputs are stored."""
    import types

    sys.path.insert(0, REF)
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)
    sys.modules.setdefault("numba", nb)
    from math import log

    from rvc.f0.f0 import F0Predictor
    from rvc.f0.gen import post_process

    src = open(os.path.join(REF, "rvc", "f0", "rmvpe.py")).read()
    a, b = src.index("    def _to_local_average_cents"), src.index("    def _mel2hidden")
    c = src.index("    def _decode")
    ns = {"np": np}
    exec("class R:\n" + src[a:b] + src[c:], ns)  # the two methods verbatim, without the model-loading constructor
    R = ns["R"]
    rm = R()
    rm.cents_mapping = np.pad(20 * np.arange(360) + 1997.3794084376191, (4, 4))  # rmvpe.py:62-63
    from oracle import glue_oracle

    out = {}
    rng = np.random.default_rng(2024)
    cases = {
        "mixed": (120, 130, 0), "up": (75, 150, 7), "down": (129, 65, -5), "lead_unvoiced": (48, 48, 12),
        "tail_unvoiced": (60, 66, 0), "last_voiced_only_end": (40, 40, 3), "all_unvoiced": (33, 40, 0), "single": (1, 3, 0),
    }
    fp = F0Predictor()
    for cname, (n, p_len, key) in cases.items():
        sal = (rng.random((n, 360), dtype=np.float32) * 0.02).astype(np.float32)
        voiced = np.ones(n, bool)
        if cname == "mixed":
            voiced[(np.arange(n) % 40) < 11] = False
        elif cname == "lead_unvoiced":
            voiced[:20] = False
        elif cname == "tail_unvoiced":
            voiced[-25:] = False
        elif cname == "last_voiced_only_end":
            voiced[10:-1] = False
        elif cname == "all_unvoiced":
            voiced[:] = False
        elif cname in ("up", "down"):
            voiced[(np.arange(n) % 37) < 9] = False
        centre = (120 + 60 * np.sin(np.arange(n) / 17.0)).astype(int)
        centre[::29] = np.array([0, 1, 358, 359, 3, 356])[np.arange(len(centre[::29])) % 6]  # window clipped at both table ends
        for i in range(n):
            if voiced[i]:
                k0 = centre[i]
                for w_ in range(-6, 7):
                    if 0 <= k0 + w_ < 360:
                        sal[i, k0 + w_] += np.float32(0.9 * math.exp(-0.5 * (w_ / 2.0) ** 2))
        f0a = rm._decode(sal.copy(), thred=0.03)
        f0r = fp._interpolate_f0(fp._resize_f0(f0a, p_len))[0]
        coarse, f0k = post_process(100, f0r.copy(), key, 1, 1127 * log(1 + 50 / 700), 1127 * log(1 + 1100 / 700), None)
        pitch, pitchf = coarse[:p_len].astype(np.int64), f0k[:p_len].astype(np.float32)
        o_dec = glue_oracle.rmvpe_decode(sal, 0.03)
        o_pitch, o_pitchf = glue_oracle.rmvpe_f0(sal, p_len, key, 0.03)
        assert np.array_equal(o_dec, f0a), cname
        assert np.array_equal(o_pitch, pitch) and np.array_equal(o_pitchf, pitchf), cname
        out[cname + "::salience"] = sal
        out[cname + "::meta"] = np.array([n, p_len, key])
        out[cname + "::f0_decoded"] = f0a
        out[cname + "::pitch"] = pitch
        out[cname + "::pitchf"] = pitchf
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("%-28s %d cases, oracle == reference bit-exact  (%d KB)" % (name, len(cases), os.path.getsize(os.path.join(GOLD, name + ".npz")) // 1024))


def mute_case(name: str = "mute_hubert"):
    """The only REAL HuBERT features in the reference checkout: logs/mute/3_feature{256,768}/mute.npy (149 frames of the
    'mute' training clip).  Near-silence features are almost collinear and contain exact duplicate
    rows -- the worst case for distance ties -- so retrieval tests build their index / queries from this distribution."""
    out = {}
    for dim in (256, 768):
        a = np.load(os.path.join(REF, "logs", "mute", "3_feature%d" % dim, "mute.npy"))
        assert a.shape == (149, dim) and a.dtype == np.float32
        out["f%d" % dim] = a
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("%-28s 149 real HuBERT rows x {256, 768}  (%d KB)" % (name, os.path.getsize(os.path.join(GOLD, name + ".npz")) // 1024))


def main_front():
    glue_case()
    front_case("front_v2_B2_T50", 2, 50, [50, 43])
    front_case("front_v2_B1_T100_head6", 1, 100, [100], flow_head=6)     # > one 64-row tile, realtime flow_head
    front_case("front_v1_B1_T40", 1, 40, [40], in_channels=256)           # v1: 256-d features
    infer_full_case("infer_full_v2_48k_T40", 40)
    infer_full_case("infer_full_v2_48k_rt", 70, skip_head=40, return_length=20, return_length2=24)  # rtrvc.py:134-260 geometry


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    if os.environ.get("GOLDEN_ONLY_FULL"):
        return dec_full_case()
    if os.environ.get("GOLDEN_ONLY_MUTE"):
        return mute_case()
    if os.environ.get("GOLDEN_ONLY_WEBUI"):
        return pipeline_webui_case()
    if os.environ.get("GOLDEN_ONLY_PIPELINE"):
        pipeline_case()
        pipeline_webui_case()
        return front_full_case()
    mute_case()
    main_front()
    if os.environ.get("GOLDEN_ONLY_FRONT"):
        return
    dec_case("dec_v2_48k_B2_T24", "v2_48k", 2, 24)
    dec_case("dec_v2_48k_B1_T70", "v2_48k", 1, 70)           # > one fused-resblock tile at every stage
    dec_case("dec_v2_32k_B1_T16", "v2_32k", 1, 16)
    dec_case("dec_v1_40k_B1_T20", "v1_40k", 1, 20)
    dec_case("dec_v1_32k_B1_T16", "v1_32k", 1, 16)           # 5 stages, last stage C = 16
    dec_case("dec_v1_48k_B1_T12", "v1_48k", 1, 12)
    dec_case("dec_nof0_v2_48k_B1_T16", "v2_48k", 1, 16, use_f0=False)
    dec_case("dec_v1_40k_nres_T31", "v1_40k", 1, 31, n_res=37)  # realtime formant shift
    dec_case("dec_v1_40k_nres_down_T31", "v1_40k", 1, 31, n_res=26)
    infer_case()
    dec_full_case()
    pipeline_case()
    pipeline_webui_case()
    front_full_case()


if __name__ == "__main__":
    main()
