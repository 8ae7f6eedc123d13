"""The bidirectional GRU of the RMVPE network (rvc/f0/e2e.py:50-67) on csrc/gru.hip -- beyond the SURVEY.md section 8 scope table -- against
torch's own ``nn.GRU`` in fp32 on the CPU (the plain PyTorch fp32 reference of the same op).  Tolerance: fp16 operands (x, W_ih, W_hh, the
broadcast copy of h) with fp32 accumulation and state: <= 2e-3 RMS / 1e-2 max-abs on outputs in (-1, 1)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _pair(seed, input_size=384, gain=1.0):
    torch.manual_seed(seed)
    ref = torch.nn.GRU(input_size, 256, num_layers=1, batch_first=True, bidirectional=True).eval()
    if gain != 1.0:
        with torch.no_grad():
            for p in ref.parameters():
                p.mul_(gain)
    return ref


@pytest.mark.parametrize("B,T,seed,gain", [(1, 1, 0, 1.0), (2, 7, 1, 1.0), (1, 1216, 2, 1.0), (3, 301, 3, 2.5), (64, 40, 4, 1.0)])
def test_gru_hip_matches_torch_fp32(B, T, seed, gain, gpu):
    import rvc_amd

    ref = _pair(seed, gain=gain)
    x = torch.randn(B, T, 384, generator=torch.Generator().manual_seed(100 + seed))
    with torch.no_grad():
        y_ref, hn_ref = ref(x)
    m = rvc_amd.GRUHIP(ref, device=gpu)
    y, hn = m(x.to(gpu))
    torch.cuda.synchronize()
    assert y.shape == y_ref.shape and hn.shape == hn_ref.shape and y.dtype == torch.float32
    e = (y.cpu() - y_ref)
    rms, mx = float(e.pow(2).mean().sqrt()), float(e.abs().max())
    assert torch.isfinite(y).all() and rms <= 2e-3 and mx <= 1e-2, "B=%d T=%d gain %.1f: RMS %.3e max %.3e" % (B, T, gain, rms, mx)
    assert float((hn.cpu() - hn_ref).abs().max()) <= 1e-2
    # h_n is the last forward / first backward output row
    assert torch.equal(hn[0], y[:, -1, :256]) and torch.equal(hn[1], y[:, 0, 256:])
    # deterministic
    y2, _ = m(x.to(gpu))
    assert torch.equal(y, y2)


def test_gru_hip_half_input_and_other_input_size(gpu):
    import rvc_amd

    ref = _pair(7, input_size=64)
    x = torch.randn(2, 50, 64, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        y_ref, _ = ref(x)
    y, hn = rvc_amd.GRUHIP(ref, device=gpu)(x.to(gpu).half())
    assert y.dtype == torch.float16 and hn.dtype == torch.float16
    assert float((y.float().cpu() - y_ref).abs().max()) <= 1e-2


def test_accelerate_rmvpe_swaps_the_gru_of_the_proxy_network_only(gpu):
    """``accelerate_rmvpe`` on the architecture proxy of rvc/f0/e2e.py (tools/e2e_proxies.py): one module replaced, the salience within the
    fp16 tolerance of the all-PyTorch network; an unsupported GRU (hidden 128) is left alone; a CPU model is left alone."""
    import rvc_amd
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from e2e_proxies import _SalienceNet

    torch.manual_seed(3)
    net = _SalienceNet().eval().to(gpu)
    mel = torch.randn(1, 128, 64, device=gpu)
    with torch.no_grad():
        want = net(mel)
        assert rvc_amd.accelerate_rmvpe(net) == 1 and isinstance(net.gru, rvc_amd.GRUHIP)
        got = net(mel)
    assert got.shape == want.shape and float((got - want).abs().max()) <= 5e-3
    small = torch.nn.Sequential(torch.nn.GRU(64, 128, batch_first=True, bidirectional=True)).to(gpu)
    assert rvc_amd.accelerate_rmvpe(small) == 0 and isinstance(small[0], torch.nn.GRU)
    cpu_net = torch.nn.Sequential(torch.nn.GRU(64, 256, batch_first=True, bidirectional=True))
    assert rvc_amd.accelerate_rmvpe(cpu_net) == 0


def test_gru_hip_rejects_what_it_does_not_implement(gpu):
    import rvc_amd

    with pytest.raises(rvc_amd.RvcmiError):
        rvc_amd.GRUHIP(torch.nn.GRU(384, 256, num_layers=2, batch_first=True, bidirectional=True), device=gpu)
    m = rvc_amd.GRUHIP(_pair(0), device=gpu)
    with pytest.raises(rvc_amd.RvcmiError):
        m(torch.zeros(1, 4, 384))  # CPU input: no fallback
    with pytest.raises(ValueError):
        m(torch.zeros(1, 4, 100, device=gpu))


def test_rebound_pipeline_swaps_the_gru_once_and_honours_the_opt_out(gpu, monkeypatch):
    """``pipeline._rmvpe_on_device`` (the f0 step of the rebound ``Pipeline.pipeline``): the first call swaps the network's GRU for the HIP one,
    ``RVCMI_RMVPE_GRU=0`` leaves torch's in place; both produce a pitch track of the asked length."""
    import types

    import numpy as np
    import rvc_amd
    import rvc_amd.pipeline as rp

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from e2e_proxies import RmvpeProxy

    audio = (0.3 * np.sin(2 * np.pi * 220.0 * np.arange(16000 * 2) / 16000.0)).astype(np.float32)
    p_len = audio.shape[0] // 160
    for env, want in (("0", 0), ("1", 1)):
        monkeypatch.setenv("RVCMI_RMVPE_GRU", env)
        r = RmvpeProxy(gpu, half=True)
        me = types.SimpleNamespace(f0_gen=types.SimpleNamespace(rmvpe=r, is_half=True, device=gpu))
        pitch, pitchf = rp._rmvpe_on_device(me, audio, p_len, 0)
        assert r._rvcmi_gru == want and isinstance(r.model.gru, rvc_amd.GRUHIP) == bool(want)
        assert pitch.shape[-1] == p_len and pitchf.shape[-1] == p_len and torch.isfinite(pitchf).all()
        rp._rmvpe_on_device(me, audio, p_len, 0)  # (second call: nothing left to swap)
        assert r._rvcmi_gru == want


def test_accelerate_f0_rmvpe_is_idempotent_and_tolerates_foreign_objects(gpu, monkeypatch):
    """The helper both rebound callers use (``Pipeline.pipeline`` and the realtime ``RVC.infer``): swaps once, remembers the count, leaves an
    object without a torch network alone."""
    import types

    from rvc_amd.gru import accelerate_f0_rmvpe

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from e2e_proxies import RmvpeProxy

    monkeypatch.delenv("RVCMI_RMVPE_GRU", raising=False)
    r = RmvpeProxy(gpu, half=False)
    assert accelerate_f0_rmvpe(r) == 1 and accelerate_f0_rmvpe(r) == 1 and r._rvcmi_gru == 1
    onnx_like = types.SimpleNamespace(model=object())
    assert accelerate_f0_rmvpe(onnx_like) == 0 and onnx_like._rvcmi_gru == 0


def test_realtime_f0_chain_replayed_from_a_hipgraph_equals_the_eager_chain(gpu, monkeypatch):
    """``realtime._rmvpe_f0_graphed``: the fixed-length f0 window of the realtime loop (mel STFT -> the RMVPE architecture proxy incl. the HIP
    GRU -> salience decode) is captured after three eager blocks and replayed; every replay must give what the eager chain gives for the same
    waveform (same kernels; to the network's own run-to-run noise).  ``RVCMI_RT_GRAPH=0`` never captures."""
    import types

    import rvc_amd.pipeline as rp
    from rvc_amd import realtime as rt

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from e2e_proxies import RmvpeProxy

    monkeypatch.delenv("RVCMI_RMVPE_GRU", raising=False)
    monkeypatch.delenv("RVCMI_RT_GRAPH", raising=False)
    n = rt.f0_extractor_frame(4096, "rmvpe", 160)
    p_len = n // 160
    me = types.SimpleNamespace(f0_gen=types.SimpleNamespace(rmvpe=RmvpeProxy(gpu, half=True), is_half=True, device=gpu))
    g = torch.Generator().manual_seed(1)
    for i in range(7):
        wav = (0.2 * torch.randn(n, generator=g)).to(gpu)
        pitch, pitchf = rt._rmvpe_f0_graphed(me, wav, p_len, 0)
        pitch, pitchf = pitch.clone(), pitchf.clone()
        entry = next(iter(me._rvcmi_f0_graphs.values()))
        assert ("graph" in entry) == (i >= rt.RT_GRAPH_AFTER - 1), (i, list(entry))
        want = rp._rmvpe_on_device(me, wav, p_len, 0)
        # (two EAGER runs of this network already differ in the last bits -- MIOpen's convolutions are not run-to-run deterministic -- so: to 1e-3)
        assert pitch.shape == (1, p_len) and int((pitch - want[0]).abs().max()) <= 1, "block %d: coarse pitch differs" % i
        assert torch.allclose(pitchf, want[1], rtol=1e-3, atol=0) and float(pitchf.max()) > 0, "block %d: replay differs from the eager chain" % i
    assert len(me._rvcmi_f0_graphs) == 1

    # the replay must follow its INPUT (the proxy's random weights make a nearly input-independent salience): the same mel front end in front of a
    # parameter-free 'network' whose salience peak moves with the signal's level
    class Tiny(torch.nn.Module):
        def forward(self, mel):  # [1, 128, T] log-mel -> [1, T, 360]: a salience bump whose bin follows the frame's level
            c = 180.0 + 25.0 * mel.float().mean(dim=1)
            bins = torch.arange(360, device=mel.device, dtype=torch.float32)
            return torch.exp(-0.5 * ((bins[None, None, :] - c[..., None]) / 3.0) ** 2).to(mel.dtype)

    tiny = RmvpeProxy(gpu, half=False)
    tiny.model = Tiny().to(gpu)
    me3 = types.SimpleNamespace(f0_gen=types.SimpleNamespace(rmvpe=tiny, is_half=False, device=gpu))
    outs = []
    for i in range(6):
        t = torch.arange(n, device=gpu) / 16000.0
        wav = 0.05 * (i + 1) * torch.sin(2 * torch.pi * 220.0 * t)
        pitch, pitchf = rt._rmvpe_f0_graphed(me3, wav, p_len, 0)
        want = rp._rmvpe_on_device(me3, wav, p_len, 0)
        assert torch.allclose(pitchf, want[1], rtol=1e-4, atol=0) and int((pitch - want[0]).abs().max()) <= 1, "tiny network, block %d" % i
        outs.append(pitchf.clone())
    assert "graph" in next(iter(me3._rvcmi_f0_graphs.values()))
    assert not torch.allclose(outs[-1], outs[-2], rtol=1e-2), "two replays with different inputs gave the same f0: the static input is not read"
    monkeypatch.setenv("RVCMI_RT_GRAPH", "0")
    me2 = types.SimpleNamespace(f0_gen=types.SimpleNamespace(rmvpe=RmvpeProxy(gpu, half=True), is_half=True, device=gpu))
    for i in range(5):
        rt._rmvpe_f0_graphed(me2, wav, p_len, 0)
    assert not getattr(me2, "_rvcmi_f0_graphs", None)
