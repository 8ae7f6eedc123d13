"""GPU parity of the HIP generator (through the Python mirror -> ctypes -> C ABI of include/rvcmi.h) against
(a) golden fixtures produced by the REAL reference modules and (b) the oracle at other sizes, plus size-independent
properties at BASELINE's full clip size.  Tolerances: the north star's bar is <= 1e-3 RMS on the waveform."""
import numpy as np
import pytest
import torch

from conftest import golden_config_and_weights, golden_names, load_golden, rms
from oracle import nsf_oracle, synth

pytestmark = pytest.mark.gpu

# RMS bars per operand type.  fp32: exact-arithmetic class.  fp16 (default): the north-star bar.  bf16 operands do NOT
# meet 1e-3 on these deliberately hard variance-preserving weights (2.7e-3 measured) -- documented in DESIGN.md; the
# test pins that it stays in its known class rather than silently regressing.
BAR = {"fp32": 2e-5, "fp16": 1e-3, "bf16": 6e-3}

_gens = {}


def hip_gen(cfg, w, operand, gpu):
    import rvc_amd

    key = (id(w), operand)
    if key not in _gens:
        cls = rvc_amd.NSFGeneratorHIP if cfg.use_f0 else rvc_amd.GeneratorHIP
        _gens.clear()  # one live handle at a time keeps the workspace small
        _gens[key] = cls(vars(cfg), w, device=gpu, operand=operand, max_B=2, max_T=80)
    return _gens[key]


def pin(gen, **opts):
    """Test options of ONE handle (rvcmi_nsf_set_option): e.g. RB_STREAM=1 forces the streaming ResBlock kernels, RS_SMALL picks
    their tile height; None restores the launcher's own choice (an unknown option key is an error, never silently ignored)."""
    for k, v in opts.items():
        gen.set_option(k, v)
    return gen


def run_golden(d, cfg, w, operand, gpu, **opts):
    gen = pin(hip_gen(cfg, w, operand, gpu), RB_STREAM=opts.get("RB_STREAM"), RS_SMALL=opts.get("RS_SMALL"))
    z = torch.from_numpy(d["z"]).to(gpu)
    g = torch.from_numpy(d["g"]).to(gpu)
    n_res = None if int(d.get("n_res", -1)) < 0 else int(d["n_res"])
    if cfg.use_f0:
        out = gen(z, torch.from_numpy(d["f0"]).to(gpu), g, n_res, noise=torch.from_numpy(d["noise"]).to(gpu))
    else:
        out = gen(z, g, n_res)
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("operand", ["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("name", golden_names("dec_"))
def test_generator_matches_reference_golden(name, operand, gpu):
    d = load_golden(name)
    cfg, w = golden_config_and_weights(d)
    out = run_golden(d, cfg, w, operand, gpu)
    assert out.shape == d["out"].shape
    assert torch.isfinite(out).all()
    e = rms(out, d["out"])
    assert e <= BAR[operand], "%s/%s: RMS error %.3e vs the reference exceeds %.1e" % (name, operand, e, BAR[operand])


@pytest.mark.parametrize("operand", ["fp32", "fp16"])
def test_generator_inside_the_reference_infer_flow(operand, gpu):
    """z*x_mask, pitchf, g as the reference's own enc_p/flow produced them inside net_g.infer, weights as its loader
    folded them from a legacy fp16 weight-norm checkpoint (fixture infer_v2_48k_T20)."""
    import rvc_amd

    d = load_golden("infer_v2_48k_T20")
    cfg = nsf_oracle.CONFIGS["v2_48k"]
    _, w = synth.make_legacy_checkpoint(cfg, "v2", int(d["seed"]))
    gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=gpu, operand=operand, max_B=1, max_T=32)
    out = gen(torch.from_numpy(d["z"]).to(gpu), torch.from_numpy(d["f0"]).to(gpu), torch.from_numpy(d["g"]).to(gpu),
              noise=torch.from_numpy(d["noise"]).to(gpu)).cpu()
    assert rms(out, d["out"]) <= BAR[operand]


def test_sine_source_and_stage_taps_fp32(gpu):
    """Per-layer parity (har, conv_pre, every ups+noise_conv, every resblock stage) of the exact-fp32 path."""
    import rvc_amd

    cfg = nsf_oracle.CONFIGS["v2_48k"]
    w = synth.make_dec_weights(cfg, 99)
    B, T = 2, 33
    z, f0, g = synth.make_dec_inputs(cfg, B, T, 99)
    noise = nsf_oracle.reference_noise(B, T, cfg.upp, 5)
    taps = {}
    with torch.no_grad():
        nsf_oracle.generator_forward(cfg, w, z, f0, g, noise, taps=taps)
    gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=gpu, operand="fp32", max_B=B, max_T=T)
    for k, v in taps.items():
        got = gen.debug_tap(k, z.to(gpu), f0.to(gpu), g.to(gpu), noise=noise.to(gpu))
        exp = v * cfg.num_kernels if k.startswith("stage") else v
        assert got.shape == exp.shape, k
        assert rms(got, exp) <= 3e-6 * max(1.0, float(exp.abs().max())), k


def test_mfma_stage_taps_fp16(gpu):
    import rvc_amd

    cfg = nsf_oracle.CONFIGS["v2_48k"]
    w = synth.make_dec_weights(cfg, 7)
    B, T = 1, 40
    z, f0, g = synth.make_dec_inputs(cfg, B, T, 7)
    noise = nsf_oracle.reference_noise(B, T, cfg.upp, 6)
    taps = {}
    with torch.no_grad():
        nsf_oracle.generator_forward(cfg, w, z, f0, g, noise, taps=taps)
    gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=gpu, operand="fp16", max_B=B, max_T=T)
    for k, v in taps.items():
        got = gen.debug_tap(k, z.to(gpu), f0.to(gpu), g.to(gpu), noise=noise.to(gpu))
        exp = v * cfg.num_kernels if k.startswith("stage") else v
        rel = rms(got, exp) / float(exp.pow(2).mean().sqrt())
        assert rel <= (1e-6 if k == "har" else 2e-3), "%s: relative RMS %.2e" % (k, rel)


def test_rng_stream_matches_torch_draw_order(gpu):
    """noise=None draws rand(1,1,1) then randn([B,T*upp,1]) on f0's device, like generators.py:164,192."""
    import rvc_amd

    cfg = nsf_oracle.CONFIGS["v1_40k"]
    w = synth.make_dec_weights(cfg, 3)
    B, T = 1, 12
    z, f0, g = synth.make_dec_inputs(cfg, B, T, 3)
    gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=gpu, operand="fp16", max_B=B, max_T=T)
    zd, fd, gd = z.to(gpu), f0.to(gpu), g.to(gpu)
    torch.manual_seed(1234)
    a = gen(zd, fd, gd)
    torch.manual_seed(1234)
    b = gen(zd, fd, gd)
    torch.manual_seed(1234)
    torch.rand(1, 1, 1, device=gpu)
    noise = torch.randn(B, T * cfg.upp, 1, device=gpu)
    c = gen(zd, fd, gd, noise=noise)
    assert torch.equal(a, b) and torch.equal(a, c)
    assert not torch.equal(a, gen(zd, fd, gd))  # the stream advanced


def test_handle_rejects_bad_shapes_and_grows_its_workspace(gpu):
    import rvc_amd

    cfg = nsf_oracle.CONFIGS["v1_40k"]
    w = synth.make_dec_weights(cfg, 3)
    gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=gpu, operand="fp16", max_B=1, max_T=8)
    z, f0, g = synth.make_dec_inputs(cfg, 2, 20, 3)
    with pytest.raises(ValueError):
        gen(z.to(gpu)[:, :100], f0.to(gpu), g.to(gpu))
    with pytest.raises(ValueError):
        gen(z.to(gpu), None, g.to(gpu))
    out = gen(z.to(gpu), f0.to(gpu), g.to(gpu), noise=torch.zeros(2, 20 * cfg.upp, device=gpu))  # beyond max_B/max_T: re-created
    assert out.shape == (2, 1, 20 * cfg.upp)
    bad = dict(w)
    bad.pop("ups.1.bias")
    with pytest.raises(rvc_amd.RvcmiError, match="missing weight"):
        rvc_amd.NSFGeneratorHIP(vars(cfg), bad, device=gpu)


@pytest.mark.parametrize("rb_stream", ["0", "1"])
def test_full_clip_size_properties(rb_stream, gpu):
    """(Both ResBlock kernel families: option RB_STREAM=0 the tile kernels, =1 the streaming kernel; a fixed choice, because
    the launcher otherwise picks per clip length and the two families differ in the last fp32 bit of the residual add.)
    BASELINE size (v2/48k, T = 1198 frames = one 10 s clip): determinism, batch independence and locality.
    Locality is the size-independent property of a conv stack: the waveform of frames [a, b) computed from the
    whole clip equals the one computed from a window with a receptive-field margin -- across completely different
    tile boundaries, phase offsets and grid sizes -- so it exercises every halo / tiling decision at full scale."""
    import rvc_amd

    cfg = nsf_oracle.CONFIGS["v2_48k"]
    w = synth.make_dec_weights(cfg, 1234)
    T = 1198
    z, f0, g = synth.make_dec_inputs(cfg, 1, T, 1234)
    f0 = torch.zeros_like(f0)  # unvoiced everywhere: the harmonic phase carries no history, so locality is exact
    noise = nsf_oracle.reference_noise(1, T, cfg.upp, 114514)
    gen = pin(rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=gpu, operand="fp16", max_B=2, max_T=T), RB_STREAM=int(rb_stream))
    zd, fd, gd, nd = z.to(gpu), f0.to(gpu), g.to(gpu), noise.to(gpu)
    full = gen(zd, fd, gd, noise=nd)
    assert torch.isfinite(full).all() and full.shape == (1, 1, T * cfg.upp)
    assert torch.equal(full, gen(zd, fd, gd, noise=nd))  # bitwise reproducible
    # batch independence: the same clip twice in a batch of 2
    both = gen(zd.repeat(2, 1, 1), fd.repeat(2, 1), gd.repeat(2, 1, 1), noise=nd.repeat(2, 1))
    assert torch.equal(both[0], full[0]) and torch.equal(both[1], full[0])
    # locality: window [400, 700) with a 40-frame margin each side (receptive field of the stack < 30 frames)
    a, b, m = 400, 700, 40
    win = gen(zd[:, :, a - m:b + m].contiguous(), fd[:, a - m:b + m].contiguous(), gd,
              noise=nd[:, (a - m) * cfg.upp:(b + m) * cfg.upp].contiguous())
    x = full[0, 0, a * cfg.upp:b * cfg.upp]
    y = win[0, 0, m * cfg.upp:(m + b - a) * cfg.upp]
    assert rms(x, y) <= 2e-6, "tiling / halo inconsistency at full size: %.3e" % rms(x, y)
    # and against the oracle on that window only (the oracle needs ~1 s for 380 frames)
    with torch.no_grad():
        ref = nsf_oracle.generator_forward(cfg, w, z[:, :, a - m:b + m], f0[:, a - m:b + m], g, noise[:, (a - m) * cfg.upp:(b + m) * cfg.upp])
    assert rms(win.cpu(), ref) <= 1e-3


@pytest.mark.parametrize("cfg_name,T", [("v1_40k", 210), ("v2_32k", 170), ("v1_32k", 190), ("v1_48k", 150)])
def test_every_shipped_config_multi_tile_vs_oracle(cfg_name, T, gpu):
    """Several tiles per kernel at every stage for the other shipped configs (different polyphase tap patterns,
    5-stage models whose last stage has 16 channels), fp16 operands against the oracle."""
    import rvc_amd

    cfg = nsf_oracle.CONFIGS[cfg_name]
    w = synth.make_dec_weights(cfg, 21)
    z, f0, g = synth.make_dec_inputs(cfg, 1, T, 21)
    noise = nsf_oracle.reference_noise(1, T, cfg.upp, 9)
    with torch.no_grad():
        ref = nsf_oracle.generator_forward(cfg, w, z, f0, g, noise)
    gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=gpu, operand="fp16", max_B=1, max_T=T)
    out = gen(z.to(gpu), f0.to(gpu), g.to(gpu), noise=noise.to(gpu)).cpu()
    assert rms(out, ref) <= 1e-3, "%s: %.3e" % (cfg_name, rms(out, ref))
    gen32 = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=gpu, operand="fp32", max_B=1, max_T=T)
    out32 = gen32(z.to(gpu), f0.to(gpu), g.to(gpu), noise=noise.to(gpu)).cpu()
    assert rms(out32, ref) <= 2e-5


def test_smoke_entry_point(gpu):
    import __graft_entry__

    __graft_entry__.smoke()


def _full_clip_inputs():
    """bench.py's exact inputs: synth.make_dec_inputs(cfg, 1, 1198, 1234) (80 % voiced f0) and the reference's noise seed."""
    cfg = nsf_oracle.CONFIGS["v2_48k"]
    w = synth.make_dec_weights(cfg, 1234)
    z, f0, g = synth.make_dec_inputs(cfg, 1, 1198, 1234)
    noise = nsf_oracle.reference_noise(1, 1198, cfg.upp, 114514)
    return cfg, w, z, f0, g, noise


def test_full_clip_voiced_whole_waveform_vs_reference_golden_and_oracle(gpu):
    """BASELINE configs[1]: the WHOLE 10 s clip (T = 1198, voiced) against (a) the waveform the real reference produced
    (fixture full_v2_48k_T1198_voiced, oracle/make_golden.py:dec_full_case) and (b) the oracle run here, plus the `har`
    tap: with T > 256 every thread of k_phase_scan scans several frames and the running phase crosses waves through
    LDS -- with non-zero increments, which the f0 = 0 locality test cannot exercise."""
    import hashlib

    import rvc_amd

    cfg, w, z, f0, g, noise = _full_clip_inputs()
    d = load_golden("full_v2_48k_T1198_voiced")
    sha = lambda t: hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()
    assert synth.weights_sha256(w) == d["weights_sha256"]
    assert sha(z) + sha(f0) + sha(g) + sha(noise) == str(d["inputs_sha256"]), "seeded inputs differ from the fixture's"
    assert float((f0 > 0).float().mean()) > 0.5
    gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=gpu, operand="fp16", max_B=1, max_T=1198)
    zd, fd, gd, nd = z.to(gpu), f0.to(gpu), g.to(gpu), noise.to(gpu)
    out = gen(zd, fd, gd, noise=nd).cpu()
    assert out.shape == d["out"].shape and torch.isfinite(out).all()
    e_ref = rms(out, d["out"])
    assert e_ref <= 1e-3, "whole 10 s clip vs the reference waveform: RMS %.3e" % e_ref
    taps = {}
    with torch.no_grad():
        ora = nsf_oracle.generator_forward(cfg, w, z, f0, g, noise, taps=taps)
    assert rms(ora, d["out"]) <= 2e-6  # the oracle reproduces the reference at full size on this host too
    assert rms(out, ora) <= 1e-3
    har = gen.debug_tap("har", zd, fd, gd, noise=nd)
    rel = rms(har, taps["har"]) / float(taps["har"].pow(2).mean().sqrt())
    assert rel <= 1e-6, "har (sine source over 1198 voiced frames): relative RMS %.2e" % rel
    # both K loops of the streaming ResBlock kernel of the C = 128 stage on the whole clip
    for small in ("1", "kl2"):
        pin(gen, **_rs_opts(small))
        o = gen(zd, fd, gd, noise=nd).cpu()
        assert torch.isfinite(o).all() and rms(o, d["out"]) <= 1e-3, "%s: RMS %.3e vs the reference waveform" % (small, rms(o, d["out"]))
        assert rms(o, out) <= 5e-4
    pin(gen, **{k: None for k in _rs_opts("0")})
    # inter-stage streams: fp16 (default: option Y_F16 = the ResBlock outputs of stages 1-3, X0_F16 = the ups output of the k_rb_full
    # stages and -- round 5 -- of the streaming stage) against fp32 streams.  Budget: inside the 5e-4 gate on this clip and within 25 %
    # of what the fp32 streams give (the operand rounding of 72 convolutions dominates both).  Measured: fp32 streams 2.75e-4; fp16 Y + X0
    # at C <= 64 (round 4) 3.12e-4; + fp16 X0 of the streaming stage 3.20e-4 (x 1.16)
    o32s = pin(gen, Y_F16=0)(zd, fd, gd, noise=nd).cpu()
    pin(gen, Y_F16=None)
    e32s = rms(o32s, d["out"])
    assert e32s <= 1e-3 and e_ref <= 5e-4 and e_ref <= 1.25 * e32s, "fp16 inter-stage streams %.3e vs fp32 streams %.3e" % (e_ref, e32s)
    assert 0 < rms(out, o32s) <= 3e-4
    # the streaming kernel's fp32-input instantiation (X0 of its stage kept fp32; what round 4 shipped): still green, a hair closer
    o4 = pin(gen, X0_F16_NOSTREAM=1)(zd, fd, gd, noise=nd).cpu()
    pin(gen, X0_F16_NOSTREAM=None)
    # (measured 3.12e-4 against 3.20e-4; the two waveforms are 2.9e-4 apart: rounding X0 once re-rounds every operand downstream, the two
    #  are different realisations of the same rounding noise)
    assert rms(o4, d["out"]) <= e_ref * 1.02 and 0 < rms(o4, out) <= 4e-4
    for k in ("stage1", "stage2", "stage3"):  # the un-divided stage sums read back from the fp16 streams
        got = gen.debug_tap(k, zd, fd, gd, noise=nd)
        exp = taps[k] * cfg.num_kernels
        rel = rms(got, exp) / float(exp.pow(2).mean().sqrt())
        assert rel <= 2e-3, "%s (fp16 streams): relative RMS %.2e" % (k, rel)
    # exact-fp32 path on the same clip
    gen32 = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=gpu, operand="fp32", max_B=1, max_T=1198)
    assert rms(gen32(zd, fd, gd, noise=nd).cpu(), d["out"]) <= 2e-5


@pytest.mark.parametrize("rb_stream", ["0", "1", "kl2"])
def test_batch_16_full_clips_equal_their_single_clip_results(rb_stream, gpu):
    """BASELINE configs[2] geometry (grid.z = batch, multi-GB streams, the large-batch launch shapes): 16 different full-size
    voiced clips in one call; every item must be BIT-equal to the same clip run alone -- with the ResBlock kernel family
    pinned (tile kernels / streaming kernel, whose strip partition changes completely between B = 16 and B = 1) -- and item
    0 must meet the parity bar against the reference golden.  Unpinned (the launcher's own choice) the two runs may pick
    different families per stage and agree to fp32 rounding instead."""
    import rvc_amd

    cfg = nsf_oracle.CONFIGS["v2_48k"]
    w = synth.make_dec_weights(cfg, 1234)
    B, T = 16, 1198
    zs, fs, gs, ns = [], [], [], []
    for b in range(B):
        z, f0, g = synth.make_dec_inputs(cfg, 1, T, 1234 + b)
        zs.append(z), fs.append(torch.roll(f0, 37 * b, dims=1)), gs.append(g)
        ns.append(nsf_oracle.reference_noise(1, T, cfg.upp, 114514 + b))
    Z, F, G, N = torch.cat(zs).to(gpu), torch.cat(fs).to(gpu), torch.cat(gs).to(gpu), torch.cat(ns).to(gpu)
    gen = pin(rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=gpu, operand="fp16", max_B=B, max_T=T),
              RB_STREAM=int(rb_stream) if rb_stream in "01" else 1, RS_KL=2 if rb_stream == "kl2" else 1)
    out = gen(Z, F, G, noise=N)
    assert out.shape == (B, 1, T * cfg.upp) and torch.isfinite(out).all()
    d = load_golden("full_v2_48k_T1198_voiced")
    assert rms(out[0:1].cpu(), d["out"]) <= 1e-3
    for b in (0, 1, 7, 15):
        one = gen(Z[b:b + 1].contiguous(), F[b:b + 1].contiguous(), G[b:b + 1].contiguous(), noise=N[b:b + 1].contiguous())
        assert torch.equal(one[0], out[b]), "batch item %d differs from its single-clip result" % b
    pin(gen, RB_STREAM=None)
    auto = gen(Z, F, G, noise=N)
    assert rms(auto.cpu(), out.cpu()) <= 5e-4  # different families: last-bit fp32 differences re-round some fp16 operands


def test_batch_64_bench_config_3_geometry(gpu):
    """BASELINE configs[2] exactly: 64 full-size voiced clips in one call (35 GB workspace, 4.7 GB stage-3 streams: element
    offsets beyond 2^32 bytes, grid of 64 utterances).  First and last items (the ones at the ends of the address range) and
    one in the middle are bit-equal to their single-clip results with the kernel family pinned; item 0 meets the parity bar
    against the reference golden; the launcher's own (unpinned) choice agrees to operand rounding."""
    import rvc_amd

    free, _ = torch.cuda.mem_get_info(gpu)
    if free < 60 * 2**30:
        pytest.skip("needs ~40 GB of free HBM")
    cfg = nsf_oracle.CONFIGS["v2_48k"]
    w = synth.make_dec_weights(cfg, 1234)
    B, T = 64, 1198
    zs, fs, gs, ns = [], [], [], []
    for b in range(B):
        z, f0, g = synth.make_dec_inputs(cfg, 1, T, 1234 + b)
        zs.append(z), fs.append(torch.roll(f0, 17 * b, dims=1)), gs.append(g)
        ns.append(nsf_oracle.reference_noise(1, T, cfg.upp, 114514 + b))
    Z, F, G, N = torch.cat(zs).to(gpu), torch.cat(fs).to(gpu), torch.cat(gs).to(gpu), torch.cat(ns).to(gpu)
    gen = pin(rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=gpu, operand="fp16", max_B=B, max_T=T), RB_STREAM=1)
    out = gen(Z, F, G, noise=N)
    assert out.shape == (B, 1, T * cfg.upp) and torch.isfinite(out).all()
    d = load_golden("full_v2_48k_T1198_voiced")
    assert rms(out[0:1].cpu(), d["out"]) <= 1e-3
    for b in (0, 41, 63):
        one = gen(Z[b:b + 1].contiguous(), F[b:b + 1].contiguous(), G[b:b + 1].contiguous(), noise=N[b:b + 1].contiguous())
        assert torch.equal(one[0], out[b]), "batch item %d differs from its single-clip result" % b
    pin(gen, RB_STREAM=None)
    auto = gen(Z, F, G, noise=N)
    assert rms(auto.cpu(), out.cpu()) <= 5e-4


# ---- streaming fused ResBlock kernel (csrc/rb_stream_kernels.hpp) -------------------------------------------------------
# At full clip size the launcher picks it by itself (the full-size tests above run it); option RB_STREAM=1 forces it for
# the small golden cases too (single short strips, sequence ends inside the first step), RS_SMALL selects the time-tile
# height, RS_KL the K loop (1 = conv_run, 2 = the lean kconv with the coalesced step IO: the shipped default).

def _rs_opts(small):
    """test parameter -> options of the streaming ResBlock launcher (tile height and K loop pinned explicitly)"""
    o = {"RS_SMALL": None, "RS_KL": 1}
    if small == "kl2":     # k_rb_stream with the lean K loop (kconv), dual-written X tail, LDS-only barriers
        o["RS_SMALL"], o["RS_KL"] = 1, 2
    else:
        o["RS_SMALL"] = int(small)
    return o


@pytest.mark.parametrize("small", ["0", "1", "kl2"])
@pytest.mark.parametrize("name", ["dec_v2_48k_B1_T70", "dec_v2_48k_B2_T24", "dec_v1_40k_B1_T20", "dec_v1_32k_B1_T16",
                                  "dec_nof0_v2_48k_B1_T16", "dec_v1_40k_nres_T31"])
def test_streaming_resblock_kernel_on_reference_goldens(name, small, gpu):
    d = load_golden(name)
    cfg, w = golden_config_and_weights(d)
    for operand in ("fp16", "bf16"):
        gen = hip_gen(cfg, w, operand, gpu)
        o = _rs_opts(small)
        pin(gen, **o)
        out = run_golden(d, cfg, w, operand, gpu, RB_STREAM=1, RS_SMALL=o["RS_SMALL"])
        pin(gen, **{k: None for k in o})
        assert torch.isfinite(out).all()
        e = rms(out, d["out"])
        assert e <= BAR[operand], "%s/%s (streaming resblocks): RMS error %.3e" % (name, operand, e)


@pytest.mark.parametrize("small", ["0", "1", "kl2"])
def test_streaming_resblock_kernel_stage_taps_and_many_strips(small, gpu):
    """Forced onto a clip of 300 frames: hundreds of strips of one to three steps each (every strip boundary, warm-up
    and tail case), batch of 2 with different inputs; per-stage taps and the waveform against the oracle."""
    import rvc_amd

    cfg = nsf_oracle.CONFIGS["v2_48k"]
    w = synth.make_dec_weights(cfg, 31)
    B, T = 2, 300
    z, f0, g = synth.make_dec_inputs(cfg, B, T, 31)
    noise = nsf_oracle.reference_noise(B, T, cfg.upp, 17)
    taps = {}
    with torch.no_grad():
        ref = nsf_oracle.generator_forward(cfg, w, z, f0, g, noise, taps=taps)
    gen = pin(rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=gpu, operand="fp16", max_B=B, max_T=T), RB_STREAM=1, **_rs_opts(small))
    zd, fd, gd, nd = z.to(gpu), f0.to(gpu), g.to(gpu), noise.to(gpu)
    out = gen(zd, fd, gd, noise=nd).cpu()
    assert rms(out, ref) <= 1e-3, "streaming resblocks, T=300 B=2: %.3e" % rms(out, ref)
    for k in ("stage0", "stage1", "stage2", "stage3"):
        got = gen.debug_tap(k, zd, fd, gd, noise=nd)
        exp = taps[k] * cfg.num_kernels
        rel = rms(got, exp) / float(exp.pow(2).mean().sqrt())
        assert rel <= 2e-3, "%s: relative RMS %.2e" % (k, rel)
    pin(gen, RB_STREAM=0)  # and the tile kernels on the same input agree with it to operand rounding
    out0 = gen(zd, fd, gd, noise=nd).cpu()
    assert rms(out0, ref) <= 1e-3 and rms(out0, out) <= 5e-4


# ---- ragged batches: rvcmi_nsf_forward `lengths` -----------------------------------------------------------------
@pytest.mark.parametrize("family", ["tiles", "stream", "fp32"])
def test_ragged_batch_items_equal_separate_calls_bit_for_bit(family, gpu):
    """Four utterances of different lengths in ONE call with ``lengths``: every item must be BIT-equal to a separate call of its
    own length (each layer zero-pads behind the item's own end; the padded batch of the reference would leak the rows behind a
    short item into its tail) -- with the ResBlock kernel family pinned, as for the equal-length batches above -- and the rows
    behind an item's end must be zero.  Lengths chosen to end inside a tile, on a tile edge, after a single frame's worth of rows;
    the longest item doubles as the un-ragged reference (item 0 == the same call without ``lengths``)."""
    import rvc_amd

    cfg = nsf_oracle.CONFIGS["v2_48k"]
    w = synth.make_dec_weights(cfg, 99)
    lens = [300, 137, 64, 1, 256] if family != "fp32" else [40, 17, 1]
    B, T = len(lens), max(lens)
    zs, fs, gs, ns = [], [], [], []
    for b in range(B):
        z, f0, g = synth.make_dec_inputs(cfg, 1, T, 500 + b)
        zs.append(z), fs.append(torch.roll(f0, 11 * b, dims=1)), gs.append(g)
        ns.append(nsf_oracle.reference_noise(1, T, cfg.upp, 900 + b))
    Z, F, G, N = torch.cat(zs).to(gpu), torch.cat(fs).to(gpu), torch.cat(gs).to(gpu), torch.cat(ns).to(gpu)
    for b, n in enumerate(lens):
        Z[b, :, n:] = 0  # what z * x_mask hands the decoder
    gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=gpu, operand="fp32" if family == "fp32" else "fp16", max_B=B, max_T=T)
    if family != "fp32":
        pin(gen, RB_STREAM=1 if family == "stream" else 0, NO_RB_SPLIT=1)
    out = gen(Z, F, G, noise=N, lengths=torch.tensor(lens))
    assert out.shape == (B, 1, T * cfg.upp) and torch.isfinite(out).all()
    full = gen(Z, F, G, noise=N)
    assert torch.equal(full[0], out[0])  # the longest item: nothing is masked
    for b, n in enumerate(lens):
        one = gen(Z[b:b + 1, :, :n].contiguous(), F[b:b + 1, :n].contiguous(), G[b:b + 1].contiguous(),
                  noise=N[b:b + 1, :n * cfg.upp].contiguous())
        assert torch.equal(one[0, 0], out[b, 0, :n * cfg.upp]), "item %d (%d frames) differs from its separate call: rms %.3e" % (
            b, n, rms(one[0, 0].cpu(), out[b, 0, :n * cfg.upp].cpu()))
        assert not out[b, 0, n * cfg.upp:].any()
        if n < T and n > 8:  # ... and the padded batch (no lengths) is NOT that: its tail is contaminated -- the reason for the argument
            assert not torch.equal(full[b, 0, :n * cfg.upp], out[b, 0, :n * cfg.upp])
    # the oracle (= the reference's arithmetic) on one short item, run alone
    with torch.no_grad():
        n = lens[1]
        ref = nsf_oracle.generator_forward(cfg, w, zs[1][:, :, :n], fs[1][:, :n], gs[1], ns[1][:, :n * cfg.upp])
    assert rms(out[1:2, :, :n * cfg.upp].cpu(), ref) <= (2e-5 if family == "fp32" else 1e-3)
    with pytest.raises(ValueError):
        gen(Z, F, G, noise=N, lengths=torch.tensor([T + 1] + lens[1:]))
    with pytest.raises(ValueError):
        gen(Z, F, G, 10, noise=N, lengths=torch.tensor(lens))


def test_conv_post_kernels_agree_bit_for_bit(gpu):
    """``conv_post`` + tanh exists in two forms (option POST_DMA): 0 = ``k_post`` (register-staged; fp32 streams, other shapes), 1 = ``k_post_dma``
    with the next tile in flight by LDS-DMA (the default for the shipped three-fp16-stream case).  The arithmetic and its order are the same, so
    the waveforms must be BIT-equal -- on a ragged batch (tiles behind an item's end, clamped rows at both ends, an item one frame long) and on a
    clip long enough for every persistent block to walk several tiles, repeatedly (a race between the DMA pieces -- issued through inline asm,
    invisible to the compiler's wait-count insertion -- and the convert pass would show up as run-to-run differences)."""
    import rvc_amd

    cfg = nsf_oracle.CONFIGS["v2_48k"]
    w = synth.make_dec_weights(cfg, 99)
    lens = [300, 137, 64, 1, 256]
    B, T = len(lens), max(lens)
    zs, fs, gs, ns = [], [], [], []
    for b in range(B):
        z, f0, g = synth.make_dec_inputs(cfg, 1, T, 500 + b)
        zs.append(z), fs.append(f0), gs.append(g), ns.append(nsf_oracle.reference_noise(1, T, cfg.upp, 900 + b))
    Z, F, G, N = torch.cat(zs).to(gpu), torch.cat(fs).to(gpu), torch.cat(gs).to(gpu), torch.cat(ns).to(gpu)
    gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=gpu, operand="fp16", max_B=B, max_T=1198)
    z, f0, g = synth.make_dec_inputs(cfg, 1, 1198)
    nz = nsf_oracle.reference_noise(1, 1198, cfg.upp)
    outs, full = {}, {}
    for mode in (0, 1):
        pin(gen, POST_DMA=mode)
        outs[mode] = [gen(Z, F, G, noise=N, lengths=torch.tensor(lens)).clone() for _ in range(4)]
        full[mode] = [gen(z.to(gpu), f0.to(gpu), g.to(gpu), noise=nz.to(gpu)).clone() for _ in range(4)]
        for runs in (outs[mode], full[mode]):
            assert all(torch.equal(runs[0], o) for o in runs[1:]), "POST_DMA=%d: run-to-run difference" % mode
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(full[0][0], full[1][0])
    pin(gen, POST_DMA=None)
