"""CPU suite, part 1: the oracle against the golden fixtures made from the real reference, and the two
independent restatements of the IVF search against each other."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_config_and_weights, golden_names, load_golden
from oracle import front_oracle, glue_oracle, ivf_oracle, nsf_oracle, synth
from oracle.front_oracle import FrontConfig


@pytest.mark.parametrize("name", golden_names("dec_"))
def test_generator_oracle_matches_reference_golden(name):
    """oracle/nsf_oracle.py == rvc/layers/nsf.py NSFGenerator.forward (fixture produced by the reference itself)."""
    d = load_golden(name)
    cfg, w = golden_config_and_weights(d)
    f0 = torch.from_numpy(d["f0"]) if "f0" in d else None
    noise = torch.from_numpy(d["noise"]) if "noise" in d else None
    n_res = None if int(d["n_res"]) < 0 else int(d["n_res"])
    taps = {}
    with torch.no_grad():
        out = nsf_oracle.generator_forward(cfg, w, torch.from_numpy(d["z"]), f0, torch.from_numpy(d["g"]), noise, n_res=n_res, taps=taps)
    assert out.shape == d["out"].shape
    assert np.abs(out.numpy() - d["out"]).max() < 5e-6
    if "har" in d:
        assert np.abs(taps["har"].numpy() - d["har"]).max() < 1e-6


def test_generator_oracle_infer_boundary_golden():
    """Decoder inputs/outputs captured INSIDE the reference's net_g.infer (loader, enc_p, flow, RNG order)."""
    d = load_golden("infer_v2_48k_T20")
    cfg = nsf_oracle.CONFIGS["v2_48k"]
    _, expect = synth.make_legacy_checkpoint(cfg, "v2", int(d["seed"]))
    for k, v in d.items():  # the folded weights the reference loader produced (small tensors are stored)
        if k.startswith("w::"):
            assert np.allclose(expect[k[3:]].numpy(), v, rtol=0, atol=2e-7 * max(1.0, np.abs(v).max()))
    with torch.no_grad():
        out = nsf_oracle.generator_forward(cfg, expect, torch.from_numpy(d["z"]), torch.from_numpy(d["f0"]), torch.from_numpy(d["g"]),
                                           torch.from_numpy(d["noise"]))
    assert np.abs(out.numpy() - d["out"]).max() < 5e-6


@pytest.mark.parametrize("name", golden_names("front_"))
def test_front_oracle_matches_reference_golden(name):
    """oracle/front_oracle.py == the reference's TextEncoder + prior sample + reversed ResidualCouplingBlock
    (fixtures produced by the reference modules themselves), including every stored intermediate stage."""
    d = load_golden(name)
    fcfg = FrontConfig(in_channels=int(d["in_channels"]))
    wf = synth.make_front_weights(fcfg, int(d["seed"]))
    assert synth.weights_sha256(wf) == d["weights_sha256"]
    fh = None if int(d["flow_head"]) < 0 else int(d["flow_head"])
    taps = {}
    with torch.no_grad():
        z, m1, g = front_oracle.infer_front(fcfg, wf, torch.from_numpy(d["phone"]), torch.from_numpy(d["pitch"]), torch.from_numpy(d["lengths"]),
                                            torch.from_numpy(d["sid"]), torch.from_numpy(d["noise"]), fh, taps)
        z = z * m1
    assert np.abs(g.numpy() - d["g"]).max() == 0
    assert np.abs(z.numpy() - d["z"]).max() < 2e-5
    for k, ok in (("emb", "emb"), ("attn0", "attn0"), ("layer0", "layer0"), ("layer5", "layer%d" % (fcfg.n_layers - 1)), ("z_p", "z_p")):
        assert np.abs(taps[ok].transpose(1, 2).numpy() - d[k]).max() < 2e-5, k


@pytest.mark.parametrize("name", golden_names("infer_full_"))
def test_whole_infer_oracle_matches_reference_golden(name):
    """front oracle + generator oracle == the reference's net_g.infer waveform (full and realtime-partial geometry)."""
    d = load_golden(name)
    cfg, fcfg = nsf_oracle.CONFIGS["v2_48k"], FrontConfig()
    wd, wf = synth.make_dec_weights(cfg, int(d["seed"])), synth.make_front_weights(fcfg, int(d["seed"]))
    assert synth.weights_sha256(wd) == str(d["dec_sha256"]) and synth.weights_sha256(wf) == str(d["front_sha256"])
    T = d["phone"].shape[1]
    sh, rl, rl2 = (None if int(d[k]) < 0 else int(d[k]) for k in ("skip_head", "return_length", "return_length2"))
    fh = None if sh is None else max(sh - 24, 0)
    with torch.no_grad():
        z, m1, g = front_oracle.infer_front(fcfg, wf, torch.from_numpy(d["phone"]), torch.from_numpy(d["pitch"]), torch.tensor([T]),
                                            torch.from_numpy(d["sid"]), torch.from_numpy(d["noise_zp"]), fh)
        z = z * m1
        pf = torch.from_numpy(d["pitchf"])
        if sh is not None:  # synthesizers.py:172-185
            z = z[:, :, sh - fh:sh - fh + rl]
            pf = pf[:, sh:sh + rl]
        out = nsf_oracle.generator_forward(cfg, wd, z, pf, g, torch.from_numpy(d["noise_dec"]), n_res=rl2)
    assert out.shape == d["out"].shape and np.abs(out.numpy() - d["out"]).max() < 2e-5


def test_front_oracle_masking_semantics():
    """Padding frames never influence valid ones and come out as exact zeros (x_mask, encoders.py:145-148; the -1e4 fill of
    attentions.py:114-115)."""
    fcfg = FrontConfig()
    wf = synth.make_front_weights(fcfg, 3)
    B, T, L1 = 2, 24, 17
    phone = synth.make_phone(B, T, 768, 3)
    pitch = synth.make_pitch(synth.make_f0(B, T))
    lengths, sid = torch.tensor([T, L1]), torch.tensor([0, 1])
    noise = torch.randn(B, 192, T, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        z0, m1, _ = front_oracle.infer_front(fcfg, wf, phone, pitch, lengths, sid, noise)
        ph2 = phone.clone()
        ph2[1, L1:] = 37.0
        z1, _, _ = front_oracle.infer_front(fcfg, wf, ph2, pitch, lengths, sid, noise)
    assert torch.equal((z0 * m1)[1, :, :L1], (z1 * m1)[1, :, :L1]) and ((z1 * m1)[1, :, L1:] == 0).all()


def test_glue_oracle_matches_reference_golden():
    """oracle/glue_oracle.py == RMVPE._decode + F0Predictor._resize_f0/_interpolate_f0 + post_process of the reference
    (fixture produced by those functions themselves): bit-exact, all edge layouts."""
    d = load_golden("glue_f0")
    for c in sorted({k.split("::")[0] for k in d}):
        n, p_len, key = (int(v) for v in d[c + "::meta"])
        assert np.array_equal(glue_oracle.rmvpe_decode(d[c + "::salience"], 0.03), d[c + "::f0_decoded"]), c
        pitch, pitchf = glue_oracle.rmvpe_f0(d[c + "::salience"], p_len, key, 0.03)
        assert np.array_equal(pitch, d[c + "::pitch"]) and np.array_equal(pitchf, d[c + "::pitchf"]), c


def test_glue_interpolate_f0_quirks():
    """The reference's gap filling (f0.py:31-66): leading gaps copy the next voiced value, inner gaps ramp with step (next - prev) / gap_length
    (so the last filled frame already equals the next voiced value), a trailing
    gap repeats the last voiced value, and a gap that ends on the LAST frame overwrites it."""
    f = glue_oracle.interpolate_f0(np.array([0, 0, 100.0, 0, 0, 130.0, 0, 0]))
    assert np.allclose(f, [100, 100, 100, 115, 130, 130, 130, 130])
    f = glue_oracle.interpolate_f0(np.array([100.0, 0, 0, 200.0]))
    assert np.allclose(f, [100, 100, 100, 100])
    assert np.array_equal(glue_oracle.interpolate_f0(np.zeros(5)), np.zeros(5))


def test_sine_source_phase_is_continuous_and_unvoiced_is_noise_only():
    f0 = torch.tensor([[0.0, 0.0, 220.0, 220.0, 0.0, 330.0]])
    s = nsf_oracle.sine_source(f0, 480, 48000, None).reshape(6, 480)
    assert torch.all(s[0] == 0) and torch.all(s[1] == 0) and torch.all(s[4] == 0)  # no noise injected -> silence when unvoiced
    assert abs(s[2].abs().max().item() - 0.1) < 1e-3
    # phase continuity across frames: frames 2+3 form ONE uninterrupted 220 Hz sine of 960 samples
    n = torch.arange(1, 961, dtype=torch.float64)
    expect = 0.1 * torch.sin(2 * torch.pi * 220.0 / 48000.0 * n)
    assert (torch.cat([s[2], s[3]]).double() - expect).abs().max() < 1e-5


def _c_oracle():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libivf_oracle.so"))
    return lib


def _c_search(lib, idx, q, k, nprobe, f32=0):
    nq, d = q.shape
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    P = np.empty((nq, k), np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.ivf_search(vp(q), C.c_int64(nq), C.c_int(d), vp(idx["centroids"]), C.c_int64(idx["nlist"]), C.c_int(nprobe),
                   vp(idx["list_offsets"]), vp(idx["ids"]), vp(idx["vecs"]), C.c_int(k), vp(D), vp(I), vp(P), C.c_int(f32))
    return D, I, P


@pytest.mark.parametrize("n,d,nprobe,k", [(2000, 64, 1, 8), (3000, 256, 3, 8), (500, 32, 1, 5), (60, 16, 2, 8)])
def test_ivf_python_and_c_restatements_agree(n, d, nprobe, k):
    idx = synth.make_ivf(n, d, seed=n + d, dup=7)
    rng = np.random.default_rng(5)
    q = rng.standard_normal((97, d), dtype=np.float32)
    q[:4] = idx["xb"][:4]  # exact hits
    D1, I1 = ivf_oracle.search(idx, q, k, nprobe)
    D2, I2, _ = _c_search(_c_oracle(), idx, q, k, nprobe)
    assert np.array_equal(I1, I2)
    assert np.allclose(D1, D2, rtol=1e-6, atol=0)
    assert np.all(np.diff(D1.astype(np.float64), axis=1) >= 0)  # ascending
    assert np.all(D1[:4, 0] == 0) and np.array_equal(np.sort(I1[:4, 0]), np.sort(I1[:4, 0]))


def test_c_oracle_wrappers_equal_the_numpy_restatement():
    """``ivf_oracle.search_c`` / ``blend_c`` (what the GPU tests at BASELINE configs[2] / [3] scale check against: 38 336 queries are
    minutes of numpy loops, seconds of OpenMP C) against the numpy restatement on a size both finish at once."""
    idx = synth.make_ivf(3000, 64, seed=11, dup=5)
    q = np.random.default_rng(2).standard_normal((257, 64), dtype=np.float32)
    q[:3] = idx["xb"][:3]
    D1, I1 = ivf_oracle.search(idx, q, 8)
    D2, I2, P2 = ivf_oracle.search_c(idx, q, 8)
    assert np.array_equal(I1, I2) and np.array_equal(np.where(P2 >= 0, idx["ids"][P2], -1), I2)
    assert np.allclose(D1, D2, rtol=1e-6, atol=0)
    q2 = q[3:]  # (an exact hit has distance 0: weight inf, NaN row -- covered on the GPU side)
    b1 = ivf_oracle.search_blend(idx, q2, 0.75, 8)
    D3, _, P3 = ivf_oracle.search_c(idx, q2, 8)
    b2 = ivf_oracle.blend_c(idx, q2, D3, P3, 0.75)
    assert np.sqrt(np.mean((b1 - b2) ** 2)) <= 1e-6


def test_ivf_edge_semantics_short_lists_and_padding():
    """Lists shorter than k pad with id -1 / FLT_MAX (faiss' L2 heap sentinel); empty query set; ties -> lowest id."""
    idx = synth.make_ivf(40, 8, nlist=10, seed=3)
    q = np.random.default_rng(0).standard_normal((20, 8), dtype=np.float32)
    D, I = ivf_oracle.search(idx, q, 8)
    sizes = np.diff(idx["list_offsets"])
    assert sizes.min() < 8
    assert (I == -1).any() and np.all(D[I == -1] == ivf_oracle.FLT_MAX)
    assert np.all((I >= 0).sum(1) == np.minimum(8, sizes[ivf_oracle.coarse_assign(idx, q, 1)[:, 0]]))
    D0, I0 = ivf_oracle.search(idx, q[:0], 8)
    assert D0.shape == (0, 8) and I0.shape == (0, 8)
    # exact duplicates: the lower id must come first
    idx2 = synth.make_ivf(300, 16, nlist=4, seed=9)
    a, b = int(idx2["ids"][0]), int(idx2["ids"][1])
    idx2["vecs"][1] = idx2["vecs"][0]
    D2, I2 = ivf_oracle.search(idx2, idx2["vecs"][:1].copy(), 2, nprobe=4)
    assert D2[0, 0] == 0 and D2[0, 1] == 0 and list(I2[0]) == sorted([a, b])


def test_blend_matches_the_pipeline_expression_and_keeps_its_edge_cases():
    """pipeline.py:129-138 verbatim in numpy vs oracle.blend, including big_npy[-1] for id -1 and NaN on an exact hit."""
    idx = synth.make_ivf(400, 32, nlist=40, seed=11)
    big = ivf_oracle.reconstruct_n(idx)
    assert np.array_equal(big, idx["xb"])
    q = np.random.default_rng(1).standard_normal((50, 32), dtype=np.float32)
    q[0] = idx["xb"][17]
    score, ix = ivf_oracle.search(idx, q, 8)
    with np.errstate(all="ignore"):
        weight = np.square(1 / score)
        weight /= weight.sum(axis=1, keepdims=True)
        npy = np.sum(big[ix] * np.expand_dims(weight, axis=2), axis=1)
        ref = (torch.from_numpy(npy).unsqueeze(0) * 0.75 + (1 - 0.75) * torch.from_numpy(q).unsqueeze(0))[0].numpy()
    out = ivf_oracle.blend(q, score, ix, big, 0.75)
    assert np.isnan(out[0]).all() and np.isnan(ref[0]).all()  # inf/inf, as in the reference
    assert np.array_equal(out[1:], ref[1:])
    assert (ix == -1).any()  # some short lists: weight (1/FLT_MAX)^2 underflows to 0 and big_npy[-1] is harmless


def test_faiss_file_layout_roundtrip_python(tmp_path):
    idx = synth.make_ivf(700, 24, nlist=30, seed=2)
    for sparse in (False, True):
        p = str(tmp_path / ("a%d.index" % sparse))
        ivf_oracle.write_index(idx, p, sparse=sparse)
        with open(p, "rb") as f:
            assert f.read(4) == b"IwFl"
        r = ivf_oracle.read_index(p)
        for k in ("centroids", "list_offsets", "ids", "vecs"):
            assert np.array_equal(r[k], idx[k]), k
        assert (r["d"], r["ntotal"], r["nlist"], r["nprobe"]) == (24, 700, 30, 1)


def test_index_recipe_nlist():
    # web.py:544  min(int(16*sqrt(N)), N//39)
    assert synth.ivf_nlist(10000) == 256 and synth.ivf_nlist(200000) == 5128 and synth.ivf_nlist(1000000) == 16000


def test_real_hubert_rows_tie_semantics_and_the_faiss_validator(tmp_path, capsys):
    """The oracle on the real-feature distribution (fixture mute_hubert: duplicate rows => exact ties): python and C
    restatements agree, ties resolve to the lowest id; and oracle/validate_faiss_index.py accepts a well-formed file with a
    matching (D, I) dump and names the problem in a damaged one."""
    import ctypes as C
    import sys

    from conftest import load_golden
    from oracle import validate_faiss_index as V

    feats = load_golden("mute_hubert")["f256"]
    x = synth.make_mute_rows(feats, copies=12)
    idx = synth.make_ivf_from_rows(x)
    q = np.concatenate([feats[:80], (feats[:40] + np.float32(1e-3)).astype(np.float32)])
    D, I = ivf_oracle.search(idx, q, 8)
    assert (D[:80, :2] == 0).all()
    for i in range(80):  # the zero-distance group is listed in ascending id order
        z = I[i][D[i] == 0]
        assert (np.diff(z) > 0).all()
    path = str(tmp_path / "mute.index")
    ivf_oracle.write_index(idx, path)
    np.savez(str(tmp_path / "qdi.npz"), q=q, D=D, I=I, big_head=x[:64])
    old = sys.argv
    try:
        sys.argv = ["validate", path, "--dump", str(tmp_path / "qdi.npz")]
        assert V.main() == 0
        out = capsys.readouterr().out
        assert "layout: as expected" in out and "genuine id mismatches 0" in out
        buf = bytearray(open(path, "rb").read())
        assert bytes(buf[4 + 33 + 16:4 + 33 + 20]) == b"IxF2"
        buf[4 + 33 + 16:4 + 33 + 20] = b"IxXX"  # the quantizer fourcc
        open(path, "wb").write(buf)
        sys.argv = ["validate", path]
        assert V.main() == 1
        assert "quantizer fourcc is b'IxXX'" in capsys.readouterr().out
    finally:
        sys.argv = old


# ---- what faiss' own fp32 arithmetic would return (oracle/ivf_faisslike.py): the unpinned retrieval risk, quantified -------

def test_faisslike_fp32_kernels_agree_with_exact_distances_to_fp32_rounding():
    from oracle import ivf_faisslike as fl

    rng = np.random.default_rng(3)
    x = rng.standard_normal(768, dtype=np.float32)
    y = rng.standard_normal((200, 768), dtype=np.float32)
    exact = ((y.astype(np.float64) - x.astype(np.float64)) ** 2).sum(1)
    for name, v in fl.VARIANTS.items():
        got = fl.l2sqr_fp32(x, y, **v).astype(np.float64)
        bound = (2e-5 if v["lanes"] == 1 else 2e-6)  # one accumulator: ~d ulps; 8+ lanes: an order of magnitude better
        assert np.max(np.abs(got - exact) / exact) < bound, name
    # lane structure is really what it says: 8 lanes == sum over the 8 interleaved partial sums in the epilogue's order
    a = (y[:1] - x).astype(np.float32)
    lanes = np.zeros(8, np.float32)
    for s_ in range(96):
        lanes = (lanes + (a[0, 8 * s_:8 * s_ + 8] * a[0, 8 * s_:8 * s_ + 8]).astype(np.float32)).astype(np.float32)
    lo = (lanes[:4] + lanes[4:]).astype(np.float32)
    want = np.float32(np.float32(lo[0] + lo[1]) + np.float32(lo[2] + lo[3]))
    assert fl.l2sqr_fp32(x, y[:1], lanes=8, fma=False)[0] == want


def test_faisslike_search_vs_exact_on_the_bench_index_and_on_real_hubert_rows():
    """On bench.py's index (10000 x 768, 599 queries) every fp32 variant returns the SAME top-1 and the same top-8 set as the
    exact answer the HIP path reproduces.  On the real mute.npy rows (near-duplicate features, exact duplicate rows) the
    fp32 BLAS expansion of faiss' coarse quantizer picks another list for some queries: there faiss' own answer depends on
    its build, and 'bit-exact to faiss' is not a well-defined target -- the flips are all coarse flips, none inside a list."""
    from oracle import ivf_faisslike as fl

    idx = synth.make_ivf(10000, 768, seed=4321, kmeans_iters=1)
    q = synth.make_phone(1, 599, 768)[0].numpy()
    exact = ivf_oracle.search(idx, q, 8)
    lists = ivf_oracle.coarse_assign(idx, q, 1)[:, 0]
    for v in ("avx8", "avx32_fma"):
        c = fl.compare(exact, lists, fl.search_faisslike(idx, q, 8, v))
        assert c["top1_flips"] == 0 and c["top8_set_flips"] == 0 and c["coarse_list_flips"] == 0, (v, c)
        assert c["max_rel_dD_same_list"] < 1e-6
    feats = load_golden("mute_hubert")["f256"]
    x = synth.make_mute_rows(feats)
    idm = synth.make_ivf_from_rows(x)
    qq = np.concatenate([feats, (feats[:60] + np.float32(1e-3)).astype(np.float32)])
    exact = ivf_oracle.search(idm, qq, 8)
    lists = ivf_oracle.coarse_assign(idm, qq, 1)[:, 0]
    c = fl.compare(exact, lists, fl.search_faisslike(idm, qq, 8, "avx8"))
    assert c["top1_flips_same_list"] == 0 and c["top8_set_flips_same_list"] == 0  # inside the probed list fp32 changes nothing
    assert c["top1_flips"] == c["coarse_list_flips"]                              # every flip is a coarse-quantizer flip
    # with the direct-difference coarse path (nq < 20) the expansion's cancellation error is gone and so are most flips
    c16 = fl.compare((exact[0][:16], exact[1][:16]), lists[:16], fl.search_faisslike(idm, qq[:16], 8, "avx8"))
    assert c16["coarse_list_flips"] <= 1


def test_committed_faisslike_report_has_every_case():
    import json

    rep = json.load(open(os.path.join(ROOT, "profiles", "r03_faisslike_flips.json")))
    names = set(rep["cases"])
    assert {"baseline_10000x768_iid_599q", "stress_1000000x256_clustered_599q", "mute_hubert_768_real_rows_with_exact_duplicates"} <= names
    base = rep["cases"]["baseline_10000x768_iid_599q"]["variants"]
    assert all(v["top1_flips"] == 0 for v in base.values())


def test_front_oracle_matches_the_reference_modules_at_benchmark_size():
    """oracle/front_oracle.py == the reference's enc_p + flow^-1 at T = 1198 (fixture written by the reference modules)."""
    d = load_golden("bigfront_v2_B1_T1198_z")
    seed, T = int(d["seed"]), int(d["T"])
    fcfg = FrontConfig()
    wf = synth.make_front_weights(fcfg, seed)
    assert synth.weights_sha256(wf) == str(d["weights_sha256"])
    phone = synth.make_phone(1, T, 768, seed)
    pitch = synth.make_pitch(synth.make_f0(1, T))
    noise = torch.randn(1, 192, T, generator=torch.Generator().manual_seed(seed + 9))
    with torch.no_grad():
        z, m1, _ = front_oracle.infer_front(fcfg, wf, phone, pitch, torch.tensor([T]), torch.from_numpy(d["sid"]), noise)
    assert np.abs((z * m1).numpy() - d["z"]).max() < 5e-5


def test_sinc_resample_oracle_against_scipy_and_the_product_table():
    """The formant-shift resampler's oracle (torchaudio's published windowed-sinc formula; UNPINNED: torchaudio is absent) is a
    sane band-limited resampler -- within 1e-3 of scipy's polyphase filter on a band-limited signal -- and the product's
    filter table (rvc_amd.sinc_resample_kernel, torch float32) is the table the oracle builds in numpy float32."""
    from scipy import signal

    import rvc_amd

    n = 4230 * 3
    t = np.arange(n) / 42300.0
    x = (0.5 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 3000 * t + 1)).astype(np.float32)
    y = glue_oracle.sinc_resample(x, 423, 400)
    y2 = signal.resample_poly(x.astype(np.float64), 400, 423)
    assert y.shape == y2.shape == (12000,)
    assert np.sqrt(np.mean((y[200:-200] - y2[200:-200]) ** 2)) / np.sqrt(np.mean(y2 ** 2)) < 1e-3
    k, width, of, nf = rvc_amd.sinc_resample_kernel(538, 480)
    assert (width, of, nf) == (7, 269, 240) and tuple(k.shape) == (240, 2 * 7 + 269)
    imp = np.zeros(269 * 4, np.float32)
    imp[300] = 1.0  # the response to a unit impulse reads the table back: out[j*nf + p] = kernel[p][300 + width - j*of]
    r = glue_oracle.sinc_resample(imp, 538, 480)
    kk = k.numpy()
    for j in range(2):
        for p in (0, 7, 239):
            tap = 300 + width - j * of
            if 0 <= tap < kk.shape[1]:
                assert abs(r[j * nf + p] - kk[p, tap]) <= 1e-6


def test_three_instruction_division_by_three_is_the_ieee_quotient():
    """csrc/exact_fp.hpp:div3_exact (the stage mean `/ num_kernels` of nsf.py:186 in k_ups / k_post): every mantissa, both signs."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("check_div3", os.path.join(os.path.dirname(__file__), "..", "tools", "check_div3.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.mismatches(0, 1.0) == 0 and m.mismatches(3, -1.0) == 0


@pytest.mark.parametrize("k,dils,L,strips", [(11, [1, 3, 5], 1000, 3), (3, [1, 3, 5], 300, 1), (7, [3, 5], 777, 2)])
def test_streaming_resblock_schedule_model(k, dils, L, strips):
    """tools/model_rb_stream.py restates the buffer layout, row arithmetic, masks and history copies of the streaming ResBlock
    kernel in numpy; it must equal a direct evaluation of ResBlock1 (rvc/layers/residuals.py:68-85)."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("model_rb_stream", os.path.join(os.path.dirname(__file__), "..", "tools", "model_rb_stream.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = np.random.default_rng(k)
    C, nd = 16, len(dils)
    W1 = [rng.standard_normal((C, C, k), dtype=np.float32) / np.float32(np.sqrt(C * k)) for _ in range(nd)]
    W2 = [rng.standard_normal((C, C, k), dtype=np.float32) / np.float32(np.sqrt(C * k)) for _ in range(nd)]
    B1 = [(rng.standard_normal(C, dtype=np.float32) * np.float32(0.1)).astype(np.float32) for _ in range(nd)]
    B2 = [(rng.standard_normal(C, dtype=np.float32) * np.float32(0.1)).astype(np.float32) for _ in range(nd)]
    x = rng.standard_normal((L, C), dtype=np.float32)
    ref = m.reference(x, W1, B1, W2, B2, k, dils)
    sl = -(-L // strips)
    out = np.full((L, C), np.nan, np.float32)
    for s_ in range(strips):
        m.run_strip(x, W1, B1, W2, B2, k, dils, s_ * sl, min(L, (s_ + 1) * sl), 192, out)
    assert np.isfinite(out).all() and np.abs(out - ref).max() < 2e-4


def test_webui_golden_f0_track_is_what_the_fake_rmvpe_and_the_glue_oracle_make():
    """Fixture pipeline_v2_48k_webui was produced by the REAL ``Pipeline.pipeline`` -> ``Generator.calculate`` ->
    ``RMVPE.compute_f0`` (real ``_mel2hidden`` / ``_decode`` / resize / interpolate) -> ``post_process``.  The stand-in the
    GPU test uses on a box without the reference (``synth.FakeRMVPE``: restated pad-to-32 / crop) and the glue oracle must give
    exactly that track from the same seed -- with the threshold 0.03 of rvc/f0/gen.py:113, not the UI's filter_radius 3."""
    from oracle import glue_oracle, synth

    d = load_golden("pipeline_v2_48k_webui")
    fake = synth.FakeRMVPE(torch.device("cpu"), int(d["seed"]))
    n_pad = int(d["n_audio"]) + 2 * 16000 * int(d["cfg_x_pad"])
    sal = fake._mel2hidden(fake.mel_extractor(torch.zeros(1, n_pad), center=True)).squeeze(0).numpy()
    assert sal.shape == (n_pad // 160 + 1, 360) and sal.dtype == np.float32
    pitch, pitchf = glue_oracle.rmvpe_f0(sal, n_pad // 160, int(d["f0_up_key"]), 0.03)
    assert np.array_equal(pitch, d["pitch"]) and np.array_equal(pitchf, d["pitchf"])
    assert int(d["filter_radius"]) == 3 and float(d["rms_mix_rate"]) == 0.25 and float(d["index_rate"]) == 0.75
    # with the UI's filter_radius as the threshold (the round-3 bug) every frame would be unvoiced
    p3, f3 = glue_oracle.rmvpe_f0(sal, n_pad // 160, 0, 3.0)
    assert (f3 == 0).all() and (p3 == 1).all() and (pitchf > 0).any()
