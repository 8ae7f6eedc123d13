"""CPU suite, part 2: the C-ABI library loads and exports every symbol of include/rvcmi.h (no compute without a
GPU), host-side logic, loud failure without a GPU, and the N>1 path under gloo with world_size 2."""
import ctypes as C
import os
import re
import socket

import numpy as np
import pytest
import torch

from conftest import ROOT

import rvc_amd
from rvc_amd import _lib


def _header_functions():
    src = open(os.path.join(ROOT, "include", "rvcmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rvcmi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_symbol_of_the_header():
    names = _header_functions()
    assert len(names) >= 25
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "librvcmi.so does not export %s" % n
    bound = {s[0] for s in _lib.SYMBOLS}
    assert bound == set(names), "python binding and header drifted: %s" % (bound ^ set(names))
    assert _lib.lib().rvcmi_version() == _lib.RVCMI_VERSION == 2


def test_struct_layout_matches_the_header():
    # rvcmi_nsf_config: 6 ints + 2*8 + 1 + 4 + 4 + 16 + 1 ints
    assert C.sizeof(_lib.NsfConfig) == 4 * (6 + 16 + 1 + 4 + 4 + 16 + 1)
    assert C.sizeof(_lib.KernelStat) == 48 + 8 + 8 + 8 + 8
    assert C.sizeof(_lib.Tensor) == 8 + 8 + 8 + 32
    assert C.sizeof(_lib.FrontConfig) == 4 * 15  # rvcmi_front_config: 15 ints


def test_fails_loudly_without_a_gpu_and_on_bad_input(tmp_path):
    with pytest.raises(_lib.RvcmiError):
        rvc_amd.NSFGeneratorHIP({}, {}, device="cpu")
    with pytest.raises(_lib.RvcmiError):
        from oracle.front_oracle import FrontConfig
        rvc_amd.FrontHIP(vars(FrontConfig()), {}, device="cpu")
    with pytest.raises(_lib.RvcmiError):
        rvc_amd.read_index(str(tmp_path / "missing.index"), device="cpu")
    h = C.c_void_p()
    rc = _lib.lib().rvcmi_ivf_create_from_file(str(tmp_path / "missing.index").encode(), 0, C.byref(h))
    assert rc == -3 and b"cannot open" in _lib.lib().rvcmi_last_error()
    bad = tmp_path / "bad.index"
    bad.write_bytes(b"IxF2" + b"\0" * 64)
    rc = _lib.lib().rvcmi_ivf_create_from_file(str(bad).encode(), 0, C.byref(h))
    assert rc == -3 and b"IwFl" in _lib.lib().rvcmi_last_error()
    assert _lib.lib().rvcmi_nsf_create(None, None, 0, 0, 1, 1, C.byref(h)) == -1
    assert _lib.lib().rvcmi_front_create(None, None, 0, 0, 1, 1, C.byref(h)) == -1


def test_config_marshalling():
    from oracle.nsf_oracle import CONFIGS
    from rvc_amd.nsf import _cfg_struct

    c = _cfg_struct(vars(CONFIGS["v1_32k"]), "fp16")
    assert c.n_ups == 5 and list(c.upsample_rates)[:5] == [10, 4, 2, 2, 2] and c.operand == 2
    assert c.n_resblock_kernels == 3 and list(c.resblock_dilation_sizes[2])[:3] == [1, 3, 5] and c.sr == 32000
    with pytest.raises(ValueError):
        _cfg_struct(vars(CONFIGS["v1_32k"]), "int8")


def test_shard_range_partitions_exactly():
    from rvc_amd.dist import shard_range

    for n in (0, 1, 7, 64, 512, 513):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, n_clips):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rvc_amd.dist import broadcast_bytes, shard_range

    payload = None
    if rank == 0:
        g = torch.Generator().manual_seed(7)
        payload = torch.randint(0, 256, (100003,), dtype=torch.uint8, generator=g)
    got = broadcast_bytes(payload, src=0, device=torch.device("cpu"))
    g = torch.Generator().manual_seed(7)
    expect = torch.randint(0, 256, (100003,), dtype=torch.uint8, generator=g)
    assert torch.equal(got, expect)
    lo, hi = shard_range(n_clips, rank, world)
    cover = torch.zeros(n_clips, dtype=torch.int64)
    cover[lo:hi] = 1
    dist.all_reduce(cover)
    assert torch.all(cover == 1)  # every clip converted exactly once, no cross-rank dependency afterwards
    dist.destroy_process_group()


def test_multi_rank_broadcast_and_sharding_gloo():
    import torch.multiprocessing as mp

    mp.spawn(_gloo_worker, args=(2, _free_port(), 67), nprocs=2, join=True)


def _convert_batch_worker(rank, world, port, n_files):
    import torch.distributed as dist

    from rvc_amd.dist import convert_batch, shard_range

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    paths = ["clip_%02d.wav" % i for i in range(n_files)]
    seen = []

    def fake_convert(path, index):  # stands for vc_single: (info, (sr, audio)); one file fails like a corrupt input would
        seen.append(path)
        if path == "clip_03.wav":
            raise ValueError("cannot decode %s" % path)
        return "Success", (48000, np.full(4, int(path[5:7]), np.int16)), rank, index

    res = convert_batch(paths, fake_convert, index="added.index", gather=True, device=torch.device("cpu"))
    lo, hi = shard_range(n_files, rank, world)
    assert seen == paths[lo:hi], "rank %d converted %s" % (rank, seen)          # exactly its contiguous shard, in order
    if rank == 0:
        assert [p for p, _ in res] == paths                                      # all files, input order
        for i, (p, r) in enumerate(res):
            if p == "clip_03.wav":
                assert isinstance(r, str) and "cannot decode clip_03.wav" in r    # the traceback text, not an exception
                continue
            owner = next(k for k in range(world) if shard_range(n_files, k, world)[0] <= i < shard_range(n_files, k, world)[1])
            assert r[0] == "Success" and r[2] == owner and r[3] == "added.index" and int(r[1][1][0]) == i
    else:
        assert [p for p, _ in res] == paths[lo:hi]
    own = convert_batch(paths, lambda p, ix: p.upper(), gather=False, device=torch.device("cpu"))
    assert own == [(p, p.upper()) for p in paths[lo:hi]]
    # convert_many: the rank's shard in ONE call (rvc_amd.pipeline.convert_files batches the files on the GPU); a failing batch is
    # retried file by file so that the failure stays with its file
    batches = []

    def many(ps, index):
        batches.append(list(ps))
        if "clip_03.wav" in ps:
            raise ValueError("cannot decode clip_03.wav")
        return [p.upper() for p in ps]

    own = convert_batch(paths, index="added.index", gather=False, device=torch.device("cpu"), convert_many=many)
    assert batches[0] == paths[lo:hi]
    exp = [(p, p.upper()) for p in paths[lo:hi]]
    if "clip_03.wav" in paths[lo:hi]:
        assert batches[1:] == [[p] for p in paths[lo:hi]]
        assert [r for p, r in own if p != "clip_03.wav"] == [p.upper() for p in paths[lo:hi] if p != "clip_03.wav"]
        assert "cannot decode" in dict(own)["clip_03.wav"]
    else:
        assert own == exp and len(batches) == 1
    with pytest.raises(ValueError):
        convert_batch(paths, gather=False, device=torch.device("cpu"))
    # what may be GATHERED is bounded: a waveform-sized array inside ONE rank's results raises on EVERY rank before the collective
    # (nobody is left waiting in gather_object); the same results stay legal with gather=False
    big = lambda p, ix: ("Success", (48000, np.zeros(480000, np.int16))) if p == "clip_05.wav" else "Success"
    with pytest.raises(ValueError, match="MAX_GATHER_ELEMENTS"):
        convert_batch(paths, big, gather=True, device=torch.device("cpu"))
    kept = convert_batch(paths, big, gather=False, device=torch.device("cpu"))
    assert [p for p, _ in kept] == paths[lo:hi]
    dist.barrier()
    dist.destroy_process_group()


def test_convert_batch_shards_files_over_ranks_gloo():
    """``rvc_amd.dist.convert_batch`` = vc_multi (infer/modules/vc/modules.py:201-266) over the ranks of a node: contiguous shards,
    a per-file failure becomes that file's result, rank 0 ends up with every result in input order; world 1 = the plain loop."""
    import torch.multiprocessing as mp

    from rvc_amd.dist import convert_batch

    assert convert_batch(["a", "b"], lambda p, ix: (p, ix), index=7) == [("a", ("a", 7)), ("b", ("b", 7))]
    mp.spawn(_convert_batch_worker, args=(2, _free_port(), 7), nprocs=2, join=True)


def _run_bench(args, env_extra=None, timeout=300):
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=timeout)


def test_bench_gpus_flag_launches_that_many_ranks_gloo():
    """`bench.py --gpus 2` without a launcher must become 2 ranks (here: the gloo dry run of the same skeleton); the line's
    n_gpus is the process group's world size, the blob is the one rank 0 made, the shards cover the clip list."""
    import json

    r = _run_bench(["--gpus", "2", "--dist-selftest"])
    assert r.returncode == 0, r.stderr[-800:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_agree"] and d["shards_cover"] and d["index_blob_bytes"] == 1 << 20
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr


def test_bench_config3_world_8_dry_run_gloo():
    """BASELINE configs[3]'s partition without hardware: `bench.py --gpus 8 --config 3 --dist-selftest` self-launches 8 gloo ranks,
    64 clips per rank (512 in all), ONE broadcast of rank 0's blob, `ranks_agree` on the bytes and on a value derived from them,
    rank 0 alone prints the line.  (No scaling curve is simulated: the line carries no throughput.)"""
    import json

    r = _run_bench(["--gpus", "8", "--config", "3", "--dist-selftest"], timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_agree"] and d["shards_cover"] and d["clips"] == 512 and d["clips_per_rank"] == [64] * 8
    assert "value" not in d and "--nproc-per-node=8" in r.stderr


def test_bench_refuses_a_mislabelled_multi_gpu_run():
    """Fewer GPUs than --gpus, or a launcher whose WORLD_SIZE disagrees with --gpus: non-zero exit, no JSON line."""
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])  # no GPU in the build container; 1 GPU on the test box
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("2+ GPUs visible: the real launch would run")
    assert r.returncode != 0 and "--gpus 2 requested but only" in r.stderr and "{" not in r.stdout
    r = _run_bench(["--gpus", "2", "--dist-selftest"], {"WORLD_SIZE": "4", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and "{" not in r.stdout


def test_torchrun_argv_and_ranks_agree_helpers():
    from rvc_amd.dist import torchrun_argv

    av = torchrun_argv("/x/bench.py", ["--gpus", "8"], 8, master_port=29511)
    assert av[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in av and av[-3:] == ["/x/bench.py", "--gpus", "8"]
    assert av[av.index("--master-addr") + 1] == "127.0.0.1" and av[av.index("--master-port") + 1] == "29511"


def _agree_worker(rank, world, port):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rvc_amd.dist import ranks_agree

    same = torch.arange(40, dtype=torch.float32)
    assert ranks_agree(same)
    diff = same.clone()
    if rank == 1:
        diff[7] = float(np.nextafter(np.float32(7.0), np.float32(8.0)))  # one ulp on one rank must be seen by EVERY rank
    assert not ranks_agree(diff)
    assert ranks_agree(torch.arange(9, dtype=torch.int64))
    dist.destroy_process_group()


def test_ranks_agree_detects_a_one_ulp_difference_gloo():
    import torch.multiprocessing as mp

    mp.spawn(_agree_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_reference_module_introspection_when_reference_is_present():
    """config_from_reference / _plain_state_dict against the REAL reference classes (skipped on the GPU box)."""
    ref = os.environ.get("RVC_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "rvc")):
        pytest.skip("reference checkout not mounted")
    import sys

    sys.path.insert(0, ref)
    from rvc.layers.nsf import NSFGenerator

    from oracle.nsf_oracle import CONFIGS
    from rvc_amd.nsf import _plain_state_dict, config_from_reference

    cfg = CONFIGS["v1_40k"]
    net = NSFGenerator(cfg.inter_channels, "1", cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes, cfg.upsample_rates,
                       cfg.upsample_initial_channel, cfg.upsample_kernel_sizes, cfg.gin_channels, cfg.sr)
    with pytest.raises(ValueError):
        _plain_state_dict(net)  # weight norm still attached
    net.remove_weight_norm()
    got = config_from_reference(net)
    for k in ("inter_channels", "upsample_rates", "upsample_kernel_sizes", "upsample_initial_channel", "gin_channels", "sr",
              "resblock_kernel_sizes", "resblock_dilation_sizes", "use_f0"):
        assert got[k] == getattr(cfg, k), k
    sd = _plain_state_dict(net)
    assert "ups.0.weight" in sd and sd["ups.0.weight"].shape == (512, 256, 16)


def test_front_introspection_and_infer_patch_when_reference_is_present():
    """front_config_from_reference against the REAL reference synthesizer (skipped on the GPU box)."""
    ref = os.environ.get("RVC_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "rvc")):
        pytest.skip("reference checkout not mounted")
    import sys

    sys.path.insert(0, ref)
    from rvc.layers.synthesizers import SynthesizerTrnMsNSFsid

    from oracle.front_oracle import FrontConfig

    cl = [1025, 32, 192, 192, 768, 2, 6, 3, 0, "1", [3, 7, 11], [[1, 3, 5]] * 3, [10, 10, 2, 2], 512, [16, 16, 4, 4], 109, 256, 40000]
    net = SynthesizerTrnMsNSFsid(*cl, encoder_dim=256, use_f0=True)
    got = rvc_amd.front_config_from_reference(net)
    want = vars(FrontConfig(in_channels=256))
    for k, v in got.items():
        assert want[k] == v, k
    with pytest.raises(ValueError):
        rvc_amd.FrontHIP.from_reference(net, device="cuda:0")  # weight norm still attached


@pytest.mark.parametrize("name", ["infer_full_v2_48k_T40", "infer_full_v2_48k_rt"])
def test_infer_hip_orchestration_against_reference_golden_on_cpu(name):
    """rvc_amd.infer_hip (the replacement of SynthesizerTrnMsNSFsid.infer, synthesizers.py:160-203) driven with oracle-backed
    stand-ins for the HIP front / generator must reproduce the REFERENCE's net_g.infer waveform: this pins its slicing
    (flow_head = max(skip_head - 24, 0), dec_head, pitchf window, return_length2 -> n_res) without a GPU."""
    from conftest import load_golden
    from oracle import front_oracle, nsf_oracle, synth
    from oracle.front_oracle import FrontConfig

    d = load_golden(name)
    cfg, fcfg = nsf_oracle.CONFIGS["v2_48k"], FrontConfig()
    wd, wf = synth.make_dec_weights(cfg, int(d["seed"])), synth.make_front_weights(fcfg, int(d["seed"]))

    class Front:
        cfg = {"use_f0": True}

        def __call__(self, phone, pitch, lengths, g, flow_head=0, *, noise=None):
            sid = torch.tensor([int(d["sid"][0])])
            z, m1, _ = front_oracle.infer_front(fcfg, wf, phone, pitch, lengths, sid, noise, flow_head if flow_head else None)
            return z * m1

    class Net:
        emb_g = staticmethod(lambda sid: wf["emb_g.weight"][sid])

        @staticmethod
        def dec(z, pitchf, g=None, n_res=None, noise=None):
            return nsf_oracle.generator_forward(cfg, wd, z, pitchf, g, noise, n_res=n_res)

    T = d["phone"].shape[1]
    opt = lambda k: None if int(d[k]) < 0 else int(d[k])
    with torch.no_grad():
        out = rvc_amd.infer_hip(Net(), Front(), torch.from_numpy(d["phone"]), torch.tensor([T]), torch.from_numpy(d["sid"]),
                                torch.from_numpy(d["pitch"]), torch.from_numpy(d["pitchf"]), opt("skip_head"), opt("return_length"),
                                opt("return_length2"), noise_zp=torch.from_numpy(d["noise_zp"]), noise_dec=torch.from_numpy(d["noise_dec"]))
    assert out.shape == d["out"].shape
    assert np.abs(out.numpy() - d["out"]).max() < 2e-5


def test_install_rebinds_names_without_editing_the_callers():
    """rvc_amd.install() on the compute-free RVC skeleton (tests/skeleton): names bound before and after the call, the faiss
    shim behind `import faiss`, and a clean uninstall.  (Device work is covered by tests/test_gpu_dropin.py.)"""
    import sys

    skel = os.path.join(os.path.dirname(os.path.abspath(__file__)), "skeleton")
    if os.path.isdir(os.path.join(os.environ.get("RVC_REFERENCE", "/root/reference"), "rvc")):
        for m in [m for m in sys.modules if m.split(".")[0] in ("rvc", "infer")]:
            del sys.modules[m]  # a previous test imported the real reference package
    had_faiss = sys.modules.get("faiss")
    sys.path.insert(0, skel)
    try:
        import infer.modules.vc.modules as vm
        import rvc.synthesizer as rs

        orig = rs.get_synthesizer
        rvc_amd.install()
        rvc_amd.install()
        assert rs.get_synthesizer is not orig and rs.get_synthesizer._rvcmi_original is orig
        assert vm.get_synthesizer is rs.get_synthesizer and vm.load_synthesizer is rs.load_synthesizer
        import infer.modules.vc.pipeline as pl

        assert type(pl.faiss).__name__ == "_FaissShim"
        with pytest.raises(rvc_amd.RvcmiError):
            pl.load_index("/nonexistent/added.index")  # routed to the HIP reader (which fails loudly)
        if had_faiss is None:
            with pytest.raises(AttributeError, match="faiss is not installed"):
                pl.faiss.index_factory
        rvc_amd.uninstall()
        assert rs.get_synthesizer is orig and vm.get_synthesizer is orig and sys.modules.get("faiss") is had_faiss
    finally:
        rvc_amd.uninstall()
        sys.path.remove(skel)
        for m in [m for m in sys.modules if m.split(".")[0] in ("rvc", "infer")]:
            del sys.modules[m]


def test_bf16_operands_warn_before_anything_else():
    """operand='bf16' misses the 1e-3 bar: constructing with it warns -- even where the constructor then
    fails for lack of a GPU."""
    from oracle import nsf_oracle, synth

    cfg = nsf_oracle.CONFIGS["v1_40k"]
    with pytest.warns(RuntimeWarning, match="bf16"):
        with pytest.raises(Exception):
            rvc_amd.NSFGeneratorHIP(vars(cfg), {}, device="cuda:0" if torch.cuda.is_available() else "cuda:0", operand="bf16")


def test_realtime_pitch_cache_and_frame_arithmetic():
    """rtrvc.py:203-219 on the host: the f0 window length per method and the rolling pitch caches."""
    import rvc_amd

    assert rvc_amd.f0_extractor_frame(4000, "fcpe") == 4800
    assert rvc_amd.f0_extractor_frame(4000, "rmvpe") == 5120 - 160 and rvc_amd.f0_extractor_frame(5000, "rmvpe") == 2 * 5120 - 160
    pc = rvc_amd.PitchCache("cpu")
    cp = np.zeros(1024, dtype=np.int64)
    cf = np.zeros(1024, dtype=np.float32)
    rng = np.random.default_rng(0)
    for blk in range(6):
        n = int(rng.integers(20, 60))
        pitch = torch.from_numpy(rng.integers(1, 255, size=n))
        pitchf = torch.from_numpy(rng.uniform(50, 800, size=n).astype(np.float32))
        block = int(rng.choice([1600, 4000, 8000]))
        pc.update(pitch, pitchf, block)
        shift = block // 160
        cp[:-shift] = cp[shift:].copy()
        cf[:-shift] = cf[shift:].copy()
        cp[4 - n:] = pitch.numpy()[3:-1]
        cf[4 - n:] = pitchf.numpy()[3:-1]
        assert np.array_equal(pc.pitch.numpy(), cp) and np.array_equal(pc.pitchf.numpy(), cf)
    p, f = pc.window(100, 25, 29)
    assert p.shape == (1, 100) and np.allclose(f[0].numpy(), cf[-100:] * 29 / 25)
    with pytest.raises(ValueError):
        pc.update(torch.zeros(3, dtype=torch.long), torch.zeros(3), 1600)


def test_infer_segments_batches_pads_and_draws_noise_like_sequential_calls():
    """``rvc_amd.pipeline.infer_segments`` (host logic, CPU tensors, a recording stand-in for ``net_g.infer``): segments are grouped into
    as few ragged batches as MAX_BATCH_FRAMES allows, padded with zeros / pitch 1, every item's noise is drawn in the order and shapes of
    sequential ``infer`` calls (randn(1, 192, T); rand(1, 1, 1); randn(1, T * upp, 1)), and each output is cut to its own length."""
    import types

    import rvc_amd.pipeline as pl

    upp, IC, d = 4, 192, 8
    calls = []

    def infer(phone, lengths, sid, pitch=None, pitchf=None, noise_zp=None, noise_dec=None, ragged=False):
        assert ragged and phone.shape[0] == lengths.numel() == sid.numel()
        calls.append(dict(lens=lengths.tolist(), phone=phone.clone(), pitch=pitch.clone(), pitchf=pitchf.clone(), nz=noise_zp.clone(), nd=noise_dec.clone()))
        B, T = phone.shape[:2]
        out = torch.zeros(B, 1, T * upp)
        for b in range(B):
            out[b, 0] = 100.0 * (len(calls) - 1) + b + torch.arange(T * upp) / 1000.0
        return out

    infer._rvcmi_ragged = True
    net_g = types.SimpleNamespace(infer=infer, dec=types.SimpleNamespace(upp=upp, cfg={"inter_channels": IC}))
    # a ragged-capable `infer` is not enough: the generator behind it must be the HIP one, which alone can be told the lengths
    # (the positive case runs on the GPU: tests/test_gpu_dropin.py drives the batch path through it)
    assert not pl._ragged_capable(net_g) and not pl._ragged_capable(types.SimpleNamespace(infer=lambda *a, **k: None))
    lens = [5, 9, 3, 7]
    items = []
    for i, T in enumerate(lens):
        items.append((torch.full((1, T, d), float(i + 1)), torch.full((1, T), 10 + i, dtype=torch.long), torch.full((1, T), 100.0 + i), T))
    old = pl.MAX_BATCH_FRAMES
    try:
        pl.MAX_BATCH_FRAMES = 18  # padded frames: 2 x 9 fit, 3 x 9 do not; then 2 x 7
        torch.manual_seed(77)
        outs = pl.infer_segments(net_g, torch.tensor([3]), items)
    finally:
        pl.MAX_BATCH_FRAMES = old
    assert [c["lens"] for c in calls] == [[5, 9], [3, 7]]
    torch.manual_seed(77)  # the sequential draws, item by item
    k = 0
    for c in calls:
        Tm = max(c["lens"])
        assert c["phone"].shape == (len(c["lens"]), Tm, d)
        for b, T in enumerate(c["lens"]):
            nz = torch.randn(1, IC, T)
            torch.rand(1, 1, 1)
            nd = torch.randn(1, T * upp, 1)
            assert torch.equal(c["nz"][b, :, :T], nz[0]) and not c["nz"][b, :, T:].any()
            assert torch.equal(c["nd"][b, :T * upp], nd[0, :, 0]) and not c["nd"][b, T * upp:].any()
            assert (c["phone"][b, :T] == k + 1).all() and not c["phone"][b, T:].any()
            assert (c["pitch"][b, :T] == 10 + k).all() and (c["pitch"][b, T:] == 1).all() and (c["pitchf"][b, :T] == 100.0 + k).all()
            k += 1
    for i, (o, T) in enumerate(zip(outs, lens)):
        call, b = (0, i) if i < 2 else (1, i - 2)
        assert o.shape == (T * upp,) and torch.equal(o, 100.0 * call + b + torch.arange(T * upp) / 1000.0)


def test_convert_files_host_logic_fallback_and_argument_checks():
    """``rvc_amd.pipeline.convert_files`` without a GPU: a synthesizer that is not the HIP one (or an index only real faiss reads) takes
    the plain per-file loop over ``self.pipeline`` with the reference's argument order; one f0_file per input; if_f0 == 2 (ONE
    precomputed pitch pair) is refused for several inputs."""
    import types

    import rvc_amd.pipeline as pl

    calls = []

    def pipeline(model, net_g, sid, audio, times, f0_up_key, f0_method, file_index, index_rate, if_f0, filter_radius, tgt_sr, resample_sr,
                 rms_mix_rate, version, protect, f0_file=None):
        calls.append((len(audio), f0_up_key, f0_method, file_index, index_rate, if_f0, filter_radius, tgt_sr, resample_sr, rms_mix_rate, version,
                      protect, f0_file))
        return np.full(3, float(len(audio)), np.float32)

    self_ = types.SimpleNamespace(pipeline=pipeline, device="cpu")
    net_g = types.SimpleNamespace(infer=lambda *a, **k: None)  # not ragged-capable
    audios = [np.zeros(5, np.float32), np.zeros(9, np.float32)]
    out = pl.convert_files(self_, "hubert", net_g, 3, audios, [0, 0, 0], -2, "rmvpe", "", 0.75, 1, 3, 48000, 0, 0.25, "v2", 0.33, f0_files=["a", None])
    assert [o[0] for o in out] == [5.0, 9.0]
    assert calls == [(5, -2, "rmvpe", "", 0.75, 1, 3, 48000, 0, 0.25, "v2", 0.33, "a"), (9, -2, "rmvpe", "", 0.75, 1, 3, 48000, 0, 0.25, "v2", 0.33, None)]
    with pytest.raises(ValueError, match="one entry per input"):
        pl.convert_files(self_, "hubert", net_g, 3, audios, [0, 0, 0], 0, "rmvpe", "", 0.75, 1, 3, 48000, 0, 0.25, "v2", 0.33, f0_files=[None])
    with pytest.raises(ValueError, match="if_f0 == 2"):
        pl.convert_files(self_, "hubert", net_g, 3, audios, [0, 0, 0], 0, (None, None), "", 0.75, 2, 3, 48000, 0, 0.25, "v2", 0.33)


def test_blend_segments_concatenates_and_slices_like_the_per_segment_calls(monkeypatch):
    """``rvc_amd.pipeline.blend_segments`` host logic (CPU tensors, a recording stand-in for the one-launch search + blend + x2 + protect
    kernel): the HuBERT frames of several segments go through ONE call per ``MAX_BATCH_QUERIES`` frames, every segment's frame-rate pitchf
    sits at ITS output rows (2 x its first query row), and each segment gets back exactly rows [2 o, 2 o + p_len) of the result."""
    import rvc_amd.glue as glue
    import rvc_amd.pipeline as pl

    calls = []

    def fake(out, F, index, index_rate, pf, protect, guard):
        calls.append((int(F.shape[0]), index, index_rate, None if pf is None else pf.clone(), protect, guard))
        n = F.shape[0]
        out[:] = torch.repeat_interleave(F, 2, dim=0)[: out.shape[0]] + (0 if pf is None else pf[: out.shape[0], None])

    monkeypatch.setattr(glue, "_blend_expand_into", fake)
    d = 4
    raw, want = [], []
    for i, (nq, p_len) in enumerate(((5, 9), (3, 6), (7, 13))):
        f = torch.full((1, nq, d), float(i + 1)) + torch.arange(nq, dtype=torch.float32)[None, :, None] / 10
        pitchf = torch.full((1, p_len), 100.0 * (i + 1))
        raw.append((f, torch.zeros(1, p_len, dtype=torch.long), pitchf, p_len))
        want.append(torch.repeat_interleave(f[0], 2, dim=0)[:p_len] + 100.0 * (i + 1))
    items = pl.blend_segments(raw, "IDX", 0.75, 0.33)
    assert len(calls) == 1 and calls[0][0] == 15 and calls[0][1] == "IDX" and calls[0][4] == 0.33 and calls[0][5] is False
    pf = calls[0][3]
    assert pf.shape == (30,) and torch.equal(pf[0:9], torch.full((9,), 100.0)) and pf[9] == 1.0  # (padding rows: voiced = untouched)
    assert torch.equal(pf[10:16], torch.full((6,), 200.0)) and torch.equal(pf[16:29], torch.full((13,), 300.0))
    for (feats, pt, pff, p_len), w, r in zip(items, want, raw):
        assert feats.shape == (1, p_len, d) and torch.equal(feats[0], w) and pt is r[1] and pff is r[2]
    # bounded calls: the same slices from several retrieval calls; no index -> the plain x2 (+ protect) kernel, protect >= 0.5 -> no pitchf
    monkeypatch.setattr(pl, "MAX_BATCH_QUERIES", 8)
    del calls[:]
    items2 = pl.blend_segments(raw, None, 0.75, 0.5)
    assert [c[0] for c in calls] == [8, 7] and all(c[1] is None and c[3] is None for c in calls)
    for (feats, _, _, p_len), r in zip(items2, raw):
        assert torch.equal(feats[0], torch.repeat_interleave(r[0][0], 2, dim=0)[:p_len])


def test_index_cache_of_the_rebound_pipeline(monkeypatch, tmp_path):
    """``rvc_amd.pipeline._open_index`` keeps the parsed index of a path resident between calls (the reference re-reads the file in every
    ``pipeline()`` call, pipeline.py:205-218): same object for an unchanged file, a new read when size / mtime change, at most
    ``INDEX_CACHE_ENTRIES`` files resident (oldest out), ``RVCMI_INDEX_CACHE=0`` = read per call, index_rate 0 / missing file = no index."""
    import types

    import rvc_amd.pipeline as pl
    from rvc_amd import ivf

    reads = []
    monkeypatch.setattr(ivf, "read_index", lambda path, device=None: (reads.append(path), object())[1])
    monkeypatch.setattr(pl, "_INDEX_CACHE", {})
    self = types.SimpleNamespace(device="cpu")
    a, b, c = [str(tmp_path / n) for n in ("a.index", "b.index", "c.index")]
    for p_ in (a, b, c):
        open(p_, "wb").write(b"x" * 10)
    assert pl._open_index(self, a, 0.0) == (None, False) and pl._open_index(self, str(tmp_path / "nope.index"), 0.75) == (None, False)
    i1, _ = pl._open_index(self, a, 0.75)
    i2, _ = pl._open_index(self, a, 0.75)
    assert i1 is i2 and reads == [a]
    open(a, "wb").write(b"y" * 11)  # another size (and mtime): a different index behind the same name
    i3, _ = pl._open_index(self, a, 0.75)
    assert i3 is not i1 and reads == [a, a]
    pl._open_index(self, b, 0.75)
    pl._open_index(self, c, 0.75)  # a third file: the oldest entry leaves
    assert len(pl._INDEX_CACHE) == pl.INDEX_CACHE_ENTRIES and reads == [a, a, b, c]
    assert pl._open_index(self, c, 0.75)[0] is pl._open_index(self, c, 0.75)[0] and len(reads) == 4
    monkeypatch.setenv("RVCMI_INDEX_CACHE", "0")
    assert pl._open_index(self, c, 0.75)[0] is not pl._open_index(self, c, 0.75)[0] and len(reads) == 6


def test_e2e_rmvpe_proxy_has_the_reference_architecture():
    """``tools/e2e_proxies.py`` (the stand-in ``bench.py --e2e`` times for the RMVPE network): same parameter count and the same FLOPs as the
    reference's own ``E2E(4, 1, (2, 2))`` (rvc/f0/e2e.py, rvc/f0/models.py), output [1, T, 360] in (0, 1).  Skipped on the GPU box."""
    ref = os.environ.get("RVC_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "rvc")):
        pytest.skip("reference checkout not mounted")
    import sys
    import types

    from torch.utils.flop_counter import FlopCounterMode

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, ref)
    sys.modules.setdefault("numba", types.SimpleNamespace(jit=lambda *a, **k: (lambda f: f), njit=lambda *a, **k: (lambda f: f)))
    try:
        from e2e_proxies import _SalienceNet
        from rvc.f0.e2e import E2E

        theirs, ours = E2E(4, 1, (2, 2)).eval(), _SalienceNet().eval()
        assert sum(p.numel() for p in ours.parameters()) == sum(p.numel() for p in theirs.parameters()) == 90423165
        mel = torch.randn(1, 128, 64)
        fl = []
        for m in (theirs, ours):
            with FlopCounterMode(display=False) as fc, torch.no_grad():
                y = m(mel)
            fl.append(fc.get_total_flops())
            assert y.shape == (1, 64, 360) and float(y.min()) > 0 and float(y.max()) < 1
        assert fl[0] == fl[1]
        # the one layer that runs on csrc/gru.hip: the reference's own E2E holds it where accelerate_rmvpe looks (fc[0].gru), in a shape GRUHIP
        # supports; on a CPU model nothing is swapped
        from rvc_amd.gru import accelerate_rmvpe, supports

        g = theirs.fc[0].gru
        assert supports(g) and (g.input_size, g.hidden_size) == (384, 256) and supports(ours.gru)
        assert accelerate_rmvpe(theirs) == 0 and theirs.fc[0].gru is g
    finally:
        sys.path.remove(ref)
        sys.path.remove(os.path.join(ROOT, "tools"))
        for m in [m for m in sys.modules if m.split(".")[0] in ("rvc", "e2e_proxies")]:
            del sys.modules[m]
