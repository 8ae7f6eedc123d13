#!/usr/bin/env python
"""Headline benchmark: real-time factor of the RVC hot path (IVF retrieval + NSF-HiFi-GAN decode)
on 10 s clips, v2 / 48 kHz, synthetic data, one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of B clips per rank: 599 HuBERT-shaped queries per
clip through search(k=8)+blend against the index, then the generator on T=1198 frames (what a 10 s
clip costs inside the pipeline with x_pad=1).  Inputs are resident in HBM before
the timed region.  Rank 0 prints ONE JSON line.

What is imported from ``oracle/`` and why: ``oracle.synth`` / the config dataclasses only GENERATE the seeded synthetic
weights, features, f0 and index (no checkpoints or datasets exist offline); nothing under ``oracle/`` executes inside a
timed region except in the ``cpu_baseline`` leg, which times the CPU restatement itself.  Every timed step calls the
HIP library through ``rvc_amd`` only.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CLIP_SECONDS = 10.0
T_CLIP = 1198        # vocoder frames per 10 s clip inside the pipeline (pipeline.py:146-148, x_pad=1)
NQ_CLIP = 599        # HuBERT frames = retrieval queries per clip
GEN_FLOP_PER_CLIP = 1319.6e9  # SURVEY.md 8d, T=1198
PEAK = {"bf16": 2.5e15, "fp16": 2.5e15, "fp32": 1.573e14}  # MI355X dense MFMA / fp32 peaks (MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--repeats", type=int, default=10, help="extra repetitions of the K-step loop for the median / spread report")
    p.add_argument("--batch", type=int, default=None, help="clips per rank per step (default 1 = BASELINE configs[1]; 64 = configs[2])")
    p.add_argument("--config", type=int, default=None, choices=[1, 2, 3],
                   help="BASELINE configs[N] preset: 1 = one clip, 10000x768 index; 2 = 64 clips, one GPU; 3 = 64 clips PER RANK against a "
                        "1M x 256 index built on rank 0 and broadcast once over RCCL (configs[3] is 512 clips on 8 GPUs)")
    p.add_argument("--operand", default="fp16", choices=["fp16", "bf16", "fp32"])
    p.add_argument("--frames", type=int, default=T_CLIP)
    p.add_argument("--index-n", type=int, default=None)
    p.add_argument("--index-d", type=int, default=None)
    p.add_argument("--dist-selftest", action="store_true",
                   help="CPU/gloo dry run of the multi-rank skeleton (self-launch, rendezvous, blob broadcast, rank agreement); no GPU work")
    p.add_argument("--index-rate", type=float, default=0.75)
    p.add_argument("--graph", type=int, default=1, help="replay the step from a captured hipGraph")
    p.add_argument("--whole", action="store_true",
                   help="time the whole net_g.infer (retrieval + enc_p + flow^-1 + decode) as THE step instead of the BASELINE hot path")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-gpu-torch-baseline", action="store_true")
    p.add_argument("--torch-gpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--no-extra", action="store_true", help="skip the batch64 / stream / bf16 objects of the default N=1 line")
    p.add_argument("--e2e", action="store_true",
                   help="end-to-end Pipeline.convert_files with HuBERT / RMVPE architecture proxies on PyTorch-ROCm: wall per clip and its split, 1 and N files")
    p.add_argument("--e2e-files", type=int, default=64)
    p.add_argument("--stream", action="store_true",
                   help="BASELINE configs[4]: realtime chunks (v1/40k generator, T=31 frames -> n_res, 16 queries); prints p50 latency")
    a = p.parse_args()
    preset = {None: (1, 10000, 768), 1: (1, 10000, 768), 2: (64, 10000, 768), 3: (64, 1000000, 256)}[a.config]
    a.batch = preset[0] if a.batch is None else a.batch
    a.index_n = preset[1] if a.index_n is None else a.index_n
    a.index_d = preset[2] if a.index_d is None else a.index_d
    return a


def ensure_world(a):
    """``--gpus N`` MEANS N ranks.  Under a launcher (WORLD_SIZE set) the two must agree; without one and N > 1 this process
    replaces itself with ``python -m torch.distributed.run --nproc-per-node N ... bench.py <same args>`` (one rank per GPU,
    rendezvous on 127.0.0.1); with fewer than N GPUs visible it exits non-zero.  It never silently measures one GPU."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != a.gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks; refusing to report a mislabelled run" % (a.gpus, ws))
        return
    if a.gpus <= 1:
        return
    if not a.dist_selftest:
        ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if ndev < a.gpus:
            raise SystemExit("bench.py: --gpus %d requested but only %d GPU(s) are visible; not running a smaller job under that label" % (a.gpus, ndev))
    from rvc_amd.dist import torchrun_argv

    argv = torchrun_argv(os.path.abspath(__file__), sys.argv[1:], a.gpus)
    print("[bench] self-launching %d ranks: %s" % (a.gpus, " ".join(argv)), file=sys.stderr)
    sys.stderr.flush()
    os.execv(argv[0], argv)


def dist_selftest(a):
    """The multi-rank skeleton of main() without GPU work, on gloo: rendezvous, ONE broadcast of a blob only rank 0 knows
    (`rvc_amd.dist.broadcast_bytes`, what `broadcast_index` moves), `ranks_agree` on the received bytes AND on a value derived from
    them (the stand-in for every rank's first search), contiguous shards of the clip list (`--config 3`: 64 clips per rank, i.e.
    512 on 8 ranks = BASELINE configs[3]'s partition; otherwise a count that does not divide), barrier + max-over-ranks timing."""
    import torch.distributed as dist
    from rvc_amd.dist import broadcast_bytes, ranks_agree, shard_range

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dist.init_process_group(backend="gloo")
    blob = None
    if rank == 0:
        g = torch.Generator().manual_seed(4321)
        blob = torch.randint(0, 256, (1 << 20,), dtype=torch.uint8, generator=g)
    dist.barrier()
    t0 = time.perf_counter()
    got = broadcast_bytes(blob, src=0, device=torch.device("cpu"))
    t_b = time.perf_counter() - t0
    agree = ranks_agree(got)
    first = got.to(torch.float32).reshape(1024, 1024).sum(1)  # "first search": a function of THIS rank's copy of the blob
    agree = agree and ranks_agree(first)
    per_rank = 64 if a.config == 3 else None
    clips = per_rank * world if per_rank else 64 * world + 3
    lo, hi = shard_range(clips, rank, world)
    cover = torch.zeros(clips, dtype=torch.int64)
    cover[lo:hi] = 1
    dist.all_reduce(cover)
    sizes = torch.zeros(world, dtype=torch.int64)
    sizes[rank] = hi - lo
    dist.all_reduce(sizes)
    tt = torch.tensor([t_b], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ok = bool(agree) and bool((cover == 1).all()) and (per_rank is None or bool((sizes == per_rank).all()))
    if rank == 0:
        emit({"selftest": True, "backend": "gloo", "n_gpus": dist.get_world_size(), "index_blob_bytes": int(got.numel()),
              "index_broadcast_s": float(tt.item()), "ranks_agree": bool(agree), "shards_cover": bool((cover == 1).all()),
              "clips": clips, "clips_per_rank": [int(x) for x in sizes.tolist()], "scaling": "weak",
              "config": {"workload": "dry run of the N-rank skeleton%s: no GPU work" % (" of BASELINE configs[3] (64 clips per rank)" if per_rank else "")}})
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def _reference_generator(cfg, w):
    """The reference's own NSFGenerator (rvc/layers/nsf.py:64-206) carrying the seeded weights, when an RVC checkout is
    importable (RVC_REFERENCE, default /root/reference: present in the build container, absent on the GPU box)."""
    ref = os.environ.get("RVC_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "rvc", "layers")):
        return None
    try:
        sys.path.insert(0, ref)
        from rvc.layers.nsf import NSFGenerator

        net = NSFGenerator(cfg.inter_channels, "1", cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes, cfg.upsample_rates,
                           cfg.upsample_initial_channel, cfg.upsample_kernel_sizes, cfg.gin_channels, cfg.sr)
        net.eval()
        net.remove_weight_norm()
        net.load_state_dict(w, strict=True)
        return net
    except Exception as e:  # noqa
        print("[bench] reference generator not usable (%s); timing the oracle port" % e, file=sys.stderr)
        return None
    finally:
        if ref in sys.path:
            sys.path.remove(ref)


def cpu_baseline(cfg, w, idx, phone, z, f0, g, noise, index_rate):
    """The CPU side timed next to the GPU on this host's cores (3 warm-ups, median of 5).
    Generator: the reference's own ``NSFGenerator`` on torch-CPU fp32 when an RVC checkout is importable (kind
    'reference'), else the oracle restatement (kind 'port': the same ATen/oneDNN conv kernels the reference calls).
    Retrieval: the C restatement in fp32 arithmetic with OpenMP (faiss itself is not installable).
    Sample: ONE 10 s clip (T frames + 599 queries)."""
    import ctypes as C

    from oracle import nsf_oracle

    ref_net = _reference_generator(cfg, w)

    last = {}

    def gen_fwd(zz, ff, gg, nn):
        if ref_net is not None:
            return ref_net(zz, ff, g=gg)  # draws its own noise (same work)
        last["o"] = nsf_oracle.generator_forward(cfg, w, zz, ff, gg, nn)  # injected noise: comparable with the HIP output (parity leg)
        return last["o"]

    ncpu = os.cpu_count() or 1
    # torch's intra-op pool scales badly past a few dozen threads on this op mix (61 s/clip with 256 threads on
    # a 256-core host vs seconds with 32): probe a few pool sizes on a short clip and keep the fastest.
    best_t, cores = None, 1
    for n in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(n)
        with torch.no_grad():
            gen_fwd(z[:1, :, :64], f0[:1, :64], g[:1], noise[:1, :64 * cfg.upp])
            t0 = time.perf_counter()
            gen_fwd(z[:1, :, :160], f0[:1, :160], g[:1], noise[:1, :160 * cfg.upp])
            dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, cores = dt, n
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(min(ncpu, 64))
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libivf_oracle.so"))
    q = np.ascontiguousarray(phone[0].numpy())
    nq, d = q.shape
    D = np.empty((nq, 8), np.float32)
    I = np.empty((nq, 8), np.int64)
    P = np.empty((nq, 8), np.int64)
    pos_last = int(np.nonzero(idx["ids"] == idx["ntotal"] - 1)[0][0])
    vp = lambda a: a.ctypes.data_as(C.c_void_p)

    def once(Tn=None):
        Tn = z.shape[-1] if Tn is None else Tn
        t0 = time.perf_counter()
        feats = q.copy()
        lib.ivf_search(vp(feats), C.c_int64(nq), C.c_int(d), vp(idx["centroids"]), C.c_int64(idx["nlist"]), C.c_int(1),
                       vp(idx["list_offsets"]), vp(idx["ids"]), vp(idx["vecs"]), C.c_int(8), vp(D), vp(I), vp(P), C.c_int(1))
        lib.ivf_blend(vp(feats), C.c_int64(nq), C.c_int(d), vp(D), vp(P), C.c_int(8), vp(idx["vecs"]), C.c_int64(pos_last),
                      C.c_float(index_rate), C.c_float(1.0 - index_rate))
        t1 = time.perf_counter()
        with torch.no_grad():
            gen_fwd(z[:1, :, :Tn], f0[:1, :Tn], g[:1], noise[:1, :Tn * cfg.upp])
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1

    once(400), once(400), once()  # 3 warm-ups: two short clips (allocator / thread pool), one full clip
    runs = [once() for _ in range(5)]
    t_ivf = float(np.median([r[0] for r in runs]))
    t_gen = float(np.median([r[1] for r in runs]))
    tot = sorted(r[0] + r[1] for r in runs)
    kind = "reference" if ref_net is not None else "port"
    return {"_oracle_out": last.get("o") if last.get("o") is not None and last["o"].shape[-1] == z.shape[-1] * cfg.upp else None,
            "value": CLIP_SECONDS / (t_ivf + t_gen), "unit": "x real-time (audio-sec/wall-sec)", "cores": cores, "kind": kind,
            "spread_s": [tot[0], tot[-1]],
            "sample": "1 clip: 599 queries vs %dx%d IVF (C restatement, fp32, OpenMP) %.3fs + generator T=%d (%s, torch-CPU fp32) %.3fs; "
                      "3 warm-ups (2 short clips + 1 full), median of 5; torch threads chosen from {8,16,32,64}"
                      % (idx["ntotal"], idx["d"], t_ivf, z.shape[-1],
                         "the reference's rvc.layers.nsf.NSFGenerator" if ref_net is not None else "oracle restatement of NSFGenerator.forward",
                         t_gen)}


def torch_gpu_baseline_worker(a):
    """The "hipify-equivalent" number: the SAME generator executed by stock
    PyTorch-ROCm (MIOpen / rocBLAS kernels chosen by torch) on this GPU -- the oracle's functional restatement of
    NSFGenerator.forward moved to cuda, fp32 and .half() (the reference's is_half mode, infer/modules/vc/modules.py:94-95).
    Runs in a child process so that a slow first-time MIOpen kernel build cannot stall the headline run.  3 warm-ups, median
    of 5, inputs resident on the device, torch.cuda.synchronize() around each forward."""
    from oracle import nsf_oracle, synth

    quota = _cpu_quota()  # (the same CPU-pool cap as the main mode: the baseline must not be the one that gets throttled)
    if quota and torch.get_num_threads() > quota:
        torch.set_num_threads(quota)
    dev = torch.device("cuda", 0)
    cfg = nsf_oracle.CONFIGS["v2_48k"]
    w = synth.make_dec_weights(cfg, 1234)
    z, f0, g = synth.make_dec_inputs(cfg, 1, a.frames, seed=1234)
    noise = nsf_oracle.reference_noise(1, a.frames, cfg.upp, 114514)
    out = {"what": "oracle restatement of NSFGenerator.forward on device=cuda via stock PyTorch-ROCm (torch %s), generator only, B=1, T=%d; 3 warm-ups, median of 5"
                   % (torch.__version__, a.frames)}
    ref = None
    for name, dt in (("fp32", torch.float32), ("fp16", torch.float16)):
        wd = {k: v.to(dev, dt) for k, v in w.items()}
        zd, fd, gd, nd = z.to(dev, dt), f0.to(dev, dt), g.to(dev, dt), noise.to(dev, dt)
        ts = []
        with torch.no_grad():
            for i in range(8):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                o = nsf_oracle.generator_forward(cfg, wd, zd, fd, gd, nd)
                torch.cuda.synchronize()
                if i >= 3:
                    ts.append(time.perf_counter() - t0)
                elif i == 0:
                    out[name + "_first_call_s"] = time.perf_counter() - t0
        ts.sort()
        med = ts[len(ts) // 2]
        out[name] = {"ms_per_clip": 1e3 * med, "min_ms": 1e3 * ts[0], "max_ms": 1e3 * ts[-1],
                     "value": CLIP_SECONDS * (a.frames / T_CLIP) / med, "unit": "x real-time (generator only)",
                     "tflops": GEN_FLOP_PER_CLIP * (a.frames / T_CLIP) / med / 1e12, "finite": bool(torch.isfinite(o).all())}
        if ref is None:
            ref = o.float()
        else:
            out[name]["rms_vs_torch_fp32"] = float((o.float() - ref).pow(2).mean().sqrt())
    print("TORCH_GPU_BASELINE " + json.dumps(out))


def torch_gpu_baseline(a, timeout_s=420):
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--torch-gpu-baseline-worker", "--frames", str(a.frames)]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, text=True)
        for ln in r.stdout.splitlines():
            if ln.startswith("TORCH_GPU_BASELINE "):
                return json.loads(ln[len("TORCH_GPU_BASELINE "):])
        return {"error": "worker exited %d: %s" % (r.returncode, r.stderr[-300:])}
    except subprocess.TimeoutExpired:
        return {"error": "stock PyTorch-ROCm generator did not finish 2 x 8 forwards within %d s (first-time MIOpen kernel builds)" % timeout_s}


def _percentiles(lat):
    lat = sorted(lat)
    n = len(lat)
    return {"p50_ms": lat[n // 2], "p90_ms": lat[int(0.9 * n)], "p99_ms": lat[min(n - 1, int(0.99 * n))], "min_ms": lat[0], "chunks": n}


def _time_chunks(fn, n=300, warm=30, graph=True):
    """Per-chunk wall latency (sync - launch - sync).  With `graph`, the chunk is captured once (fixed (T, n_res, queries)
    bucket, inputs copied into static buffers by the graph's first nodes) and replayed, as a realtime loop would for the
    block size the GUI was started with (gui.py:783-840 fixes it)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    run, captured = fn, False
    if graph:
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            run, captured = g.replay, True
        except Exception as e:  # noqa
            print("[bench] stream graph capture failed (%s); eager launches" % e, file=sys.stderr)
            torch.cuda.synchronize()
    lat = []
    for i in range(n + warm):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        if i >= warm:
            lat.append(1e3 * (time.perf_counter() - t0))
    return _percentiles(lat), captured


def stream_mode(a, chunks=300, warm=30, index=None, breakdown=True, feeders=False):
    """Realtime chunk latency (gui.py geometry): block 0.256 s -> decoder T=31 frames
    (n_res = 31, no formant shift), retrieval on the last 16 HuBERT frames; v1/40k generator, 768-d index.
    p50 / p90 / p99 over `chunks` chunks after `warm` warm-ups (--stream: 300 / 30; the `stream` object of the default line: 200 / 20,
    SURVEY 8d config 5), replayed from a hipGraph of the fixed-shape chunk (--graph 0: eager).  Returns the JSON line as a dict."""
    import rvc_amd
    from oracle import nsf_oracle, synth

    dev = torch.device("cuda", 0)
    cfg = nsf_oracle.CONFIGS["v1_40k"]
    w = synth.make_dec_weights(cfg, 1234)
    T, NQ = 31, 16
    z, f0, g = synth.make_dec_inputs(cfg, 1, T)
    noise = nsf_oracle.reference_noise(1, T, cfg.upp)
    if index is None:
        idx = synth.make_ivf(a.index_n, a.index_d, seed=4321, kmeans_iters=1)
        index = rvc_amd.IVFFlatHIP.from_arrays(idx["centroids"], idx["list_offsets"], idx["ids"], idx["vecs"], device=dev)
    index.reserve(NQ)
    gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=dev, operand=a.operand, max_B=1, max_T=64)
    zd, fd, gd, nd = z.to(dev), f0.to(dev), g.to(dev), noise.to(dev)
    feats = synth.make_phone(1, NQ, a.index_d)[0].to(dev).contiguous()
    fbuf = torch.empty_like(feats)
    hold = {}

    def hot():
        fbuf.copy_(feats)
        index.search_blend(fbuf, a.index_rate, 8, skip_if_short=True)
        hold["o"] = gen(zd, fd, gd, noise=nd)

    st, cap = _time_chunks(hot, n=chunks, warm=warm, graph=bool(a.graph))
    assert torch.isfinite(hold["o"]).all()

    def kernel_breakdown(fn, handles, n=50):
        """us per chunk of every launch (HIP events around each launch, eager, AFTER the timed chunks): where a chunk's time goes"""
        for h_ in handles:
            h_.profile(True)
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        out = {}
        for h_ in handles:
            for s_ in h_.profile_read():
                out[s_["name"]] = round(1e3 * s_["ms"] / n, 2)
            h_.profile(False)
        return out

    hot_kernels = kernel_breakdown(hot, [index, gen]) if breakdown else None
    line = {"metric": "realtime chunk latency p50 (retrieval + NSF decode), v1/40k, 256 ms block", "value": st["p50_ms"],
            "unit": "ms", "p90": st["p90_ms"], "p99": st["p99_ms"], "higher_is_better": False, "n_gpus": 1, "dtype": a.operand,
            "data": "synthetic", "hot_path": dict(st, **({"kernels_us_per_chunk": hot_kernels} if breakdown else {})),
            "config": {"workload": "BASELINE configs[4] (hot path only): T=31 frames -> 12400 samples, 16 queries", "hipgraph": cap}}
    if a.operand != "fp32" and a.index_d == 768:
        # the whole RVC.infer of the realtime loop after HuBERT / f0 (infer/lib/rtrvc.py:163-251, gui.py:1057-1090):
        # 141 feature rows, retrieval on the last 16 (guarded), x2 -> 282 frames, protect mix, infer(skip_head=250,
        # return_length=31) = enc_p on 282 frames, flow on the last 56, decoder on 31, then the SOLA stitch.
        from oracle.front_oracle import FrontConfig

        fcfg = FrontConfig()
        front = rvc_amd.FrontHIP(vars(fcfg), synth.make_front_weights(fcfg, 1234), device=dev, operand=a.operand, max_B=1, max_T=288)
        NF, P_LEN, SKIP, RET = 141, 282, 250, 31
        hub = synth.make_phone(1, NF, 768)[0].to(dev).contiguous()
        hbuf = torch.empty_like(hub)
        pitchf_all = synth.make_f0(1, P_LEN).to(dev)
        pitch_all = synth.make_pitch(pitchf_all.cpu()).to(dev)
        zc, Lb, Ls = cfg.sr // 100, 4 * (cfg.sr // 100), cfg.sr // 100
        blk = RET * zc - Lb - Ls
        sola_buf = torch.zeros(Lb, device=dev)
        fade_in = torch.sin(0.5 * math.pi * torch.linspace(0.0, 1.0, Lb, device=dev)) ** 2
        fade_out = 1 - fade_in
        sid_g = gd
        net = type("N", (), {})()
        net.emb_g = lambda sid: sid_g.reshape(1, -1)
        net.dec = gen
        sid0 = torch.zeros(1, dtype=torch.long, device=dev)

        def whole():
            hbuf.copy_(hub)
            ph = rvc_amd.glue.retrieve_blend_expand(hbuf.unsqueeze(0), index, a.index_rate, pitchf_all, 0.33, P_LEN, realtime_guard=True,
                                                    skip_rows=SKIP // 2)   # rtrvc.py:167-186 (search + blend of the new rows), :221-233 (x2, protect)
            wav = rvc_amd.infer_hip(net, front, ph, None, sid0, pitch_all, pitchf_all, SKIP, RET, RET)[0, 0]  # :236-247
            hold["w"] = rvc_amd.glue.sola(wav.contiguous(), sola_buf, fade_in, fade_out, blk, Ls)  # gui.py:1057-1090

        st2, cap2 = _time_chunks(whole, n=chunks, warm=warm, graph=bool(a.graph))
        assert torch.isfinite(hold["w"]).all()
        if breakdown:
            st2 = dict(st2, kernels_us_per_chunk=kernel_breakdown(whole, [index, front, gen]))
        line["whole_chunk"] = dict(st2, what="retrieval (16 rows, guarded) + x2 + protect + enc_p(282) + flow(56) + decode(31) + SOLA", hipgraph=cap2)
        if feeders:
            # (--stream only) what the realtime loop runs AHEAD of that per block, as architecture proxies on PyTorch-ROCm (tools/e2e_proxies.py, eager): HuBERT
            # on the 2.82 s rolling window (rtrvc.py:142-162) and RMVPE on its f0 window (rtrvc.py:203-207: 4960 samples for a 4096-sample block) incl. the
            # salience decode -- with the network's GRU on torch / MIOpen and on csrc/gru.hip (what the rebound RVC.infer does, realtime.rvc_infer_hip)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from e2e_proxies import HubertProxy, RmvpeProxy
            from rvc_amd.gru import accelerate_f0_rmvpe
            from rvc_amd.realtime import f0_extractor_frame

            half = True
            hubp = HubertProxy(dev, half=half)
            win = torch.randn(1, 45120, device=dev).half()
            nf0 = f0_extractor_frame(4096, "rmvpe", 160)
            wav_f0 = torch.randn(1, nf0, device=dev)
            fd = {}
            import types

            from rvc_amd.realtime import _rmvpe_f0_graphed

            for label in ("torch_gru", "hip_gru", "hip_gru_f0_graph"):
                rm = RmvpeProxy(dev, half=half)
                if label != "torch_gru":
                    accelerate_f0_rmvpe(rm)
                holder = types.SimpleNamespace(f0_gen=types.SimpleNamespace(rmvpe=rm, is_half=half, device=dev))

                def feed():
                    with torch.no_grad():
                        hubp.extract_features(win, None, 9)
                        if label == "hip_gru_f0_graph":  # what realtime.rvc_infer_hip does: the f0 chain replayed from a hipGraph after 3 eager blocks
                            return _rmvpe_f0_graphed(holder, wav_f0[0], nf0 // 160, 0)
                        hid = rm._mel2hidden(rm.mel_extractor(wav_f0, center=True))
                    return rvc_amd.glue.rmvpe_f0(hid.squeeze(0).float(), int(hid.shape[1]), 0, 0.03)

                def hubert_only():
                    with torch.no_grad():
                        hubp.extract_features(win, None, 9)

                run_whole = whole
                if cap2:
                    gw_ = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gw_):
                        whole()
                    run_whole = gw_.replay

                def chunk():
                    feed()
                    run_whole()

                stf, _ = _time_chunks(chunk, n=min(chunks, 100), warm=10, graph=False)
                sth, _ = _time_chunks(hubert_only, n=50, warm=5, graph=False)
                fd[label] = dict(stf, hubert_proxy_p50_ms=sth["p50_ms"], rmvpe_proxy_p50_ms=round(stf["p50_ms"] - sth["p50_ms"] - st2["p50_ms"], 4))
            line["chunk_with_feeders"] = dict(fd, what="HuBERT proxy (45120 samples) + RMVPE proxy (%d samples -> 32 frames) + salience decode, eager, then the whole chunk above" % nf0)
    return line



def _ubench_ceiling():
    """What the K loop's instruction mix reaches when it has the chip to itself, from the NEWEST committed run of
    tools/ubench/kloop2 (tools/gpu_round.sh runs it every round and stores profiles/rNN_ubench_kloop2_issue_model.txt)."""
    import glob
    import re

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ubench_kloop2_issue_model.txt")))
    if not files:
        return None
    out = {"source": "profiles/" + os.path.basename(files[-1]), "kind": "parsed from the newest committed micro-benchmark run, NOT measured in this run"}
    for ln in open(files[-1]):
        m = re.search(r"\((\d+) TF/s, eff\. clock ([0-9.]+) GHz\)\s+(.*)$", ln)
        if not m or "waves/SIMD=1" not in ln or "MI=1 NJ=6" not in ln or "FILL=0" not in ln:
            continue
        if m.group(3).startswith("MFMA only"):
            out["mfma_only_tflops"], out["mfma_only_clock_ghz"] = float(m.group(1)), float(m.group(2))
        elif m.group(3).startswith("the K loop as shipped"):
            out["kloop_tflops"], out["kloop_clock_ghz"] = float(m.group(1)), float(m.group(2))
    return out if "kloop_tflops" in out else None


def _time_hot(gen, index, phone_d, zd, f0d, gd, nd, index_rate, steps, warmup, use_graph=True):
    """K timed passes of the hot path (feats copy + search_blend + generator) on resident inputs, replayed from a hipGraph.
    Returns (seconds per step, last output, the eager step function, captured?)."""
    feats = torch.empty_like(phone_d)
    hold = {}

    def step():
        feats.copy_(phone_d)
        index.search_blend(feats, index_rate, 8)
        hold["o"] = gen(zd, f0d, gd, noise=nd)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    run, cap = step, False
    if use_graph:
        try:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                step()
            torch.cuda.synchronize()
            run, cap = gr.replay, True
        except Exception as e:  # noqa
            print("[bench] graph capture failed (%s); timing eager launches" % e, file=sys.stderr)
            torch.cuda.synchronize()
    for _ in range(warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(hold["o"]).all(), "non-finite generator output"
    return dt, hold["o"], step, cap


def leg_batch64(a, dev, cfg, w, index, B=64, steps=3, warmup=1):
    """BASELINE configs[2] inside the default line: 64 x 10 s clips per step on one GPU, hipGraph-captured, 3 timed steps."""
    import rvc_amd
    from oracle import nsf_oracle, synth

    T = a.frames
    z, f0, g = synth.make_dec_inputs(cfg, B, T, seed=1234)
    noise = nsf_oracle.reference_noise(B, T, cfg.upp, 114514)
    phone = synth.make_phone(B, NQ_CLIP, a.index_d, seed=1234)
    index.reserve(B * NQ_CLIP)
    gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=dev, operand=a.operand, max_B=B, max_T=T)
    phone_d = phone.to(dev).reshape(B * NQ_CLIP, a.index_d).contiguous()
    dt, _, step, cap = _time_hot(gen, index, phone_d, z.to(dev), f0.to(dev), g.to(dev), noise.to(dev), a.index_rate, steps, warmup, bool(a.graph))
    gen.profile(True)
    step()
    torch.cuda.synchronize()
    gs = gen.profile_read()
    gen.profile(False)
    dom = max(gs, key=lambda s_: s_["ms"])
    ach = dom["flops"] / (dom["ms"] * 1e-3)
    tot_ms = sum(s_["ms"] for s_ in gs)
    out = {"what": "BASELINE configs[2]: %d x 10 s clips per step, v2/48k, T=%d, %d queries, same index; hipGraph-captured; %d warm-up + %d timed steps"
                   % (B, T, B * NQ_CLIP, warmup, steps),
           "ms_per_step": 1e3 * dt, "ms_per_clip": 1e3 * dt / B, "value": B * CLIP_SECONDS * (T / T_CLIP) / dt, "unit": "x real-time", "steps": steps,
           "hipgraph": cap, "workspace_GB": gen.workspace_bytes / 1e9,
           "roofline": {"bound": "mfma", "kernel": dom["name"], "achieved": ach / 1e12, "peak": PEAK[a.operand] / 1e12, "unit": "TFLOP/s",
                        "frac": ach / PEAK[a.operand], "avg_launch_us": 1e3 * dom["ms"] / dom["launches"],
                        "generator_all_kernels_frac": GEN_FLOP_PER_CLIP * B * (T / T_CLIP) / (tot_ms * 1e-3) / PEAK[a.operand]},
           "kernels_ms_per_step": {s_["name"]: round(s_["ms"], 3) for s_ in gs if s_["ms"] >= 0.5}}
    del gen
    torch.cuda.empty_cache()
    return out


def leg_operands(a, dev, cfg, w, index, phone_d, zd, f0d, gd, nd, out_default, oracle_out=None):
    """BASELINE configs[1] AS WRITTEN (bf16 operands) timed beside the default operand type, and what each costs in parity:
    RMS distance of the waveform to the exact-fp32 HIP kernels on the same inputs (operand='fp32': <= 2e-5 of the reference,
    tests/test_gpu_generator.py) and -- when the cpu_baseline leg ran the oracle restatement on this clip with the same
    injected noise -- to the oracle's waveform itself."""
    import warnings

    import rvc_amd

    T = a.frames
    res = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g32 = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=dev, operand="fp32", max_B=1, max_T=T)
        ref = g32(zd[:1], f0d[:1], gd[:1], noise=nd[:1]).float()
        del g32
        gb = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=dev, operand="bf16", max_B=1, max_T=T)
        dt, ob, _, cap = _time_hot(gb, index, phone_d[:NQ_CLIP], zd[:1], f0d[:1], gd[:1], nd[:1], a.index_rate, a.steps, a.warmup, bool(a.graph))
        del gb
    rms = lambda x_, y_: float((x_.float() - y_.float()).pow(2).mean().sqrt())
    res["what"] = ("BASELINE configs[1] with bf16 MFMA operands (as BASELINE.json writes it), same step / inputs / graph as the headline; parity = RMS of the "
                   "waveform against the exact-fp32 HIP kernels%s; north-star bar 1e-3" % ("" if oracle_out is None else " and against the CPU oracle"))
    res["bf16"] = {"ms_per_step": 1e3 * dt, "value": CLIP_SECONDS * (T / T_CLIP) / dt, "unit": "x real-time", "hipgraph": cap,
                   "rms_vs_fp32_kernels": rms(ob[:1], ref), "meets_1e-3": rms(ob[:1], ref) <= 1e-3}
    res[a.operand] = {"rms_vs_fp32_kernels": rms(out_default[:1], ref), "meets_1e-3": rms(out_default[:1], ref) <= 1e-3}
    res["output_rms"] = float(ref.pow(2).mean().sqrt())
    if oracle_out is not None:
        oc = oracle_out.to(dev).reshape(ref.shape)
        res["bf16"]["rms_vs_cpu_oracle"] = rms(ob[:1], oc)
        res[a.operand]["rms_vs_cpu_oracle"] = rms(out_default[:1], oc)
        res["fp32_kernels_rms_vs_cpu_oracle"] = rms(ref, oc)
    torch.cuda.empty_cache()
    return res


def _cpu_quota():
    """CPUs the cgroup lets this process use (cgroup v2 cpu.max / v1 cfs quota), None when unlimited or unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, int(q) // int(per))
    except Exception:  # noqa
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, q // per)
    except Exception:  # noqa
        return None


def e2e_mode(a):
    """BASELINE.md section 3's end-to-end picture (`--e2e`): what a user's `VC.vc_multi` / `vc_single` costs per 10 s clip once the hot
    path is the HIP one -- `Pipeline.convert_files` (bound by `rvc_amd.install()` on the RVC-shaped skeleton of tests/skeleton) with the
    WebUI defaults (rmvpe, index file + index_rate 0.75, rms_mix_rate 0.25, protect 0.33), fed by ARCHITECTURE PROXIES of the two networks the
    north star leaves on PyTorch-ROCm (tools/e2e_proxies.py: HuBERT-base via transformers, the RMVPE U-Net + GRU restated; random weights),
    for 1 file and for 64 files.  Two passes per case: (1) the plain call, wall clock, no extra synchronisation = the figure a user sees;
    (2) the same call with every stage bracketed by `torch.cuda.synchronize()` + `perf_counter` = the split (its total is a little larger:
    the brackets remove the overlap of host work with queued GPU work).  Stages: host high-pass `filtfilt` (pipeline.py:216) and cut-point
    search (:219-232), reading the index file (the reference re-reads it per call too, :205-218), RMVPE proxy incl. mel + salience decode, HuBERT proxy, retrieval + blend, `net_g.infer` (front + generator), finish
    (`change_rms`, scaling, D2H copy), other (padding, slicing, Python)."""
    import tempfile
    import types

    import rvc_amd
    import rvc_amd.pipeline as rp
    from oracle import ivf_oracle, nsf_oracle, synth
    from oracle.front_oracle import FrontConfig

    sys.path.insert(0, os.path.join(ROOT, "tests", "skeleton"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from e2e_proxies import HubertProxy, RmvpeProxy

    dev = torch.device("cuda", 0)
    half = a.operand != "fp32"
    # torch sizes its CPU thread pool by the HOST's cores (128 on the GPU box) while the container's CFS quota is 16 CPUs: once the GRU is off the
    # critical path the conversion is host-bound, the OpenMP workers of any small CPU op spin after it, the cgroup's 100 ms quota is gone in ~20 ms and
    # the whole process is throttled for the rest of the period (measured: 70-85 ms stalls at random places, even inside scipy's filtfilt; 98 ms per
    # clip instead of 17).  A deployment inside a CPU quota sets OMP_NUM_THREADS to it; so does this mode.
    quota, host_threads = _cpu_quota(), torch.get_num_threads()
    if quota and host_threads > max(1, quota // 2):
        torch.set_num_threads(max(1, quota // 2))
    rvc_amd.install(device=dev, operand=a.operand)
    import infer.modules.vc.pipeline as pl
    import rvc.synthesizer as rs

    cfg, fcfg = nsf_oracle.CONFIGS["v2_48k"], FrontConfig()
    weight = dict(synth.make_front_weights(fcfg, 1234))
    weight.update({"dec." + k: v for k, v in synth.make_dec_weights(cfg, 1234).items()})
    cpt = dict(weight=weight, f0=1, version="v2", info="synthetic", sr="48k",
               config=[1025, 32, 192, 192, 768, 2, 6, 3, 0, "1", cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes, cfg.upsample_rates,
                       cfg.upsample_initial_channel, cfg.upsample_kernel_sizes, 109, cfg.gin_channels, cfg.sr])
    net_g, _ = rs.get_synthesizer(cpt, dev)
    if half:
        net_g = net_g.half()  # infer/modules/vc/modules.py:94-95
    # configs/config.py: x_pad 1 / x_query 6 / x_center 38 / x_max 41 (the geometry BASELINE's T = 1198 frames per 10 s clip assumes)
    config = types.SimpleNamespace(device=dev, is_half=half, x_pad=1, x_query=6, x_center=38, x_max=41)
    pipe = pl.Pipeline(cfg.sr, config)
    pipe.f0_gen = types.SimpleNamespace(rmvpe=RmvpeProxy(dev, half=half), is_half=half, device=dev)
    hub = HubertProxy(dev, half=half)
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "added.index")
    ivf_oracle.write_index(synth.make_ivf(a.index_n, 768, seed=4321, kmeans_iters=1), path)
    tail = (0, "rmvpe", path, a.index_rate, 1, 3, cfg.sr, 0, 0.25, "v2", 0.33)

    acc = {}
    orig = {}

    sub = {}  # nested timers (parts of a stage): kept apart from `acc`, whose entries are disjoint

    def timed(name, fn, store=None):
        store = acc if store is None else store

        def w(*args, **kw):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn(*args, **kw)
            torch.cuda.synchronize()
            store[name] = store.get(name, 0.0) + time.perf_counter() - t0
            return r
        return w

    def instrument(on):
        names = (("index_read_file_h2d", "_open_index"), ("cut_points_host", "_cut_points"), ("rmvpe_proxy_mel_net_decode", "_rmvpe_on_device"), ("hubert_proxy", "hubert_device"),
                 ("retrieval_blend", "blend_segments"), ("infer_front_generator", "infer_segments"), ("finish_rms_scale_d2h", "_finish_file"))
        if on:
            for label, attr in names:
                orig[attr] = getattr(rp, attr)
                setattr(rp, attr, timed(label, orig[attr]))
            orig["signal"] = pl.signal
            pl.signal = types.SimpleNamespace(filtfilt=timed("highpass_filtfilt_host", orig["signal"].filtfilt), butter=orig["signal"].butter)
            rm = pipe.f0_gen.rmvpe  # inside the RMVPE stage: the mel front end and the bidirectional GRU (sequential over ~1200 frames)
            orig["mel"], orig["gru"] = rm.mel_extractor, rm.model.gru.forward
            rm.mel_extractor = timed("mel_stft", orig["mel"], sub)
            rm.model.gru.forward = timed("bigru", orig["gru"], sub)
        else:
            for label, attr in names:
                setattr(rp, attr, orig[attr])
            pl.signal = orig["signal"]
            pipe.f0_gen.rmvpe.mel_extractor, pipe.f0_gen.rmvpe.model.gru.forward = orig["mel"], orig["gru"]

    def convert(audios):
        return pipe.convert_files(hub, net_g, 0, [x.copy() for x in audios], [0, 0, 0], *tail)

    def run_cases(gru_hip):
        # (the RMVPE proxy's bidirectional GRU on torch / MIOpen -- the north star's "RMVPE on PyTorch-ROCm" taken literally -- or on csrc/gru.hip,
        #  what rvc_amd.install() does by default: pipeline._rmvpe_on_device -> accelerate_rmvpe; RVCMI_RMVPE_GRU=0 opts out)
        os.environ["RVCMI_RMVPE_GRU"] = "1" if gru_hip else "0"
        pipe.f0_gen = types.SimpleNamespace(rmvpe=RmvpeProxy(dev, half=half), is_half=half, device=dev)
        cases = {}
        for n, warm, reps in ((1, 3, 7), (a.e2e_files, 1, 3)):
            audios = [synth.make_audio16k(160000, 1234 + i) for i in range(n)]
            for _ in range(warm):
                out = convert(audios)
            torch.cuda.synchronize()
            walls = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = convert(audios)
                torch.cuda.synchronize()
                walls.append(time.perf_counter() - t0)
            assert len(out) == n and all(np.isfinite(o).all() and o.shape[0] == 479040 for o in out), [o.shape for o in out]
            walls.sort()
            wall = walls[len(walls) // 2]
            instrument(True)
            acc.clear()
            sub.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            convert(audios)
            torch.cuda.synchronize()
            tot = time.perf_counter() - t0
            instrument(False)
            split = {k: round(1e3 * v / n, 3) for k, v in acc.items()}
            split["other_host_pad_slice_python"] = round(1e3 * (tot - sum(acc.values())) / n, 3)
            feeders = split["hubert_proxy"] + split["rmvpe_proxy_mel_net_decode"]
            host = split["highpass_filtfilt_host"] + split["cut_points_host"] + split["other_host_pad_slice_python"] + split["index_read_file_h2d"]
            hot = split["retrieval_blend"] + split["infer_front_generator"]
            cases["files_%d" % n] = {"files": n, "wall_ms_per_clip": round(1e3 * wall / n, 3), "rtf": CLIP_SECONDS * n / wall, "runs": reps,
                                     "wall_ms_per_clip_min_max": [round(1e3 * walls[0] / n, 3), round(1e3 * walls[-1] / n, 3)],
                                     "instrumented_total_ms_per_clip": round(1e3 * tot / n, 3), "split_ms_per_clip": split,
                                     "groups_ms_per_clip": {"feeders_pytorch_rocm_proxies": round(feeders, 3), "host": round(host, 3),
                                                            "hip_hot_path_retrieval_infer": round(hot, 3), "finish": split["finish_rms_scale_d2h"]},
                                     "rmvpe_proxy_parts_ms_per_clip": {"mel_stft": round(1e3 * sub.get("mel_stft", 0.0) / n, 3),
                                                                       "bigru_384_256_bidirectional": round(1e3 * sub.get("bigru", 0.0) / n, 3),
                                                                       "unet_head_linear_decode": round(split["rmvpe_proxy_mel_net_decode"] - 1e3 * (sub.get("mel_stft", 0.0) + sub.get("bigru", 0.0)) / n, 3)},
                                     "rmvpe_gru": "csrc/gru.hip (%d module swapped)" % getattr(pipe.f0_gen.rmvpe, "_rvcmi_gru", 0) if gru_hip else "torch nn.GRU (MIOpen)",
                                     "long_pole": max(split, key=split.get)}
        return cases

    env0 = os.environ.get("RVCMI_RMVPE_GRU")
    cases_torch = run_cases(False)
    cases = run_cases(True)
    if env0 is None:
        os.environ.pop("RVCMI_RMVPE_GRU", None)
    else:
        os.environ["RVCMI_RMVPE_GRU"] = env0
    rvc_amd.uninstall()
    import shutil

    shutil.rmtree(tmp, ignore_errors=True)  # (the synthetic index file)
    big = cases["files_%d" % a.e2e_files]
    return {"metric": "end-to-end real-time factor per GPU, Pipeline.convert_files on 10 s clips (feeders = architecture proxies on PyTorch-ROCm)",
            "value": big["rtf"], "unit": "x real-time (audio-sec/wall-sec), %d files per call" % a.e2e_files, "n_gpus": 1, "higher_is_better": True,
            "dtype": "%s (is_half=%s: HuBERT / RMVPE proxies and net_g in half like infer/modules/vc/modules.py:94-95)" % (a.operand, half),
            "data": "synthetic (seeded audio, weights, index; random-weight architecture proxies for HuBERT-base and RMVPE)",
            "config": {"workload": "Pipeline.convert_files via rvc_amd.install() on tests/skeleton: WebUI defaults (rmvpe, index file %dx768 index_rate %.2f, "
                                   "rms_mix_rate 0.25, protect 0.33, x_pad 1), 10 s / 16 kHz inputs -> 48 kHz" % (a.index_n, a.index_rate)},
            "cases": cases,
            "cases_with_torch_gru": cases_torch,
            "host": {"cpu_quota": quota, "torch_threads_default": host_threads, "torch_threads_used": torch.get_num_threads()},
            "reading": "hot path (retrieval + net_g.infer) vs everything around it: see groups_ms_per_clip; the split pass synchronises around every "
                       "stage, the wall figures do not.  `cases` / `value` = what rvc_amd.install() runs by default (RMVPE's GRU on csrc/gru.hip, the rest of "
                       "RMVPE and HuBERT on PyTorch-ROCm); `cases_with_torch_gru` = the north star's 'RMVPE on PyTorch-ROCm' taken literally (RVCMI_RMVPE_GRU=0)"}


_REAL_STDOUT = None


def emit(line: dict) -> None:
    """THE one JSON line, on the process's real stdout."""
    out = _REAL_STDOUT if _REAL_STDOUT is not None else sys.stdout
    out.write(json.dumps(line) + "\n")
    out.flush()


def _list_major_bytes(sizes, asg, d, ms_per_search):
    """What the list-major launches move for THIS assignment of queries to lists (host arithmetic, outside every timed region): a work
    item = (list, 32 sorted queries, 32 rows) stages its 32 query rows and its 32 list rows, so a list's rows are read once per
    32-query tile and a query once per 32-row tile of its list; plus the fp32 score scratch (written by the tiles, read by the
    selector), ~12 verified candidate rows and the 8 gathered rows of the blend per query."""
    import numpy as np

    cnt = np.bincount(asg, minlength=len(sizes)).astype(np.int64)
    qt, rt = (cnt + 31) // 32, (sizes + 31) // 32
    tile = float((qt * sizes + rt * cnt).sum()) * 4.0 * d
    scratch = 2.0 * float((cnt * sizes).sum()) * 4.0
    sel = float(len(asg)) * (12 + 8 + 2) * 4.0 * d
    tot = tile + scratch + sel
    return {"list_major_bytes": {"tiles": tile, "score_scratch": scratch, "select_and_blend": sel, "total": tot},
            "list_major_GBps": tot / (ms_per_search * 1e-3) / 1e9, "work_items": int((qt * rt).sum())}


def main():
    global _REAL_STDOUT
    a = parse()
    if a.torch_gpu_baseline_worker:
        return torch_gpu_baseline_worker(a)
    if not a.stream and not a.e2e:
        ensure_world(a)  # may replace this process with the N-rank launcher (before fd 1 is touched)
    if _REAL_STDOUT is None:
        # libraries write banners to fd 1 (RCCL prints its version block at the first communicator): keep stdout for the JSON line
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    if a.stream:
        return emit(stream_mode(a, feeders=True))
    if a.e2e:
        return emit(e2e_mode(a))
    if a.dist_selftest:
        return dist_selftest(a)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP hot path has no CPU fallback)")
    # torch sizes its CPU pool by the host's cores, not by the container's CFS quota (128 threads on a 16-CPU quota here): N ranks of spinning OpenMP
    # workers exhaust the quota and the kernel throttles the whole cgroup for the rest of each 100 ms period (see e2e_mode).  Cap the pool to this
    # rank's share; the cpu_baseline leg sets its own thread counts.
    quota = _cpu_quota()
    if quota and torch.get_num_threads() > max(1, quota // max(1, world)):
        torch.set_num_threads(max(1, quota // max(1, world)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    use_dist = world > 1 or os.environ.get("RVCMI_BENCH_FORCE_DIST") == "1"  # (the env switch exercises the RCCL path on one GPU)
    if use_dist:
        dist.init_process_group(backend="nccl", device_id=dev)

    import rvc_amd
    from oracle import nsf_oracle, synth

    cfg = nsf_oracle.CONFIGS["v2_48k"]
    B, T = a.batch, a.frames
    w = synth.make_dec_weights(cfg, 1234)
    z, f0, g = synth.make_dec_inputs(cfg, B, T, seed=1234 + rank)
    noise = nsf_oracle.reference_noise(B, T, cfg.upp, 114514 + rank)
    phone = synth.make_phone(B, NQ_CLIP, a.index_d, seed=1234 + 100 * rank)

    # ---- index: built on rank 0, ONE RCCL broadcast of the packed blob ----------
    idx = None
    t_bcast, blob_bytes, ranks_equal = 0.0, None, None
    big_index = a.index_n > 200000  # configs[3]: 1M x 256 -- trained on the GPU (rvcmi_ivf_build), numpy k-means would take minutes
    if rank == 0:
        if big_index:
            rows = synth.make_clustered_rows(a.index_n, a.index_d, synth.ivf_nlist(a.index_n), seed=4321)
            index = rvc_amd.IVFFlatHIP.train(rows, nlist=synth.ivf_nlist(a.index_n), niter=2, seed=4321, device=dev)
            del rows
        else:
            idx = synth.make_ivf(a.index_n, a.index_d, seed=4321, kmeans_iters=1)
            index = rvc_amd.IVFFlatHIP.from_arrays(idx["centroids"], idx["list_offsets"], idx["ids"], idx["vecs"], device=dev)
    else:
        index = None
    if use_dist:
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        index = rvc_amd.dist.broadcast_index(index, src=0, device=dev)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
        tb = torch.tensor([t_bcast], device=dev, dtype=torch.float64)
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        t_bcast = float(tb.item())
    blob_bytes = int(index.blob().numel())
    index.reserve(B * NQ_CLIP)
    if use_dist:
        # every rank searches the SAME seeded queries on ITS copy of the index; (distances, ids) must be bit-identical to rank 0's
        qv = synth.make_phone(1, 64, a.index_d, seed=999)[0].to(dev).contiguous()
        Dv, Iv = index.search(qv, 8)
        ranks_equal = rvc_amd.dist.ranks_agree(Dv.contiguous()) and rvc_amd.dist.ranks_agree(Iv.contiguous())
        if not ranks_equal:
            raise SystemExit("bench.py: rank %d's first search differs from rank 0's after the index broadcast" % rank)

    gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=dev, operand=a.operand, max_B=B, max_T=T)
    zd, f0d, gd, nd = z.to(dev), f0.to(dev), g.to(dev), noise.to(dev)
    phone_d = phone.to(dev).reshape(B * NQ_CLIP, a.index_d).contiguous()
    feats = torch.empty_like(phone_d)
    out_holder = {}

    front = None
    if a.whole:
        from oracle.front_oracle import FrontConfig

        fcfg0 = FrontConfig()
        front = rvc_amd.FrontHIP(vars(fcfg0), synth.make_front_weights(fcfg0, 1234), device=dev, operand=a.operand, max_B=B, max_T=T)
        pitch0 = synth.make_pitch(f0).to(dev)
        nz0 = torch.randn(B, fcfg0.inter_channels, T, device=dev)

    def run_whole(front_, pitch_, nz_):
        # Pipeline.vc (pipeline.py:118-175) from the HuBERT features on: search + blend + x2 frames + protect mix in ONE pass
        # (rvcmi_ivf_search_blend_expand, WebUI default protect 0.33), then enc_p -> z_p -> flow^-1 -> generator
        Tw = 2 * NQ_CLIP
        ph = rvc_amd.glue.retrieve_blend_expand(phone_d.view(1, B * NQ_CLIP, a.index_d), index, a.index_rate,
                                                pitchf=f0d[:, :Tw].reshape(1, -1), protect=0.33, p_len=B * Tw).view(B, Tw, a.index_d)
        zz = front_(ph, pitch_[:, :Tw], None, gd, 0, noise=nz_[:, :, :Tw])
        return gen(zz, f0d[:, :Tw], gd, noise=nd[:, :Tw * cfg.upp])

    def step():
        if front is None:
            feats.copy_(phone_d)
            index.search_blend(feats, a.index_rate, 8)
            out_holder["o"] = gen(zd, f0d, gd, noise=nd)
        else:
            out_holder["o"] = run_whole(front, pitch0, nz0)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    graph = None
    if a.graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
            torch.cuda.synchronize()
        except Exception as e:  # noqa
            print("[bench] graph capture failed (%s); timing eager launches" % e, file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    run = graph.replay if graph is not None else step

    for _ in range(a.warmup):
        run()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # (timing ablations -- RVCMI_DBG / RVCMI_POST_DBG, wrong results by construction -- set RVCMI_BENCH_ABLATION=1; such a line is never a result)
    assert os.environ.get("RVCMI_BENCH_ABLATION") == "1" or torch.isfinite(out_holder["o"]).all(), "non-finite generator output"

    # ---- stability report: R more repetitions of the same K-step loop (same step, same graph), per-rank wall clock ----
    rep_stats = None
    if a.repeats > 0:
        rt = []
        for _ in range(a.repeats):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                run()
            torch.cuda.synchronize()
            rt.append(1e3 * (time.perf_counter() - t1) / a.steps)
        rts = sorted(rt)
        rep_stats = {"n": a.repeats, "steps_each": a.steps, "ms_per_step_median": rts[len(rts) // 2], "ms_per_step_min": rts[0],
                "ms_per_step_max": rts[-1], "ms_per_step_all": [round(x, 4) for x in rt]}
        if use_dist:  # per-rank medians, so that the line shows every rank ran (min / max over ranks)
            tm = torch.tensor([rep_stats["ms_per_step_median"]], device=dev, dtype=torch.float64)
            lo, hi = tm.clone(), tm.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            rep_stats["rank_median_ms_min_max"] = [float(lo.item()), float(hi.item())]

    # ---- roofline leg: the same step, eager, every kernel bracketed by HIP events --------------
    roof = None
    if rank == 0 and not a.no_roofline:
        gen.profile(True)
        index.profile(True)
        if front is not None:
            front.profile(True)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        gs, ivs = gen.profile_read(), index.profile_read()
        gen.profile(False)
        index.profile(False)
        if front is not None:
            ivs = ivs + front.profile_read()
            front.profile(False)
        dom = max(gs, key=lambda s: s["ms"])
        peak = PEAK[a.operand]
        ach = dom["flops"] / (dom["ms"] * 1e-3)
        tot_ms = sum(s["ms"] for s in gs) / 3.0
        gen_ach = GEN_FLOP_PER_CLIP * B * (T / T_CLIP) / (tot_ms * 1e-3)
        # HBM-side bytes per launch of the dominant kernel: NOT measured in this run (PMC counters need their own rocprofv3
        # passes); taken from the newest committed pass of the same command, and labelled as such.
        traffic, traffic_source, traffic_all = None, None, None
        import glob

        for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):  # newest round first
            tag = os.path.basename(fn).split("_")[0]
            try:
                pmj = json.load(open(fn))
                pm = pmj["kernels"].get(dom["name"])
                if pm and B == 1 and T == T_CLIP and a.operand == "fp16":
                    traffic = pm["bytes_per_launch"]
                    traffic_source = "committed rocprofv3 --pmc pass profiles/%s_pmc_traffic.json (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), not measured in this run" % tag
                    traffic_all = {k: v["bytes_per_launch"] for k, v in pmj["kernels"].items()}
                    break
            except Exception:  # noqa
                pass
        roof = {"bound": "mfma", "kernel": dom["name"], "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_source, "traffic_all_kernels": traffic_all,
                "hbm_side": None if traffic is None else {"achieved": traffic / (dom["ms"] / dom["launches"] * 1e-3) / 1e9, "peak": 8000.0,
                                                         "unit": "GB/s", "frac": traffic / (dom["ms"] / dom["launches"] * 1e-3) / 8e12},
                "avg_launch_us": 1e3 * dom["ms"] / dom["launches"], "launches_per_step": dom["launches"] // 3,
                "alg_bytes_per_launch": dom["bytes"] / dom["launches"],
                "generator_all_kernels": {"ms_per_step": tot_ms, "achieved": gen_ach / 1e12, "frac": gen_ach / peak},
                "kernels_ms_per_step": {s["name"]: round(s["ms"] / 3.0, 4) for s in gs + ivs}}
        # Context, not a different yardstick: what the SAME K loop (MFMA + LDS B fragments + L2 weight fragments, nothing else)
        # reaches when it has the chip to itself (tools/ubench/kloop2.hip; the power management holds such a loop well below the
        # 2.4 GHz the 2.5 PF figure assumes).  Parsed from the newest committed run of the micro-benchmark.
        ub = _ubench_ceiling() if a.operand in ("fp16", "bf16") else None
        if ub is not None:
            roof["ubench_ceiling"] = dict(ub, frac_of_kloop_ceiling=ach / (ub["kloop_tflops"] * 1e12))
        def scan_total(stats):
            """The scan as one unit: the query-major kernel (`ivf_scan`), or the three launches of the list-major path
            (`ivf_plan` + `ivf_scan` = score tiles + `ivf_select`; round 4) -- time summed, bytes = the SURVEY 8d model of `ivf_scan`."""
            parts = [s_ for s_ in stats if s_["name"] in ("ivf_plan", "ivf_scan", "ivf_select")]
            main = [s_ for s_ in parts if s_["name"] == "ivf_scan"]
            if not main:
                return []
            return [dict(main[0], ms=sum(s_["ms"] for s_ in parts), parts={s_["name"]: round(1e3 * s_["ms"] / max(1, s_["launches"]), 2) for s_ in parts})]

        scan = scan_total(ivs)
        if scan and scan[0]["ms"] > 0:
            roof["ivf_scan_hbm"] = {"achieved": scan[0]["bytes"] / (scan[0]["ms"] * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                    "frac": scan[0]["bytes"] / (scan[0]["ms"] * 1e-3) / 8e12,
                                    "bytes_model": "SURVEY 8d ALGORITHMIC bytes: ntotal / nlist rows per query (lists of equal size) -- a "
                                                   "model-equivalent throughput that compares the two scan designs, NOT the traffic the "
                                                   "list-major launches perform (see list_major_bytes / the PMC traffic)",
                                    "path": "list-major (plan + score tiles + select)" if "ivf_plan" in scan[0]["parts"] else "query-major",
                                    "launches_us": scan[0]["parts"]}
            if idx is not None:
                # what the scan really walks on THIS index: i.i.d. Gaussian rows give very unequal lists and the (equally
                # random) queries land in the big ones.  Host arithmetic on the inputs, outside every timed region.
                import numpy as np

                sizes = np.diff(np.asarray(idx["list_offsets"]))
                asg = synth.assign_nearest(phone.reshape(-1, a.index_d).numpy(), np.asarray(idx["centroids"]))
                rows = int(sizes[asg].sum())
                ms1 = scan[0]["ms"] / scan[0]["launches"]
                sb = rows * (4.0 * a.index_d + 8)
                roof["ivf_scan_hbm"].update({
                    "rows_per_query_model": a.index_n / max(1, len(sizes)), "rows_per_query_scanned": rows / max(1, len(asg)),
                    "largest_list": int(sizes.max()), "scanned_bytes": sb, "scanned_GBps": sb / (ms1 * 1e-3) / 1e9,
                    "unique_list_bytes": float(sizes[np.unique(asg)].sum()) * (4.0 * a.index_d + 8),
                    **_list_major_bytes(sizes, asg, a.index_d, ms1),
                    "note": "scanned_bytes = rows of the probed list x queries (what a query-major scan walks, mostly through L2; "
                            "the list-major path reads a list's rows once per 32-query tile); unique_list_bytes = the rows probed at all"})
            if idx is not None and B == 1 and a.index_d == 768:
                # side leg, not part of `value`: the same search on an index whose rows have cluster structure (lists of
                # comparable size, the bytes model above holds), queries drawn from the same mixture
                try:
                    import numpy as np

                    xr, cen = synth.make_clustered_rows(a.index_n + NQ_CLIP, a.index_d, synth.ivf_nlist(a.index_n), return_centres=True)
                    idc = synth.make_ivf_from_rows(xr[:a.index_n], kmeans_iters=2, init=cen)  # = a converged k-means on this data
                    ixc = rvc_amd.IVFFlatHIP.from_arrays(idc["centroids"], idc["list_offsets"], idc["ids"], idc["vecs"], device=dev).reserve(NQ_CLIP)
                    qc = torch.from_numpy(xr[a.index_n:]).to(dev).contiguous()
                    fb = torch.empty_like(qc)
                    for _ in range(3):
                        fb.copy_(qc)
                        ixc.search_blend(fb, a.index_rate, 8)
                    ixc.profile(True)
                    for _ in range(10):
                        fb.copy_(qc)
                        ixc.search_blend(fb, a.index_rate, 8)
                    torch.cuda.synchronize()
                    st_c = {s_["name"]: s_ for s_ in ixc.profile_read()}
                    ixc.profile(False)
                    szc = np.diff(np.asarray(idc["list_offsets"]))
                    asc = synth.assign_nearest(xr[a.index_n:], np.asarray(idc["centroids"]))
                    rows_c = int(szc[asc].sum())
                    sc = scan_total(list(st_c.values()))[0]
                    msc = sc["ms"] / sc["launches"]
                    sbc = rows_c * (4.0 * a.index_d + 8)
                    roof["ivf_clustered_index"] = {
                        "what": "same search + blend on a 10000 x 768 index of clustered rows (synth.make_clustered_rows), 599 queries of the same mixture",
                        "scan_us": 1e3 * msc, "scan_launches_us": sc["parts"], "coarse_us": 1e3 * st_c["ivf_coarse"]["ms"] / st_c["ivf_coarse"]["launches"],
                        "rows_per_query_scanned": rows_c / len(asc), "largest_list": int(szc.max()), "scanned_bytes": sbc,
                        "scanned_GBps": sbc / (msc * 1e-3) / 1e9, "frac_of_8TBps": sbc / (msc * 1e-3) / 8e12}
                except Exception as e:  # noqa
                    roof["ivf_clustered_index"] = {"error": str(e)[:200]}

    # ---- whole net_g.infer leg: retrieval -> x2 frames -> enc_p -> z_p -> flow^-1 -> decode, one graph ----
    whole = None
    if rank == 0 and not a.no_roofline and not a.whole and a.index_d == 768 and a.operand != "fp32":
        from oracle.front_oracle import FrontConfig

        fcfg = FrontConfig()
        wf = synth.make_front_weights(fcfg, 1234)
        front = rvc_amd.FrontHIP(vars(fcfg), wf, device=dev, operand=a.operand, max_B=B, max_T=T)
        pitch_d = synth.make_pitch(f0).to(dev)
        nz_zp = torch.randn(B, fcfg.inter_channels, T, device=dev)
        reps = 2

        def step_whole():
            out_holder["w"] = run_whole(front, pitch_d, nz_zp)

        try:
            for _ in range(2):
                step_whole()
            torch.cuda.synchronize()
            runw = step_whole
            if a.graph:
                gw = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gw):
                    step_whole()
                runw = gw.replay
            for _ in range(a.warmup):
                runw()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                runw()
            torch.cuda.synchronize()
            dtw = (time.perf_counter() - t0) / a.steps
            assert torch.isfinite(out_holder["w"]).all()
            front.profile(True)
            for _ in range(3):
                step_whole()
            torch.cuda.synchronize()
            fs = front.profile_read()
            front.profile(False)
            Tw = reps * NQ_CLIP
            whole = {"what": "retrieval + blend + x2 frames + protect mix (one pass) + enc_p + z_p + flow^-1 + decode = Pipeline.vc after HuBERT, one hipGraph",
                     "ms_per_step": 1e3 * dtw, "value": B * CLIP_SECONDS * (Tw / T_CLIP) / dtw, "unit": "x real-time",
                     "front_kernels_ms_per_step": {s_["name"]: round(s_["ms"] / 3.0, 4) for s_ in fs},
                     "front_ms_per_step": round(sum(s_["ms"] for s_ in fs) / 3.0, 4)}
        except Exception as e:  # noqa
            print("[bench] whole-infer leg failed: %s" % e, file=sys.stderr)

    cpu = None
    oracle_out = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and idx is not None:
        cpu = cpu_baseline(cfg, w, idx, phone, z, f0, g, noise, a.index_rate)
        oracle_out = cpu.pop("_oracle_out", None)

    # ---- the other BASELINE configurations under the same clock (N = 1, default line only): configs[2] = batch64, configs[4] = stream,
    #      configs[1] as written = bf16 operands + what each operand type costs in parity ----
    extra = {}
    if rank == 0 and world == 1 and not a.no_extra and not a.whole and B == 1 and a.operand == "fp16" and T == T_CLIP and a.index_d == 768:
        for name, fn in (("bf16", lambda: leg_operands(a, dev, cfg, w, index, phone_d, zd, f0d, gd, nd, out_holder["o"], oracle_out)),
                         ("batch64", lambda: leg_batch64(a, dev, cfg, w, index)),
                         ("stream", lambda: (lambda l_: {"what": l_["config"]["workload"] + "; v1/40k generator (configs/v1/40k.json + v2 encoder), 20 warm-ups + 200 chunks, hipGraph replay",
                                                         "hot_path": l_["hot_path"], "whole_chunk": l_.get("whole_chunk"), "unit": "ms"})(
                             stream_mode(a, chunks=200, warm=20, index=index, breakdown=False)))):
            try:
                t_leg = time.perf_counter()
                extra[name] = fn()
                extra[name]["leg_wall_s"] = round(time.perf_counter() - t_leg, 2)
            except Exception as e:  # noqa
                extra[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    tgpu = None
    if rank == 0 and world == 1 and not a.no_gpu_torch_baseline and not a.no_cpu_baseline:
        tgpu = torch_gpu_baseline(a)

    if rank == 0:
        clips = B * world * a.steps
        value = clips * CLIP_SECONDS * (T / T_CLIP) / dt
        line = {
            "metric": "real-time factor (audio-sec/wall-sec), 48 kHz v2, 10 s clips, retrieval + NSF-HiFi-GAN decode" +
                      (" + enc_p + flow (whole net_g.infer)" if a.whole else ""),
            "value": value, "unit": "x real-time (audio-sec/wall-sec), whole job",
            "per_gpu": value / world,
            "n_gpus": dist.get_world_size() if use_dist else 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # what the arithmetic is, not a precision claim: MFMA operands AND (round 4, options Y_F16 / X0_F16, default on) the
            # inter-stage streams in HBM -- the three ResBlock outputs of stages 1-3 and the ups output X0 of stages 1-3 (round 5: also
            # of the streaming stage) -- are fp16;
            # accumulators, biases, the residual stream INSIDE a ResBlock, the excitation and conv_post are fp32
            "dtype": ("%s MFMA operands%s, fp32 accumulate / in-block residual; fp32 MFMA prefilter + fp64 verified IVF distances" % (
                a.operand, "" if os.environ.get("RVCMI_Y_F16", "1") == "0" else (
                    " + fp16 inter-stage streams (ResBlock outputs%s)" % ("" if os.environ.get("RVCMI_X0_F16", "1") == "0" else (
                        ", ups output of stages 2-3" if os.environ.get("RVCMI_X0_F16_NOSTREAM") else ", ups output of stages 1-3"))))
                      ) if a.operand != "fp32" else "fp32 (fp32 MFMA, fp32 streams)",
            "data": "synthetic (seeded weights, features, f0, noise, index; no checkpoints offline)",
            "config": {"workload": "BASELINE configs[%d]: v2/48k, %d x 10 s clip(s) per GPU, T=%d frames, 599 queries/clip, "
                                   "IVF %dx%d nlist=%d nprobe=1 k=8 index_rate=%.2f" % (
                                       (a.config if a.config else (1 if B == 1 else 2)), B, T, a.index_n, a.index_d, index.nlist, a.index_rate),
                       "clips_per_gpu_per_step": B, "hipgraph": graph is not None,
                       "index_broadcast_s": t_bcast if use_dist else None, "index_blob_bytes": blob_bytes,
                       "index_broadcast_GBps": (blob_bytes / t_bcast / 1e9) if (use_dist and t_bcast > 0) else None,
                       "ranks_first_search_equal": ranks_equal,
                       "launcher": "torch.distributed.run, one rank per GPU, RCCL" if use_dist else "single process"},
        }
        if roof is not None:
            line["roofline"] = roof
        if whole is not None:
            if not a.whole:
                # the front + x2 / protect mix as the graph runs them: the two graphs of this run differ by exactly those launches (the event-bracketed
                # eager sum above carries ~1.5 us of event overhead per launch)
                whole["front_and_mix_in_graph_ms"] = round(whole["ms_per_step"] - line["ms_per_step"], 4)
            line["whole_infer"] = whole
        if cpu is not None:
            line["cpu_baseline"] = cpu
        line.update(extra)
        if tgpu is not None:
            line["gpu_torch_baseline"] = tgpu
        if rep_stats is not None:
            line["repeats"] = rep_stats
        emit(line)
    if use_dist:
        dist.barrier()  # the other ranks wait for rank 0's untimed roofline / whole-infer legs before tearing RCCL down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
