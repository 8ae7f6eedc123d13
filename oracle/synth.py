"""TEST INFRASTRUCTURE ONLY -- seeded synthetic weights, inputs and IVF indices.

No model checkpoint (``*.pth``), HuBERT/RMVPE weight or faiss ``.index`` exists offline, so
every test / bench input is generated here from fixed seeds.  All
generation uses the torch *CPU* generator or numpy ``default_rng`` and is therefore
reproducible on the GPU box without shipping megabytes of fixtures; golden files carry a
sha256 of the weights so a silent RNG change is caught.
"""
from __future__ import annotations

import hashlib
import math
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .nsf_oracle import GenConfig


def _randn(gen: torch.Generator, *shape) -> torch.Tensor:
    return torch.randn(*shape, generator=gen, dtype=torch.float32)


def make_dec_weights(cfg: GenConfig, seed: int = 1234) -> "OrderedDict[str, torch.Tensor]":
    """Variance-preserving random generator weights, keyed like ``net_g.dec.state_dict()``
    after ``remove_weight_norm()`` (rvc/synthesizer.py:25-27, shapes: SURVEY.md 8a).

    The constructor's own ``normal_(0, 0.01)`` init (rvc/layers/utils.py:6-11) makes every
    ResBlock branch numerically negligible, which would make parity tests blind to the
    heavy kernels; these scales keep every activation O(1).
    """
    gen = torch.Generator().manual_seed(seed)
    w: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    c0 = cfg.upsample_initial_channel
    w["conv_pre.weight"] = _randn(gen, c0, cfg.inter_channels, 7) / math.sqrt(cfg.inter_channels * 7)
    w["conv_pre.bias"] = 0.1 * _randn(gen, c0)
    if cfg.gin_channels:
        w["cond.weight"] = 0.5 * _randn(gen, c0, cfg.gin_channels, 1) / math.sqrt(cfg.gin_channels)
        w["cond.bias"] = 0.1 * _randn(gen, c0)
    if cfg.use_f0:
        w["m_source.l_linear.weight"] = torch.tensor([[2.5]])
        w["m_source.l_linear.bias"] = torch.tensor([0.1])
    n_up = len(cfg.upsample_rates)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        cin, cout = c0 // 2 ** i, c0 // 2 ** (i + 1)
        w[f"ups.{i}.weight"] = 1.4 * _randn(gen, cin, cout, k) / math.sqrt(cin * k / u)
        w[f"ups.{i}.bias"] = 0.1 * _randn(gen, cout)
        if cfg.use_f0:
            if i + 1 < n_up:
                s = math.prod(cfg.upsample_rates[i + 1:])
                w[f"noise_convs.{i}.weight"] = 3.0 * _randn(gen, cout, 1, 2 * s) / math.sqrt(2 * s)
            else:
                w[f"noise_convs.{i}.weight"] = 3.0 * _randn(gen, cout, 1, 1)
            w[f"noise_convs.{i}.bias"] = 0.1 * _randn(gen, cout)
    nk = cfg.num_kernels
    for i in range(n_up):
        ch = c0 // 2 ** (i + 1)
        for j, (k, dils) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            n = i * nk + j
            for c, gain in (("convs1", 1.4), ("convs2", 0.7)):
                for m in range(len(dils)):
                    w[f"resblocks.{n}.{c}.{m}.weight"] = gain * _randn(gen, ch, ch, k) / math.sqrt(ch * k)
                    w[f"resblocks.{n}.{c}.{m}.bias"] = 0.1 * _randn(gen, ch)
    ch = c0 // 2 ** n_up
    w["conv_post.weight"] = 0.25 * _randn(gen, 1, ch, 7) / math.sqrt(ch * 7)
    return w


def weights_sha256(w: Dict[str, torch.Tensor]) -> str:
    h = hashlib.sha256()
    for k in w:
        h.update(k.encode())
        h.update(w[k].contiguous().numpy().tobytes())
    return h.hexdigest()


def make_f0(B: int, T: int) -> torch.Tensor:
    """pitchf[t] = 220*2^sin(2*pi*t/300) Hz, frames t%200<40 unvoiced (0)."""
    t = torch.arange(T, dtype=torch.float32)
    f0 = 220.0 * torch.pow(2.0, torch.sin(2 * math.pi * t / 300.0))
    f0[(torch.arange(T) % 200) < 40] = 0.0
    out = f0.unsqueeze(0).repeat(B, 1)
    for b in range(B):  # de-correlate batch rows a little
        out[b] = torch.roll(out[b], shifts=17 * b) * (1.0 + 0.03 * b)
    return out.contiguous()


def make_dec_inputs(cfg: GenConfig, B: int, T: int, seed: int = 1234):
    """z [B,inter,T] ~ N(0,1) (what flow^-1 hands the decoder), f0 [B,T], g [B,gin,1]."""
    gen = torch.Generator().manual_seed(seed + 1)
    z = _randn(gen, B, cfg.inter_channels, T)
    g = _randn(gen, B, cfg.gin_channels, 1) if cfg.gin_channels else None
    f0 = make_f0(B, T) if cfg.use_f0 else None
    return z, f0, g


def make_phone(B: int, T: int, d: int = 768, seed: int = 1234) -> torch.Tensor:
    """phone ~ N(0,1)*0.5, seed 1234+b."""
    rows = []
    for b in range(B):
        gen = torch.Generator().manual_seed(seed + b)
        rows.append(0.5 * _randn(gen, T, d))
    return torch.stack(rows)


# ----------------------------------------------------------------------------------------------
# IVF-Flat index synthesis (web.py:499-571 recipe, without faiss / sklearn in the loop)
# ----------------------------------------------------------------------------------------------

def ivf_nlist(n: int) -> int:
    """web.py:544  n_ivf = min(int(16*sqrt(N)), N//39)."""
    return max(1, min(int(16 * np.sqrt(n)), n // 39))


def make_ivf(n: int, d: int, nlist: Optional[int] = None, seed: int = 4321, kmeans_iters: int = 2,
             dup: int = 0) -> dict:
    """Build an IVF-Flat index the way ``index.train(); index.add()`` would lay it out:
    ``nlist`` centroids, every vector assigned to its exact nearest centroid (fp64), ids
    sequential in add order (web.py:561-563), list contents in id order.

    Centroids: sampled rows refined by a couple of Lloyd iterations (faiss' own k-means is not
    reproducible offline; retrieval semantics do not depend on how the centroids were found).
    ``dup`` > 0 plants exact duplicate vectors to exercise tie handling.
    """
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d), dtype=np.float32)
    if dup:
        src = rng.integers(0, n, size=dup)
        dst = rng.integers(0, n, size=dup)
        x[dst] = x[src]
    if nlist is None:
        nlist = ivf_nlist(n)
    cent = x[rng.choice(n, size=nlist, replace=False)].copy()
    assign = None
    for it in range(kmeans_iters + 1):
        assign = assign_nearest(x, cent)
        if it == kmeans_iters:
            break
        for c in range(nlist):
            m = assign == c
            if m.any():
                cent[c] = x[m].astype(np.float64).mean(0).astype(np.float32)
    order = np.argsort(assign, kind="stable")
    sizes = np.bincount(assign, minlength=nlist).astype(np.int64)
    offsets = np.zeros(nlist + 1, dtype=np.int64)
    np.cumsum(sizes, out=offsets[1:])
    return dict(d=d, ntotal=n, nlist=nlist, nprobe=1, centroids=cent,
                list_offsets=offsets, ids=order.astype(np.int64), vecs=x[order].copy(), xb=x)


def assign_nearest(x: np.ndarray, cent: np.ndarray, chunk: int = 4096) -> np.ndarray:
    """Exact (fp64) nearest-centroid assignment, ties -> lowest centroid id."""
    c64 = cent.astype(np.float64)
    cn = (c64 * c64).sum(1)
    out = np.empty(x.shape[0], dtype=np.int64)
    for i in range(0, x.shape[0], chunk):
        q = x[i:i + chunk].astype(np.float64)
        dist = (q * q).sum(1)[:, None] + cn[None, :] - 2.0 * (q @ c64.T)
        out[i:i + chunk] = np.argmin(dist, axis=1)
    return out


def make_legacy_checkpoint(cfg: GenConfig, version: str = "v2", seed: int = 1234) -> Tuple[dict, dict]:
    """A checkpoint dict in the reference's ``.pth`` schema (infer/lib/train/process_ckpt.py:15-57):
    fp16 weights, legacy ``weight_g/weight_v`` names for weight-normed layers, ``config`` positional
    list, ``f0``, ``version``.  Only the ``dec.*`` tensors are synthesised here; the caller (a test
    that has the reference importable) fills ``enc_p/flow/emb_g`` from a freshly built reference net.

    Returns (cpt, folded_fp32_dec_weights_expected).
    """
    w = make_dec_weights(cfg, seed)
    weight = OrderedDict()
    expect = OrderedDict()
    normed = ("ups.", "resblocks.")
    for k, v in w.items():
        if k.endswith(".weight") and k.startswith(normed):
            v16 = v.half()
            gnorm = v16.float().flatten(1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1))).half()
            weight["dec." + k[:-len("weight")] + "weight_g"] = gnorm
            weight["dec." + k[:-len("weight")] + "weight_v"] = v16
            vv = v16.float()
            nrm = vv.flatten(1).norm(dim=1).reshape(gnorm.shape)
            expect[k] = vv * (gnorm.float() / nrm)
        else:
            weight["dec." + k] = v.half()
            expect[k] = v.half().float()
    config = [1025, 32, cfg.inter_channels, 192, 768, 2, 6, 3, 0, "1",
              cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes, cfg.upsample_rates,
              cfg.upsample_initial_channel, cfg.upsample_kernel_sizes, 109, cfg.gin_channels, cfg.sr]
    cpt = OrderedDict(weight=weight, config=config, f0=1 if cfg.use_f0 else 0, version=version,
                      info="synthetic", sr={32000: "32k", 40000: "40k", 48000: "48k"}[cfg.sr])
    return cpt, expect
