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

from .front_oracle import FrontConfig
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


def make_front_weights(cfg: FrontConfig, seed: int = 1234) -> "OrderedDict[str, torch.Tensor]":
    """Seeded ``enc_p.* / flow.* / emb_g.weight`` tensors keyed like ``net_g.state_dict()`` after
    ``remove_weight_norm()`` (shapes: rvc/layers/encoders.py:103-116, attentions.py:36-54,230-231,
    residuals.py:190-201, norms.py:50-82).  The constructor's defaults would make the test blind
    (``post`` is zero-initialised, residuals.py:200-201 => the flow is the identity), so every layer gets
    O(1)-preserving random weights, including non-trivial LayerNorm gains and biases."""
    gen = torch.Generator().manual_seed(seed + 77)
    w: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    Hc, Fc, dk = cfg.hidden_channels, cfg.filter_channels, cfg.hidden_channels // cfg.n_heads
    w["enc_p.emb_phone.weight"] = _randn(gen, Hc, cfg.in_channels) / math.sqrt(cfg.in_channels) / 4.0
    w["enc_p.emb_phone.bias"] = 0.02 * _randn(gen, Hc)
    if cfg.use_f0:
        w["enc_p.emb_pitch.weight"] = 0.05 * _randn(gen, 256, Hc)
    for i in range(cfg.n_layers):
        a = "enc_p.encoder.attn_layers.%d." % i
        w[a + "emb_rel_k"] = _randn(gen, 1, 2 * cfg.window_size + 1, dk) * dk ** -0.5
        w[a + "emb_rel_v"] = _randn(gen, 1, 2 * cfg.window_size + 1, dk) * dk ** -0.5
        for nm, gain in (("conv_q", 1.5), ("conv_k", 1.5), ("conv_v", 1.0), ("conv_o", 0.7)):
            w[a + nm + ".weight"] = gain * _randn(gen, Hc, Hc, 1) / math.sqrt(Hc)
            w[a + nm + ".bias"] = 0.1 * _randn(gen, Hc)
        w["enc_p.encoder.norm_layers_1.%d.gamma" % i] = 1.0 + 0.1 * _randn(gen, Hc)
        w["enc_p.encoder.norm_layers_1.%d.beta" % i] = 0.1 * _randn(gen, Hc)
        f = "enc_p.encoder.ffn_layers.%d." % i
        w[f + "conv_1.weight"] = 1.4 * _randn(gen, Fc, Hc, cfg.kernel_size) / math.sqrt(Hc * cfg.kernel_size)
        w[f + "conv_1.bias"] = 0.1 * _randn(gen, Fc)
        w[f + "conv_2.weight"] = 0.7 * _randn(gen, Hc, Fc, cfg.kernel_size) / math.sqrt(Fc * cfg.kernel_size)
        w[f + "conv_2.bias"] = 0.1 * _randn(gen, Hc)
        w["enc_p.encoder.norm_layers_2.%d.gamma" % i] = 1.0 + 0.1 * _randn(gen, Hc)
        w["enc_p.encoder.norm_layers_2.%d.beta" % i] = 0.1 * _randn(gen, Hc)
    pw = _randn(gen, 2 * cfg.inter_channels, Hc, 1) / math.sqrt(Hc)
    pw[cfg.inter_channels:] *= 0.3  # logs rows: keep exp(logs) within a sane range
    w["enc_p.proj.weight"] = pw
    pb = 0.1 * _randn(gen, 2 * cfg.inter_channels)
    pb[cfg.inter_channels:] -= 0.5
    w["enc_p.proj.bias"] = pb
    half = cfg.inter_channels // 2
    for fl in range(cfg.flow_n_flows):
        p = "flow.flows.%d." % (2 * fl)
        w[p + "pre.weight"] = _randn(gen, Hc, half, 1) / math.sqrt(half)
        w[p + "pre.bias"] = 0.1 * _randn(gen, Hc)
        for l in range(cfg.flow_n_layers):
            w[p + "enc.in_layers.%d.weight" % l] = _randn(gen, 2 * Hc, Hc, cfg.flow_kernel_size) / math.sqrt(Hc * cfg.flow_kernel_size)
            w[p + "enc.in_layers.%d.bias" % l] = 0.1 * _randn(gen, 2 * Hc)
            rs = 2 * Hc if l < cfg.flow_n_layers - 1 else Hc
            w[p + "enc.res_skip_layers.%d.weight" % l] = 1.5 * _randn(gen, rs, Hc, 1) / math.sqrt(Hc)
            w[p + "enc.res_skip_layers.%d.bias" % l] = 0.1 * _randn(gen, rs)
        if cfg.gin_channels:
            w[p + "enc.cond_layer.weight"] = 0.5 * _randn(gen, 2 * Hc * cfg.flow_n_layers, cfg.gin_channels, 1) / math.sqrt(cfg.gin_channels)
            w[p + "enc.cond_layer.bias"] = 0.1 * _randn(gen, 2 * Hc * cfg.flow_n_layers)
        w[p + "post.weight"] = 0.7 * _randn(gen, half, Hc, 1) / math.sqrt(Hc)
        w[p + "post.bias"] = 0.05 * _randn(gen, half)
    w["emb_g.weight"] = _randn(gen, cfg.spk_embed_dim, cfg.gin_channels)
    return w


def make_pitch(pitchf: torch.Tensor) -> torch.Tensor:
    """Coarse pitch bins 1..255 from f0 in Hz (rvc/f0/gen.py:34-40: mel scale between 50 and 1100 Hz)."""
    mel = 1127.0 * torch.log(1 + pitchf / 700.0)
    lo, hi = 1127.0 * math.log(1 + 50.0 / 700.0), 1127.0 * math.log(1 + 1100.0 / 700.0)
    mel = torch.where(mel > 0, (mel - lo) * 254.0 / (hi - lo) + 1.0, mel)
    mel = torch.where(mel <= 1, torch.ones(()), mel)
    mel = torch.where(mel > 255, torch.full((), 255.0), mel)
    return torch.round(mel).long()


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
    return make_ivf_from_rows(x, nlist, rng, kmeans_iters)


def make_clustered_rows(n: int, d: int, nclusters: int, spread: float = 0.35, seed: int = 77, return_centres: bool = False):
    """Rows with cluster structure (a mixture of `nclusters` isotropic Gaussians of std `spread` around unit-variance
    centres): k-means on such data gives inverted lists of comparable size, as on real HuBERT features -- i.i.d. Gaussian
    rows (`make_ivf`) have no structure in 768 dimensions and end up in a few huge lists (10000 x 768: largest list 416 rows
    for a mean of 39, and random queries probe lists of ~300 rows on average)."""
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((nclusters, d), dtype=np.float32)
    lab = rng.integers(0, nclusters, size=n)
    x = (cent[lab] + spread * rng.standard_normal((n, d), dtype=np.float32)).astype(np.float32)
    return (x, cent) if return_centres else x


def make_mute_rows(feats: np.ndarray, copies: int = 40, jitter: float = 2e-3, seed: int = 5) -> np.ndarray:
    """A training-set-like matrix with the REAL HuBERT distribution of the reference's mute clip (tests/golden/mute_hubert.npz):
    the 149 rows, one exact second copy of them (exact distance ties between different ids, on top of the duplicate rows the
    clip itself contains), and `copies - 2` jittered copies (near ties)."""
    rng = np.random.default_rng(seed)
    parts = [feats, feats.copy()]
    for _ in range(max(0, copies - 2)):
        parts.append((feats + jitter * rng.standard_normal(feats.shape, dtype=np.float32)).astype(np.float32))
    x = np.concatenate(parts).astype(np.float32)
    return np.ascontiguousarray(x[rng.permutation(x.shape[0])])


def make_ivf_from_rows(x: np.ndarray, nlist: Optional[int] = None, rng=None, kmeans_iters: int = 2, init: Optional[np.ndarray] = None) -> dict:
    """The IVF layout of `make_ivf` for a given training matrix x [n, d] (ids = row numbers, the sequential `add`).
    `init`: starting centroids [nlist, d] instead of sampled rows (a converged k-means on data whose clusters are known)."""
    rng = np.random.default_rng(0) if rng is None else rng
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x.shape
    if nlist is None:
        nlist = ivf_nlist(n)
    cent = x[rng.choice(n, size=nlist, replace=False)].copy() if init is None else np.ascontiguousarray(init, dtype=np.float32).copy()
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


# ----------------------------------------------------------------------------------------------
# Stand-ins at the Pipeline.vc / Pipeline.pipeline boundary (infer/modules/vc/pipeline.py:76-360)
# ----------------------------------------------------------------------------------------------

class FakeHubert:
    """Stands in for the fairseq HuBERT (not installable offline) as ``Pipeline.vc`` calls it (pipeline.py:103-110):
    ``extract_features(source=, padding_mask=, output_layer=)`` -> ``(feats [1, n, d],)`` with the conv stack's frame count
    ``n = (len - 400) // 320 + 1`` and seeded features (a function of ``seed`` and ``n`` only), on the caller's device."""

    def __init__(self, d: int = 768, seed: int = 1234):
        self.d, self.seed, self.calls = d, seed, 0

    def extract_features(self, source, padding_mask, output_layer):
        assert source.dim() == 2 and source.shape[0] == 1 and padding_mask.shape == source.shape and output_layer in (9, 12)
        n = (int(source.shape[1]) - 400) // 320 + 1
        self.calls += 1
        return (make_phone(1, n, self.d, self.seed + n).to(source.device, source.dtype),)

    def final_proj(self, x):  # v1 only (768 -> 256)
        return x[..., :256]


def make_audio16k(n: int, seed: int = 1234) -> np.ndarray:
    """A 16 kHz float32 test signal with loud and quiet stretches (so that the quiet-point search of pipeline.py:219-232 has
    distinct minima): a chirp under a slow envelope plus a little noise."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / 16000.0
    env = 0.55 + 0.45 * np.sin(2 * np.pi * 0.9 * t + 0.3)
    x = 0.3 * env * np.sin(2 * np.pi * (180.0 * t + 40.0 * t * t)) + 0.01 * rng.standard_normal(n)
    return x.astype(np.float32)


def infer_noise(lengths, upp: int, seed: int = 114514):
    """The draws of consecutive ``net_g.infer`` calls from ONE CPU generator seeded like the reference seeds nothing but the
    tests do (``torch.manual_seed(seed)``): per call ``randn(1, 192, T)`` (z_p, synthesizers.py:182), ``rand(1, 1, 1)`` and
    ``randn(1, T * upp, 1)`` (sine source, generators.py:170-193) -> list of (noise_zp [1,192,T], noise_dec [1, T*upp])."""
    gen = torch.Generator().manual_seed(seed)
    out = []
    for T in lengths:
        nz = torch.randn(1, 192, int(T), generator=gen)
        torch.rand(1, 1, 1, generator=gen)
        nd = torch.randn(1, int(T) * upp, 1, generator=gen).squeeze(-1)
        out.append((nz, nd))
    return out


class FakeMel:
    """Stands in for ``rvc.f0.mel.MelSpectrogram`` (needs librosa's filter bank) as ``RMVPE.compute_f0`` calls it
    (rvc/f0/rmvpe.py:113): ``mel_extractor(wav [1, n], center=True)`` -> ``[1, 128, n // 160 + 1]`` (the frame count of a
    centred STFT with hop 160).  The values carry nothing; the fake network below only looks at the frame count."""

    def __call__(self, audio, center=True):
        assert audio.dim() == 2 and audio.shape[0] == 1 and center
        return torch.zeros(1, 128, int(audio.shape[1]) // 160 + 1, device=audio.device, dtype=torch.float32)


class FakeRMVPEModel:
    """Stands in for the RMVPE network (rvc/f0/models.py; its checkpoint is not available offline) as ``RMVPE._mel2hidden``
    calls it (rvc/f0/rmvpe.py:144-163): mel ``[1, 128, n_pad]`` -> salience ``[1, n_pad, 360]``.  The salience is a seeded
    function of the frame count only: a Gaussian bump (peak 0.5 .. 0.95) around a wandering cents bin on voiced frames, noise
    below the 0.03 voicing threshold everywhere else, unvoiced runs of 12 frames every 67 -- made on the CPU in float32 and
    moved to the caller's device, so both sides of a parity test see the same bits."""

    def __init__(self, seed: int = 1234):
        self.seed, self.calls = seed, 0

    def __call__(self, mel):
        n = int(mel.shape[-1])
        self.calls += 1
        rng = np.random.default_rng(self.seed + n)
        sal = (rng.random((n, 360), dtype=np.float32) * np.float32(0.02)).astype(np.float32)
        t = np.arange(n)
        centre = np.rint(150 + 70 * np.sin(t / 23.0) + 25 * np.sin(t / 5.0)).astype(np.int64)
        voiced = (t % 67) >= 12
        peak = (0.5 + 0.45 * rng.random(n)).astype(np.float32)
        w = np.arange(-6, 7)
        bump = np.exp(-0.5 * (w / 2.0) ** 2).astype(np.float32)
        for i in np.nonzero(voiced)[0]:
            k = centre[i] + w
            ok = (k >= 0) & (k < 360)
            sal[i, k[ok]] += peak[i] * bump[ok]
        return torch.from_numpy(sal).unsqueeze(0).to(mel.device, mel.dtype)


class FakeRMVPE:
    """What ``rvc_amd.pipeline`` needs of an ``rvc.f0.rmvpe.RMVPE`` instance (the GPU box has no reference checkout): the mel
    extractor, the network call with its pad-to-32 / crop (rvc/f0/rmvpe.py:144-163), ``device`` and ``is_half``.  No decode."""

    def __init__(self, device, seed: int = 1234):
        self.device, self.is_half = device, False
        self.mel_extractor, self.model = FakeMel(), FakeRMVPEModel(seed)

    def _mel2hidden(self, mel):
        n_frames = mel.shape[-1]
        n_pad = 32 * ((n_frames - 1) // 32 + 1) - n_frames
        if n_pad > 0:
            mel = torch.nn.functional.pad(mel, (0, n_pad), mode="constant")
        return self.model(mel.float())[:, :n_frames]
