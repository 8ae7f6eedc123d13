"""TEST INFRASTRUCTURE ONLY -- torch-CPU fp32 restatement of what runs BEFORE the generator inside
``SynthesizerTrnMsNSFsid.infer``:

* ``TextEncoder.forward``                     rvc/layers/encoders.py:134-159
* ``Encoder.forward``                         rvc/layers/encoders.py:65-82
* ``MultiHeadAttention.forward/_attention``   rvc/layers/attentions.py:74-146 (relative-position window 10)
* ``FFN.forward``                             rvc/layers/attentions.py:262-272
* ``LayerNorm.forward``                       rvc/layers/norms.py:20-23
* prior sampling ``z_p``                      rvc/layers/synthesizers.py:182-183,188-189
* ``ResidualCouplingBlock.forward(reverse)``  rvc/layers/residuals.py:319-321 (+ Flip :254, coupling :214-238)
* ``WN.forward``                              rvc/layers/norms.py:96-124, gate rvc/layers/utils.py:47-55

Functional: weights dict in (keys of ``net_g.state_dict()`` after ``remove_weight_norm()``, i.e.
``enc_p.*``, ``flow.*``, ``emb_g.weight``), tensors out.  The single RNG draw of this stage
(``torch.randn_like(m_p)``) is replaced by an explicit ``noise`` argument.  The relative-position terms
are written as direct band index arithmetic instead of the reference's pad/reshape skewing trick, so this
is an independent statement of the same maths; ``oracle/make_golden.py`` pins it to the reference modules.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class FrontConfig:
    """Positional config list of a checkpoint: [spec, seg, inter, hidden, filter, heads, layers, k, p, ...]."""

    in_channels: int = 768       # 768 for v2 (SynthesizerTrnMs768NSFsid), 256 for v1
    inter_channels: int = 192
    hidden_channels: int = 192
    filter_channels: int = 768
    n_heads: int = 2
    n_layers: int = 6
    kernel_size: int = 3
    window_size: int = 10        # attentions.py Encoder default (encoders.py:21)
    gin_channels: int = 256
    use_f0: bool = True
    flow_n_flows: int = 4        # residuals.py:275
    flow_n_layers: int = 3       # synthesizers.py:112  ResidualCouplingBlock(inter, hidden, 5, 1, 3, gin)
    flow_kernel_size: int = 5
    flow_dilation_rate: int = 1
    spk_embed_dim: int = 109


W = Dict[str, torch.Tensor]


def sequence_mask(lengths: torch.Tensor, T: int) -> torch.Tensor:
    # rvc/layers/utils.py:58-66
    return (torch.arange(T).unsqueeze(0) < lengths.unsqueeze(1)).to(torch.float32)


def layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    # norms.py:20-23, x [B,C,T], statistics over C
    mu = x.mean(dim=1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * gamma.view(1, -1, 1) + beta.view(1, -1, 1)


def attention(cfg: FrontConfig, w: W, pre: str, x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """attentions.py:74-146 for self-attention.  x [B,C,T], mask [B,T] (1 = valid)."""
    B, C, T = x.shape
    H, dk, ws = cfg.n_heads, C // cfg.n_heads, cfg.window_size
    q = F.conv1d(x, w[pre + "conv_q.weight"], w[pre + "conv_q.bias"])
    k = F.conv1d(x, w[pre + "conv_k.weight"], w[pre + "conv_k.bias"])
    v = F.conv1d(x, w[pre + "conv_v.weight"], w[pre + "conv_v.bias"])
    q = q.view(B, H, dk, T).transpose(2, 3) / math.sqrt(dk)  # [B,H,T,dk]  (:98 scales the query)
    k = k.view(B, H, dk, T).transpose(2, 3)
    v = v.view(B, H, dk, T).transpose(2, 3)
    scores = q @ k.transpose(-2, -1)  # [B,H,Tq,Tk]
    # relative keys (:99-108): scores[i,j] += q_i . E_k[j-i+ws] for |j-i| <= ws
    Ek, Ev = w[pre + "emb_rel_k"][0], w[pre + "emb_rel_v"][0]  # [2ws+1, dk] (heads share, :45)
    rel = q @ Ek.t()  # [B,H,T,2ws+1]
    ii = torch.arange(T).unsqueeze(1)
    jj = torch.arange(T).unsqueeze(0)
    r = jj - ii + ws
    band = (r >= 0) & (r <= 2 * ws)
    rc = r.clamp(0, 2 * ws)
    scores = scores + torch.where(band, torch.gather(rel, -1, rc.expand(B, H, T, T)), torch.zeros(()))
    am = mask.unsqueeze(1).unsqueeze(-1) * mask.unsqueeze(1).unsqueeze(2)  # encoders.py:66  [B,1,T,T]
    scores = scores.masked_fill(am == 0, -1e4)  # :114-115
    p = torch.softmax(scores, dim=-1)
    out = p @ v  # [B,H,T,dk]
    # relative values (:127-135): out_i += sum_r p[i, i+r-ws] E_v[r]
    pb = torch.where(band, p, torch.zeros(()))
    for rr in range(2 * ws + 1):
        d = rr - ws
        diag = torch.diagonal(pb, offset=d, dim1=-2, dim2=-1)  # p[i, i+d], i in the valid range
        lo = max(0, -d)
        out[:, :, lo:lo + diag.shape[-1], :] += diag.unsqueeze(-1) * Ev[rr]
    out = out.transpose(2, 3).contiguous().view(B, C, T)
    return F.conv1d(out, w[pre + "conv_o.weight"], w[pre + "conv_o.bias"])


def ffn(cfg: FrontConfig, w: W, pre: str, x: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    # attentions.py:262-272 with _same_padding (:287-295), activation None -> relu
    ks = cfg.kernel_size
    pl, pr = (ks - 1) // 2, ks // 2
    y = F.conv1d(F.pad(x * m, [pl, pr]), w[pre + "conv_1.weight"], w[pre + "conv_1.bias"])
    y = torch.relu(y)
    y = F.conv1d(F.pad(y * m, [pl, pr]), w[pre + "conv_2.weight"], w[pre + "conv_2.bias"])
    return y * m


def text_encoder(cfg: FrontConfig, w: W, phone: torch.Tensor, pitch: Optional[torch.Tensor], lengths: torch.Tensor,
                 skip_head: Optional[int] = None, taps: Optional[dict] = None):
    """encoders.py:134-159.  phone [B,T,in], pitch [B,T] int64 or None -> m, logs [B,inter,T'], x_mask [B,1,T']."""
    x = F.linear(phone, w["enc_p.emb_phone.weight"], w["enc_p.emb_phone.bias"])
    if pitch is not None:
        x = x + w["enc_p.emb_pitch.weight"][pitch]
    x = x * math.sqrt(cfg.hidden_channels)
    x = F.leaky_relu(x, 0.1)
    x = x.transpose(1, 2)  # [B,H,T]
    T = x.shape[2]
    mask = sequence_mask(lengths, T)
    m1 = mask.unsqueeze(1)
    x = x * m1  # :148 and encoders.py:67
    if taps is not None:
        taps["emb"] = x.clone()
    for i in range(cfg.n_layers):
        y = attention(cfg, w, "enc_p.encoder.attn_layers.%d." % i, x, mask)
        x = layer_norm(x + y, w["enc_p.encoder.norm_layers_1.%d.gamma" % i], w["enc_p.encoder.norm_layers_1.%d.beta" % i])
        if taps is not None and i == 0:
            taps["attn0"] = x.clone()
        y = ffn(cfg, w, "enc_p.encoder.ffn_layers.%d." % i, x, m1)
        x = layer_norm(x + y, w["enc_p.encoder.norm_layers_2.%d.gamma" % i], w["enc_p.encoder.norm_layers_2.%d.beta" % i])
        if taps is not None:
            taps["layer%d" % i] = x.clone()
    x = x * m1
    if skip_head is not None:
        x = x[:, :, int(skip_head):]
        m1 = m1[:, :, int(skip_head):]
    stats = F.conv1d(x, w["enc_p.proj.weight"], w["enc_p.proj.bias"]) * m1
    m, logs = torch.split(stats, cfg.inter_channels, dim=1)
    return m, logs, m1


def wn(cfg: FrontConfig, w: W, pre: str, x: torch.Tensor, m1: torch.Tensor, g: Optional[torch.Tensor]) -> torch.Tensor:
    # norms.py:96-124
    Hc = cfg.hidden_channels
    out = torch.zeros_like(x)
    gc = F.conv1d(g, w[pre + "cond_layer.weight"], w[pre + "cond_layer.bias"]) if g is not None else None
    for i in range(cfg.flow_n_layers):
        dil = cfg.flow_dilation_rate ** i
        pad = int((cfg.flow_kernel_size * dil - dil) / 2)
        x_in = F.conv1d(x, w[pre + "in_layers.%d.weight" % i], w[pre + "in_layers.%d.bias" % i], dilation=dil, padding=pad)
        if gc is not None:
            x_in = x_in + gc[:, i * 2 * Hc:(i + 1) * 2 * Hc, :]
        acts = torch.tanh(x_in[:, :Hc]) * torch.sigmoid(x_in[:, Hc:])  # utils.py:47-55
        rs = F.conv1d(acts, w[pre + "res_skip_layers.%d.weight" % i], w[pre + "res_skip_layers.%d.bias" % i])
        if i < cfg.flow_n_layers - 1:
            x = (x + rs[:, :Hc]) * m1
            out = out + rs[:, Hc:]
        else:
            out = out + rs
    return out * m1


def flow_reverse(cfg: FrontConfig, w: W, x: torch.Tensor, m1: torch.Tensor, g: Optional[torch.Tensor]) -> torch.Tensor:
    """residuals.py:319-321: for flow in reversed([c0, Flip, c1, Flip, c2, Flip, c3, Flip])."""
    half = cfg.inter_channels // 2
    for f in reversed(range(cfg.flow_n_flows)):
        x = torch.flip(x, [1])  # :254  (the Flip registered AFTER coupling f runs first in reverse)
        pre = "flow.flows.%d." % (2 * f)
        x0, x1 = x[:, :half], x[:, half:]
        h = F.conv1d(x0, w[pre + "pre.weight"], w[pre + "pre.bias"]) * m1
        h = wn(cfg, w, pre + "enc.", h, m1, g)
        mean = F.conv1d(h, w[pre + "post.weight"], w[pre + "post.bias"]) * m1  # mean_only: logs = 0
        x1 = (x1 - mean) * m1  # :236  exp(-0) = 1
        x = torch.cat([x0, x1], 1)
    return x


def infer_front(cfg: FrontConfig, w: W, phone: torch.Tensor, pitch: Optional[torch.Tensor], lengths: torch.Tensor,
                sid: torch.Tensor, noise: torch.Tensor, flow_head: Optional[int] = None,
                taps: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """synthesizers.py:171-189 up to (not including) the decoder's slicing: returns (z, x_mask, g).

    ``noise`` [B, inter, T'] stands for ``torch.randn_like(m_p)``."""
    g = w["emb_g.weight"][sid].unsqueeze(-1)  # [B,gin,1]
    m, logs, m1 = text_encoder(cfg, w, phone, pitch, lengths, flow_head, taps)
    z_p = (m + torch.exp(logs) * noise * 0.66666) * m1
    if taps is not None:
        taps["m"], taps["logs"], taps["z_p"] = m, logs, z_p
    z = flow_reverse(cfg, w, z_p, m1, g)
    return z, m1, g
