"""TEST INFRASTRUCTURE ONLY -- torch-CPU fp32 restatement of the RVC generator.

A functional (weights-dict in, tensor out) restatement of

* ``SineGenerator.forward/_f02sine/_f02uv``  rvc/layers/generators.py:148-202
* ``SourceModuleHnNSF.forward``              rvc/layers/nsf.py:57-61
* ``NSFGenerator.forward``                   rvc/layers/nsf.py:145-191
* ``Generator.forward`` (no-f0 models)       rvc/layers/generators.py:70-98
* ``ResBlock1.forward``                      rvc/layers/residuals.py:68-85

The RNG draws the reference makes inside ``forward`` (``torch.rand(1,1,1)`` then
``torch.randn_like([B, T*upp, 1])``, generators.py:164,192) are replaced by an
explicit ``noise`` argument so that the HIP path and the oracle see identical
noise.  Pinned against the reference itself by ``oracle/make_golden.py``.

Weight names are the keys of ``net_g.dec.state_dict()`` *after*
``remove_weight_norm()`` (rvc/synthesizer.py:25-27): ``conv_pre.weight``,
``ups.0.weight`` ([C_in, C_out, k], ConvTranspose layout), ``noise_convs.0.weight``,
``resblocks.0.convs1.0.weight`` ... ``conv_post.weight``, ``cond.weight``,
``m_source.l_linear.weight``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # rvc/layers/residuals.py:16


@dataclass
class GenConfig:
    """Generator hyper-parameters (configs/v2/48k.json:28-45 are the defaults)."""

    inter_channels: int = 192
    resblock_kernel_sizes: List[int] = field(default_factory=lambda: [3, 7, 11])
    resblock_dilation_sizes: List[List[int]] = field(
        default_factory=lambda: [[1, 3, 5], [1, 3, 5], [1, 3, 5]]
    )
    upsample_rates: List[int] = field(default_factory=lambda: [12, 10, 2, 2])
    upsample_initial_channel: int = 512
    upsample_kernel_sizes: List[int] = field(default_factory=lambda: [24, 20, 4, 4])
    gin_channels: int = 256
    sr: int = 48000
    use_f0: bool = True

    @property
    def upp(self) -> int:
        return math.prod(self.upsample_rates)

    @property
    def num_kernels(self) -> int:
        return len(self.resblock_kernel_sizes)


CONFIGS = {
    # configs/v2/48k.json:39-41
    "v2_48k": GenConfig(),
    # configs/v2/32k.json:39-41
    "v2_32k": GenConfig(upsample_rates=[10, 8, 2, 2], upsample_kernel_sizes=[20, 16, 4, 4], sr=32000),
    # configs/v1/40k.json:39-41 (also what "v2 40k" falls back to, web.py:455-456)
    "v1_40k": GenConfig(upsample_rates=[10, 10, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4], sr=40000),
    # configs/v1/32k.json:39-41
    "v1_32k": GenConfig(upsample_rates=[10, 4, 2, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4, 4], sr=32000),
    # configs/v1/48k.json:39-41
    "v1_48k": GenConfig(upsample_rates=[10, 6, 2, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4, 4], sr=48000),
}


def get_padding(kernel_size: int, dilation: int = 1) -> int:
    # rvc/layers/utils.py:14
    return int((kernel_size * dilation - dilation) / 2)


def sine_source(f0: torch.Tensor, upp: int, sr: int, noise: Optional[torch.Tensor]) -> torch.Tensor:
    """generators.py:148-194.  f0 [B,T] Hz (0 = unvoiced); noise [B,T*upp] ~N(0,1) or None (zeros).

    Returns the pre-linear sine waves [B, T*upp] (dim=1 harmonic only, harmonic_num=0).
    """
    f0 = f0.unsqueeze(-1)  # [B,T,1]
    a = torch.arange(1, upp + 1, dtype=f0.dtype, device=f0.device)
    rad = f0 / sr * a  # [B,T,upp]                                   generators.py:154-155
    rad2 = torch.fmod(rad[:, :-1, -1:].float() + 0.5, 1.0) - 0.5  # generators.py:156
    rad_acc = rad2.cumsum(dim=1).fmod(1.0).to(f0)  #                 generators.py:157
    rad = rad + F.pad(rad_acc, (0, 0, 1, 0), mode="constant")  #     generators.py:158
    rad = rad.reshape(f0.shape[0], -1, 1)
    # rand_ini[..., 0] = 0 -> the fundamental gets no random phase   generators.py:164-166
    sines = torch.sin(2 * torch.pi * rad)  #                         generators.py:167
    sine_waves = sines * 0.1  # sine_amp                             generators.py:186
    uv = (f0 > 0).to(f0.dtype)  # voiced_threshold = 0               generators.py:196-198
    uv = F.interpolate(uv.transpose(2, 1), scale_factor=float(upp), mode="nearest").transpose(2, 1)
    noise_amp = uv * 0.003 + (1 - uv) * 0.1 / 3  #                   generators.py:191
    if noise is None:
        noise = torch.zeros_like(sine_waves)
    else:
        noise = noise.reshape(sine_waves.shape)
    sine_waves = sine_waves * uv + noise_amp * noise  #              generators.py:192-193
    return sine_waves.squeeze(-1)


def har_source(w: Dict[str, torch.Tensor], f0, upp, sr, noise) -> torch.Tensor:
    """nsf.py:57-61: tanh(Linear(1->1)(sine)) -> [B,1,T*upp]."""
    s = sine_source(f0, upp, sr, noise)
    lw = w["m_source.l_linear.weight"].reshape(())
    lb = w["m_source.l_linear.bias"].reshape(())
    return torch.tanh(s * lw + lb).unsqueeze(1)


def resblock1(w: Dict[str, torch.Tensor], n: int, x: torch.Tensor, k: int, dils: List[int]) -> torch.Tensor:
    """residuals.py:68-85 (x_mask is None on the inference path)."""
    for j, d in enumerate(dils):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, w[f"resblocks.{n}.convs1.{j}.weight"], w[f"resblocks.{n}.convs1.{j}.bias"],
                      dilation=d, padding=get_padding(k, d))
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, w[f"resblocks.{n}.convs2.{j}.weight"], w[f"resblocks.{n}.convs2.{j}.bias"],
                      dilation=1, padding=get_padding(k, 1))
        x = xt + x
    return x


def generator_forward(
    cfg: GenConfig,
    w: Dict[str, torch.Tensor],
    x: torch.Tensor,
    f0: Optional[torch.Tensor],
    g: Optional[torch.Tensor],
    noise: Optional[torch.Tensor] = None,
    n_res: Optional[int] = None,
    taps: Optional[dict] = None,
) -> torch.Tensor:
    """nsf.py:145-191 (use_f0) / generators.py:70-98 (no f0).

    x [B,inter,T] fp32, f0 [B,T], g [B,gin,1] or None, noise [B,T*upp] or None.
    ``taps`` (optional dict) receives per-stage activations for kernel bring-up.
    """
    upp = cfg.upp
    har = None
    if cfg.use_f0:
        har = har_source(w, f0, upp, cfg.sr, noise)  # nsf.py:152-153
        if n_res is not None:  # nsf.py:155-162 (realtime formant shift)
            n_res = int(n_res)
            if n_res * upp != har.shape[-1]:
                har = F.interpolate(har, size=n_res * upp, mode="linear")
            if n_res != x.shape[-1]:
                x = F.interpolate(x, size=n_res, mode="linear")
    elif n_res is not None:  # generators.py:76-79
        if int(n_res) != x.shape[-1]:
            x = F.interpolate(x, size=int(n_res), mode="linear")
    if taps is not None and har is not None:
        taps["har"] = har
    x = F.conv1d(x, w["conv_pre.weight"], w["conv_pre.bias"], padding=3)  # nsf.py:164
    if g is not None:
        x = x + F.conv1d(g, w["cond.weight"], w["cond.bias"])  # nsf.py:165-166
    if taps is not None:
        taps["pre"] = x
    nk = cfg.num_kernels
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        x = F.leaky_relu(x, LRELU_SLOPE)  # nsf.py:171
        x = F.conv_transpose1d(x, w[f"ups.{i}.weight"], w[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        if cfg.use_f0:
            if i + 1 < len(cfg.upsample_rates):  # nsf.py:103-115
                s = math.prod(cfg.upsample_rates[i + 1:])
                x = x + F.conv1d(har, w[f"noise_convs.{i}.weight"], w[f"noise_convs.{i}.bias"],
                                 stride=s, padding=s // 2)
            else:
                x = x + F.conv1d(har, w[f"noise_convs.{i}.weight"], w[f"noise_convs.{i}.bias"])
        if taps is not None:
            taps[f"up{i}"] = x
        xs = None
        for j in range(nk):  # nsf.py:175-186
            r = resblock1(w, i * nk + j, x, cfg.resblock_kernel_sizes[j], cfg.resblock_dilation_sizes[j])
            xs = r if xs is None else xs + r
        x = xs / nk
        if taps is not None:
            taps[f"stage{i}"] = x
    x = F.leaky_relu(x)  # default slope 0.01, NOT 0.1               nsf.py:187
    x = F.conv1d(x, w["conv_post.weight"], None, padding=3)  #       nsf.py:188
    return torch.tanh(x)  #                                          nsf.py:189


def reference_noise(B: int, T: int, upp: int, seed: int = 114514):
    """Draw the generator-side noise in the reference's order on the CPU generator.

    Order inside ``net_g.infer`` (verified in SURVEY.md 7.1): randn_like([B,192,T]) for z_p
    (synthesizers.py:188) -> rand(1,1,1) (generators.py:164) -> randn_like([B,T*upp,1])
    (generators.py:192).  Here only the last two are needed (the decoder boundary).
    """
    gen = torch.Generator().manual_seed(seed)
    _rand_ini = torch.rand(1, 1, 1, generator=gen)
    return torch.randn(B, T * upp, 1, generator=gen).squeeze(-1)
