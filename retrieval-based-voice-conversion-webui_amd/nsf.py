"""Drop-in replacements for ``net_g.dec``: the reference's ``NSFGenerator``
(rvc/layers/nsf.py:64-206) and ``Generator`` (rvc/layers/generators.py:14-113), executed by the
hand-written HIP kernels in ``csrc/`` through the C ABI of ``include/rvcmi.h``.

Same call signatures and argument meaning as the reference modules:

    NSFGeneratorHIP.forward(x, f0, g=None, n_res=None)   # nsf.py:145
    GeneratorHIP.forward(x, g=None, n_res=None)          # generators.py:70

Differences a caller can observe: (1) the two RNG draws the reference makes inside ``forward``
(``torch.rand(1,1,1)`` then ``torch.randn_like([B,T*upp,1])``, generators.py:164,192) are made here
with the same calls on the same device, so a seeded run consumes the generator identically -- and
may be overridden with ``noise=`` for bit-reproducible parity tests against the CPU path;
(2) MFMA operands are rounded to fp16 (default) or bf16 with fp32 accumulation and an fp32
residual stream; ``operand="fp32"`` selects exact-fp32 kernels.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib


def _warn_bf16(what: str) -> None:
    """bf16 MFMA operands keep 8 mantissa bits: measured error exceeds the 1e-3 RMS parity bar.  fp16
    operands (the reference's own half mode, same MFMA rate, fp32 accumulation here) are the default and the benchmarked type."""
    import warnings

    warnings.warn("rvc_amd: operand='bf16' does not meet the 1e-3 RMS parity bar (%s); use operand='fp16' (default) or 'fp32'" % what,
                  RuntimeWarning, stacklevel=3)


def _cfg_struct(cfg: dict, operand: str) -> _lib.NsfConfig:
    c = _lib.NsfConfig()
    c.inter_channels = int(cfg["inter_channels"])
    c.upsample_initial_channel = int(cfg["upsample_initial_channel"])
    c.gin_channels = int(cfg.get("gin_channels", 0) or 0)
    c.sr = int(cfg["sr"])
    c.use_f0 = 1 if cfg.get("use_f0", True) else 0
    rates, ks = list(cfg["upsample_rates"]), list(cfg["upsample_kernel_sizes"])
    if len(rates) != len(ks) or not 1 <= len(rates) <= _lib.RVCMI_MAX_UPS:
        raise ValueError("upsample_rates / upsample_kernel_sizes mismatch")
    c.n_ups = len(rates)
    for i, (u, k) in enumerate(zip(rates, ks)):
        c.upsample_rates[i] = int(u)
        c.upsample_kernel_sizes[i] = int(k)
    rk, rd = list(cfg["resblock_kernel_sizes"]), list(cfg["resblock_dilation_sizes"])
    if len(rk) != len(rd) or not 1 <= len(rk) <= _lib.RVCMI_MAX_RB:
        raise ValueError("resblock_kernel_sizes / resblock_dilation_sizes mismatch")
    c.n_resblock_kernels = len(rk)
    for j, (k, ds) in enumerate(zip(rk, rd)):
        c.resblock_kernel_sizes[j] = int(k)
        c.n_dilations[j] = len(ds)
        for m, d in enumerate(ds):
            c.resblock_dilation_sizes[j][m] = int(d)
    if operand not in _lib.OPERANDS:
        raise ValueError("operand must be one of %s" % sorted(_lib.OPERANDS))
    c.operand = _lib.OPERANDS[operand]
    return c


class _HipGenerator(torch.nn.Module):
    """Shared implementation; see NSFGeneratorHIP / GeneratorHIP."""

    def __init__(self, cfg: dict, weights: Dict[str, torch.Tensor], device="cuda:0", operand: str = "fp16",
                 max_B: int = 1, max_T: int = 256):
        super().__init__()
        self.cfg = dict(cfg)
        self.operand = operand
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.RvcmiError("the HIP generator needs a GPU device (got %s); there is no CPU fallback" % device)
        if operand == "bf16":
            _warn_bf16("generator: 2.0-2.8e-3 RMS on the waveform")
        self.upp = math.prod(cfg["upsample_rates"])
        self.num_kernels = len(cfg["resblock_kernel_sizes"])
        self.num_upsamples = len(cfg["upsample_rates"])
        # host fp32 copies (kept so the handle can be re-created with a larger workspace)
        self._weights = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in weights.items()}
        self._handle = C.c_void_p(None)
        self._max_B = self._max_T = 0
        self._ensure(max_B, max_T)

    # -- handle management ---------------------------------------------------------------------
    def _ensure(self, B: int, T: int) -> None:
        if self._handle and B <= self._max_B and T <= self._max_T:
            return
        B, T = max(B, self._max_B), max(T, self._max_T)
        self._destroy()
        names = list(self._weights)
        arr = (_lib.Tensor * len(names))()
        keep = []
        for i, k in enumerate(names):
            t = self._weights[k]
            kb = k.encode()
            keep.append(kb)
            arr[i].name = kb
            arr[i].data = t.data_ptr()
            arr[i].ndim = t.dim()
            for j, s in enumerate(t.shape):
                arr[i].shape[j] = s
        cs = _cfg_struct(self.cfg, self.operand)
        h = C.c_void_p(None)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(_lib.lib().rvcmi_nsf_create(C.byref(cs), arr, len(names), idx, B, T, C.byref(h)))
        self._handle, self._max_B, self._max_T = h, B, T
        for k, v in getattr(self, "_options", {}).items():  # options survive a workspace re-creation
            _lib.set_option(_lib.lib().rvcmi_nsf_set_option, self._handle, k, v)

    def set_option(self, key: str, value=None) -> None:
        """Dev / test option of this handle (``rvcmi_nsf_set_option``: e.g. ``RB_STREAM`` 0 / 1 pins the ResBlock kernel family,
        ``RS_SMALL``, ``RS_KL``, ``Y_F16``, ``X0_F16`` / ``X0_F16_NOSTREAM``, ``UPS_BL``, ``CONV_KS``, ``RBF_SMALL``, ``DBG``; round 6: ``POST_DMA`` 0 / 1 =
        conv_post by the register-staged kernel / by LDS-DMA with the next tile in flight (default), ``POST_DMA_OCC``, ``POST_DBG`` (timing ablations); an unknown key is an error); ``None`` restores the default.  The library reads ``RVCMI_<KEY>`` only when a handle is created."""
        _lib.set_option(_lib.lib().rvcmi_nsf_set_option, self._handle, key, value)  # raises on a key this handle does not honour
        if not hasattr(self, "_options"):
            self._options = {}
        if value is None:
            self._options.pop(key, None)
        else:
            self._options[key] = value

    def _destroy(self) -> None:
        if getattr(self, "_handle", None):
            _lib.lib().rvcmi_nsf_destroy(self._handle)
            self._handle = C.c_void_p(None)

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def reserve(self, max_B: int, max_T: int) -> "_HipGenerator":
        """Pre-size the workspace (allocation happens here, never inside forward once sized)."""
        self._ensure(max_B, max_T)
        return self

    @property
    def workspace_bytes(self) -> int:
        return int(_lib.lib().rvcmi_nsf_workspace_bytes(self._handle))

    # the reference calls these on net_g (rvc/synthesizer.py:27, infer/modules/vc/modules.py:94-97)
    def remove_weight_norm(self):
        return None

    def __prepare_scriptable__(self):
        return self

    # -- forward -------------------------------------------------------------------------------
    def _run(self, x, f0, g, n_res, noise, tap: Optional[str] = None, lengths=None):
        if x.dim() != 3 or x.shape[1] != self.cfg["inter_channels"]:
            raise ValueError("x must be [B, %d, T], got %s" % (self.cfg["inter_channels"], tuple(x.shape)))
        B, _, T = x.shape
        dev = self.device
        out_dtype = x.dtype
        xf = x.to(dev, torch.float32).contiguous()
        use_f0 = bool(self.cfg.get("use_f0", True))
        f0f = nf = gf = None
        if use_f0:
            if f0 is None:
                raise ValueError("f0 is required by an NSF generator")
            if tuple(f0.shape) != (B, T):
                raise ValueError("f0 must be [B, T] = %s, got %s" % ((B, T), tuple(f0.shape)))
            if noise is None:
                # the reference's draws, same order / shapes / device (generators.py:164,192)
                torch.rand(1, 1, 1, device=f0.device)
                noise = torch.randn(B, T * self.upp, 1, device=f0.device, dtype=f0.dtype)
            f0f = f0.to(dev, torch.float32).contiguous()
            nf = noise.to(dev, torch.float32).reshape(B, T * self.upp).contiguous()
        if g is not None:
            gf = g.to(dev, torch.float32).reshape(B, -1).contiguous()
            if gf.shape[1] != self.cfg.get("gin_channels", 0):
                raise ValueError("g must carry %d channels" % self.cfg.get("gin_channels", 0))
        Te = T if n_res is None else int(n_res)
        ln = None
        if lengths is not None:
            # a ragged batch: item b = a separate call of lengths[b] frames (include/rvcmi.h rvcmi_nsf_forward).  Host values are
            # validated here; a device tensor is taken as is (no sync) -- the caller vouches for 1 <= lengths[b] <= T
            if n_res is not None or tap is not None:
                raise ValueError("lengths cannot be combined with n_res / debug taps")
            lt = torch.as_tensor(lengths)
            if tuple(lt.shape) != (B,):
                raise ValueError("lengths must hold %d values" % B)
            if lt.device.type != "cuda" and (int(lt.min()) < 1 or int(lt.max()) > T):
                raise ValueError("lengths must lie in [1, %d]" % T)
            ln = lt.to(dev, torch.int32).contiguous()
        self._ensure(B, max(T, Te))
        stream = torch.cuda.current_stream(dev).cuda_stream
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)
        nr = -1 if n_res is None else int(n_res)
        L = _lib.lib()
        with torch.cuda.device(dev):
            if tap is None:
                out = torch.empty(B, 1, Te * self.upp, device=dev, dtype=torch.float32)
                _lib.check(L.rvcmi_nsf_forward(self._handle, B, T, ptr(ln), ptr(xf), ptr(f0f), ptr(gf), ptr(nf), nr, ptr(out),
                                               C.c_void_p(stream)))
                return out.to(out_dtype)
            cap = B * max(self.cfg["upsample_initial_channel"] * Te, Te * self.upp * 256)
            host = np.empty(cap, dtype=np.float32)
            shape = (C.c_int64 * 3)()
            _lib.check(L.rvcmi_nsf_debug_forward(self._handle, B, T, ptr(xf), ptr(f0f), ptr(gf), ptr(nf), nr, tap.encode(),
                                                 host.ctypes.data_as(C.c_void_p), cap, shape, C.c_void_p(stream)))
            n = shape[0] * shape[1] * shape[2]
            return torch.from_numpy(host[:n].reshape(shape[0], shape[1], shape[2]).copy())

    # -- profiling (bench.py) ------------------------------------------------------------------
    def profile(self, enable: bool) -> None:
        _lib.check(_lib.lib().rvcmi_nsf_profile_enable(self._handle, 1 if enable else 0))

    def profile_read(self, reset: bool = True) -> List[dict]:
        return _lib.read_stats(_lib.lib().rvcmi_nsf_profile_read, self._handle, reset)


def config_from_reference(dec: torch.nn.Module) -> dict:
    """Recover the constructor arguments of a reference ``NSFGenerator`` / ``Generator`` instance
    (rvc/layers/nsf.py:65-141, generators.py:15-62) from the module itself."""
    ups = list(dec.ups)
    use_f0 = hasattr(dec, "m_source")
    nk = int(dec.num_kernels)
    rb = list(dec.resblocks)[:nk]
    cfg = dict(
        inter_channels=int(dec.conv_pre.in_channels),
        upsample_initial_channel=int(dec.conv_pre.out_channels),
        gin_channels=int(dec.cond.in_channels) if hasattr(dec, "cond") else 0,
        upsample_rates=[int(u.stride[0]) for u in ups],
        upsample_kernel_sizes=[int(u.kernel_size[0]) for u in ups],
        resblock_kernel_sizes=[int(r.convs1[0].kernel_size[0]) for r in rb],
        resblock_dilation_sizes=[[int(c.dilation[0]) for c in r.convs1] for r in rb],
        use_f0=use_f0,
        sr=int(dec.m_source.l_sin_gen.sampling_rate) if use_f0 else 0,
    )
    return cfg


class NSFGeneratorHIP(_HipGenerator):
    """``net_g.dec`` for f0 models -- signature of rvc/layers/nsf.py:145."""

    def forward(self, x: torch.Tensor, f0: torch.Tensor, g: Optional[torch.Tensor] = None,
                n_res: Optional[int] = None, *, noise: Optional[torch.Tensor] = None, lengths=None) -> torch.Tensor:
        """``lengths`` (keyword, not in the reference's signature): [B] valid frames per item -- a ragged batch whose items come
        out exactly as separate calls of those lengths would (segments of a long file, utterances of a folder, in one launch)."""
        return self._run(x, f0, g, n_res, noise, lengths=lengths)

    def debug_tap(self, what: str, x, f0, g=None, n_res=None, noise=None) -> torch.Tensor:
        return self._run(x, f0, g, n_res, noise, tap=what)

    @classmethod
    def from_reference(cls, dec: torch.nn.Module, device="cuda:0", operand: str = "fp16", **kw) -> "NSFGeneratorHIP":
        """Build from a reference module AFTER ``remove_weight_norm()`` has folded g*v/|v|
        (rvc/synthesizer.py:27): takes ``dec.state_dict()`` as is."""
        cfg = config_from_reference(dec)
        if not cfg["use_f0"]:
            raise ValueError("module has no m_source: use GeneratorHIP.from_reference")
        return cls(cfg, _plain_state_dict(dec), device=device, operand=operand, **kw)


class GeneratorHIP(_HipGenerator):
    """``net_g.dec`` for no-f0 models -- signature of rvc/layers/generators.py:70."""

    def forward(self, x: torch.Tensor, g: Optional[torch.Tensor] = None, n_res: Optional[int] = None, *, lengths=None) -> torch.Tensor:
        return self._run(x, None, g, n_res, None, lengths=lengths)

    def debug_tap(self, what: str, x, g=None, n_res=None) -> torch.Tensor:
        return self._run(x, None, g, n_res, None, tap=what)

    @classmethod
    def from_reference(cls, dec: torch.nn.Module, device="cuda:0", operand: str = "fp16", **kw) -> "GeneratorHIP":
        cfg = config_from_reference(dec)
        if cfg["use_f0"]:
            raise ValueError("module has an m_source: use NSFGeneratorHIP.from_reference")
        return cls(cfg, _plain_state_dict(dec), device=device, operand=operand, **kw)


def _plain_state_dict(dec: torch.nn.Module) -> Dict[str, torch.Tensor]:
    sd = dec.state_dict()
    bad = [k for k in sd if "parametrizations" in k or k.endswith(("weight_g", "weight_v"))]
    if bad:
        raise ValueError("weight norm is still attached (%s ...): call remove_weight_norm() first, as "
                         "rvc/synthesizer.py:27 does" % bad[0])
    return {k: v for k, v in sd.items()}
