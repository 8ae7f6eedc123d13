"""The part of ``SynthesizerTrnMsNSFsid.infer`` (rvc/layers/synthesizers.py:160-203) that runs before the
generator -- ``enc_p`` (TextEncoder, rvc/layers/encoders.py:86-159), the prior sample ``z_p`` and the reversed
``flow`` (rvc/layers/residuals.py:265-333) -- on the hand-written HIP kernels of ``csrc/front*.h*`` through the
C ABI (``rvcmi_front_*`` in include/rvcmi.h).  SURVEY.md section 8f row 1.

    front = FrontHIP.from_reference(net_g)            # after net_g.remove_weight_norm()
    z = front(phone, pitch, phone_lengths, g, flow_head=0)      # == flow(z_p, x_mask, g, reverse=True) * x_mask

The one RNG draw of this stage (``torch.randn_like(m_p)``, synthesizers.py:182/188) is made here with the same
shape on the same device unless ``noise=`` is given.  No CPU fallback: a non-GPU device raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib

_CFG_KEYS = ("in_channels", "inter_channels", "hidden_channels", "filter_channels", "n_heads", "n_layers", "kernel_size",
             "window_size", "gin_channels", "use_f0", "flow_n_flows", "flow_n_layers", "flow_kernel_size",
             "flow_dilation_rate")


def front_config_from_reference(net_g: torch.nn.Module) -> dict:
    """Recover the hyper-parameters of a reference synthesizer's ``enc_p`` / ``flow`` from the modules themselves
    (constructor arguments: rvc/layers/synthesizers.py:60-113)."""
    enc, flow = net_g.enc_p, net_g.flow
    att = enc.encoder.attn_layers[0]
    c0 = flow.flows[0]
    return dict(
        in_channels=int(enc.emb_phone.in_features), inter_channels=int(enc.out_channels),
        hidden_channels=int(enc.hidden_channels), filter_channels=int(enc.filter_channels), n_heads=int(enc.n_heads),
        n_layers=int(enc.n_layers), kernel_size=int(enc.kernel_size), window_size=int(att.window_size),
        gin_channels=int(flow.gin_channels), use_f0=hasattr(enc, "emb_pitch"), flow_n_flows=int(flow.n_flows),
        flow_n_layers=int(flow.n_layers), flow_kernel_size=int(flow.kernel_size), flow_dilation_rate=int(flow.dilation_rate),
    )


class FrontHIP(torch.nn.Module):
    def __init__(self, cfg: dict, weights: Dict[str, torch.Tensor], device="cuda:0", operand: str = "fp16",
                 max_B: int = 1, max_T: int = 256):
        super().__init__()
        self.cfg = {k: cfg[k] for k in _CFG_KEYS}
        if operand not in _lib.OPERANDS or _lib.OPERANDS[operand] == 0:
            raise ValueError("front operand must be 'fp16' or 'bf16'")
        self.operand = operand
        if operand == "bf16":
            from .nsf import _warn_bf16

            _warn_bf16("encoder / flow: 1.4e-2 RMS on z")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.RvcmiError("the HIP front needs a GPU device (got %s); there is no CPU fallback" % device)
        self._weights = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in weights.items()
                         if k.startswith(("enc_p.", "flow."))}
        self._handle = C.c_void_p(None)
        self._max_B = self._max_T = 0
        self._ensure(max_B, max_T)

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
            arr[i].name, arr[i].data, arr[i].ndim = kb, t.data_ptr(), t.dim()
            for j, s in enumerate(t.shape):
                arr[i].shape[j] = s
        cs = _lib.FrontConfig()
        for k in _CFG_KEYS:
            setattr(cs, k, int(self.cfg[k]))
        cs.operand = _lib.OPERANDS[self.operand]
        h = C.c_void_p(None)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(_lib.lib().rvcmi_front_create(C.byref(cs), arr, len(names), idx, B, T, C.byref(h)))
        self._handle, self._max_B, self._max_T = h, B, T
        for k, v in getattr(self, "_options", {}).items():
            _lib.set_option(_lib.lib().rvcmi_front_set_option, self._handle, k, v)

    def _destroy(self) -> None:
        if getattr(self, "_handle", None):
            _lib.lib().rvcmi_front_destroy(self._handle)
            self._handle = C.c_void_p(None)

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def reserve(self, max_B: int, max_T: int) -> "FrontHIP":
        self._ensure(max_B, max_T)
        return self

    @property
    def workspace_bytes(self) -> int:
        return int(_lib.lib().rvcmi_front_workspace_bytes(self._handle))

    def _run(self, phone, pitch, lengths, g, noise, flow_head, tap: Optional[str] = None):
        if phone.dim() != 3 or phone.shape[2] != self.cfg["in_channels"]:
            raise ValueError("phone must be [B, T, %d], got %s" % (self.cfg["in_channels"], tuple(phone.shape)))
        B, T, _ = phone.shape
        fh = int(flow_head or 0)
        if not 0 <= fh < T:
            raise ValueError("flow_head %d out of range for T=%d" % (fh, T))
        dev, IC = self.device, self.cfg["inter_channels"]
        out_dtype = phone.dtype
        ph = phone.to(dev, torch.float32).contiguous()
        pi = None
        if self.cfg["use_f0"]:
            if pitch is None or tuple(pitch.shape) != (B, T):
                raise ValueError("pitch must be [B, T] for an f0 model")
            pi = pitch.to(dev, torch.int64).contiguous()
        ln = None if lengths is None else lengths.to(dev, torch.int64).contiguous()
        gf = None
        if self.cfg["gin_channels"]:
            if g is None:
                raise ValueError("g is required (gin_channels = %d)" % self.cfg["gin_channels"])
            gf = g.to(dev, torch.float32).reshape(B, -1).contiguous()
            if gf.shape[1] != self.cfg["gin_channels"]:
                raise ValueError("g must carry %d channels" % self.cfg["gin_channels"])
        if noise is None:
            noise = torch.randn(B, IC, T - fh, device=phone.device, dtype=phone.dtype)  # randn_like(m_p)
        if tuple(noise.shape) != (B, IC, T - fh):
            raise ValueError("noise must be [B, %d, %d]" % (IC, T - fh))
        nz = noise.to(dev, torch.float32).contiguous()
        self._ensure(B, T)
        stream = torch.cuda.current_stream(dev).cuda_stream
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)
        L = _lib.lib()
        with torch.cuda.device(dev):
            if tap is None:
                out = torch.empty(B, IC, T - fh, device=dev, dtype=torch.float32)
                _lib.check(L.rvcmi_front_forward(self._handle, B, T, ptr(ph), ptr(pi), ptr(ln), ptr(gf), ptr(nz), fh, ptr(out),
                                                 C.c_void_p(stream)))
                return out.to(out_dtype)
            cap = B * T * 192
            host = np.empty(cap, dtype=np.float32)
            shape = (C.c_int64 * 3)()
            _lib.check(L.rvcmi_front_debug_forward(self._handle, B, T, ptr(ph), ptr(pi), ptr(ln), ptr(gf), ptr(nz), fh, tap.encode(),
                                                   host.ctypes.data_as(C.c_void_p), cap, shape, C.c_void_p(stream)))
            n = shape[0] * shape[1] * shape[2]
            return torch.from_numpy(host[:n].reshape(shape[0], shape[1], shape[2]).copy())

    def forward(self, phone: torch.Tensor, pitch: Optional[torch.Tensor], lengths: Optional[torch.Tensor],
                g: Optional[torch.Tensor], flow_head: int = 0, *, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """-> ``z * x_mask`` [B, inter, T - flow_head], the tensor ``infer`` hands to ``self.dec``."""
        return self._run(phone, pitch, lengths, g, noise, flow_head)

    def debug_tap(self, what: str, phone, pitch, lengths, g, flow_head: int = 0, noise=None) -> torch.Tensor:
        """Internal stage, channels-last [B, T', 192] on the host (see rvcmi_front_debug_forward)."""
        return self._run(phone, pitch, lengths, g, noise, flow_head, tap=what)

    def set_option(self, key: str, value=None) -> None:
        """Dev / test option of this handle (``rvcmi_front_set_option``: ``FR_NJ``, ``FR_NO_FFN_FUSION``, ``FR_FFN_SPLIT``, ``FR_WN_SPLIT`` 0 / 1 / 2 = a WN
        layer as one launch / gate + res_skip launches, channel pairs over 3x the blocks (bit-identical to 0) / the gate's taps over the waves
        (default for small grids)); ``None`` = default."""
        _lib.set_option(_lib.lib().rvcmi_front_set_option, self._handle, key, value)  # raises on a key this handle does not honour
        if not hasattr(self, "_options"):
            self._options = {}
        if value is None:
            self._options.pop(key, None)
        else:
            self._options[key] = value

    def profile(self, enable: bool) -> None:
        _lib.check(_lib.lib().rvcmi_front_profile_enable(self._handle, 1 if enable else 0))

    def profile_read(self, reset: bool = True) -> List[dict]:
        return _lib.read_stats(_lib.lib().rvcmi_front_profile_read, self._handle, reset)

    @classmethod
    def from_reference(cls, net_g: torch.nn.Module, device="cuda:0", operand: str = "fp16", **kw) -> "FrontHIP":
        """Build from a reference synthesizer AFTER ``remove_weight_norm()`` (rvc/synthesizer.py:27)."""
        sd = net_g.state_dict()
        bad = [k for k in sd if k.startswith(("enc_p.", "flow.")) and ("parametrizations" in k or k.endswith(("weight_g", "weight_v")))]
        if bad:
            raise ValueError("weight norm is still attached (%s ...): call remove_weight_norm() first" % bad[0])
        return cls(front_config_from_reference(net_g), sd, device=device, operand=operand, **kw)


def infer_hip(net_g, front: FrontHIP, phone, phone_lengths, sid, pitch=None, pitchf=None, skip_head=None,
              return_length=None, return_length2=None, *, noise_zp=None, noise_dec=None, ragged: bool = False):
    """``SynthesizerTrnMsNSFsid.infer`` (synthesizers.py:160-203) with enc_p / flow / dec all on the HIP path.
    Same arguments, same RNG consumption (randn_like(m_p), then the generator's two draws).

    ``ragged=True`` (not in the reference): the batch items are independent utterances of ``phone_lengths`` frames and each one
    must come out as a SEPARATE call would produce it -- the generator is told the lengths (every conv zero-pads behind the item's
    own end).  The reference's padded batch, ``ragged=False``, lets the rows behind a short item leak into its last frames."""
    g = net_g.emb_g(sid).unsqueeze(-1)
    T = phone.shape[1]
    if skip_head is not None and return_length is not None:
        head, length = int(skip_head), int(return_length)
        flow_head = max(head - 24, 0)
        dec_head = head - flow_head
        z = front(phone, pitch, phone_lengths, g, flow_head, noise=noise_zp)
        z = z[:, :, dec_head:dec_head + length]
        if pitchf is not None:
            pitchf = pitchf[:, head:head + length]
    else:
        z = front(phone, pitch, phone_lengths, g, 0, noise=noise_zp)
    # the reference's dispatch and its error (synthesizers.py:190-201)
    from .nsf import GeneratorHIP, NSFGeneratorHIP

    rag = {"lengths": phone_lengths} if ragged else {}
    if ragged and (skip_head is not None or return_length2 is not None):
        raise ValueError("ragged batches and the realtime partial decode cannot be combined")
    if pitchf is not None and isinstance(net_g.dec, NSFGeneratorHIP):
        return net_g.dec(z, pitchf, g=g, n_res=return_length2, **rag, **({"noise": noise_dec} if noise_dec is not None else {}))
    if isinstance(net_g.dec, GeneratorHIP):
        return net_g.dec(z, g=g, n_res=return_length2, **rag)
    if not isinstance(net_g.dec, (NSFGeneratorHIP, GeneratorHIP)) and callable(net_g.dec):  # a foreign dec (tests' stand-ins)
        if ragged:  # it cannot be told the lengths: the padded-batch result would come back labelled per-item exact
            raise TypeError("ragged=True needs the HIP generator as net_g.dec (got %s): a foreign dec computes the padded batch, "
                            "whose short items are contaminated by the rows behind them" % type(net_g.dec).__name__)
        if pitchf is not None:
            return net_g.dec(z, pitchf, g=g, n_res=return_length2, **({"noise": noise_dec} if noise_dec is not None else {}))
        return net_g.dec(z, g=g, n_res=return_length2)
    raise KeyError("unknown dec type: %s" % type(net_g.dec).__name__)
