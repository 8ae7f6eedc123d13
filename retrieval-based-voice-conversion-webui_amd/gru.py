"""The recurrent layer of the RMVPE f0 network on the HIP kernel of ``csrc/gru.hip`` -- BEYOND the scope table:
the north star leaves RMVPE on PyTorch-ROCm, and everything but this one layer stays there.

``bench.py --e2e`` measured ``nn.GRU(384, 256, bidirectional=True)`` (rvc/f0/e2e.py:50-67, ``E2E.BiGRU``) at 94-142 ms of
a 108-160 ms conversion: MIOpen runs the ~2400 recurrent steps of a 10 s clip one launch (or more) at a time.  ``GRUHIP`` is a module with
``torch.nn.GRU``'s forward contract -- ``(output, h_n)`` for a batch-first input, ``h_0 = 0`` -- whose recurrence is ONE persistent block
per direction; ``accelerate_rmvpe(model)`` swaps it in for every matching ``nn.GRU`` inside an RMVPE network and leaves everything it does
not recognise (other sizes, layers, a non-GPU model) on PyTorch.

    net = rmvpe.model                      # rvc/f0/rmvpe.py: the E2E network
    rvc_amd.accelerate_rmvpe(net)          # net.fc[0].gru -> GRUHIP (same weights)

Operands are fp16 (like the reference's own ``is_half`` RMVPE), accumulation, gates and the state fp32.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class GRUHIP(torch.nn.Module):
    """Inference-only stand-in for a 1-layer, bidirectional, batch-first ``torch.nn.GRU`` with hidden size 256 (rvc/f0/e2e.py:57-63)."""

    def __init__(self, gru: torch.nn.GRU, device=None):
        super().__init__()
        if not supports(gru):
            raise _lib.RvcmiError("GRUHIP: only nn.GRU(I, 256, num_layers=1, bias=True, batch_first=True, bidirectional=True) with I % 16 == 0")
        p = next(gru.parameters())
        dev = torch.device(device) if device is not None else p.device
        if dev.type != "cuda":
            raise _lib.RvcmiError("GRUHIP needs a GPU device (got %s); there is no CPU fallback" % dev)
        self.input_size, self.hidden_size = gru.input_size, gru.hidden_size
        self.batch_first, self.bidirectional, self.num_layers = True, True, 1
        self._device = dev

        def both(name):
            return torch.stack([getattr(gru, name + "_l0").detach(), getattr(gru, name + "_l0_reverse").detach()]).float().cpu().contiguous()

        w_ih, w_hh, b_ih, b_hh = both("weight_ih"), both("weight_hh"), both("bias_ih"), both("bias_hh")
        h = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().rvcmi_gru_create(self.input_size, self.hidden_size, C.c_void_p(w_ih.data_ptr()), C.c_void_p(w_hh.data_ptr()),
                                                   C.c_void_p(b_ih.data_ptr()), C.c_void_p(b_hh.data_ptr()), _lib.device_index(dev), C.byref(h)))
        self._h = h

    def __del__(self):
        h = self.__dict__.pop("_h", None)  # (not through nn.Module.__setattr__: it may be gone at interpreter shutdown)
        if h:
            try:
                _lib.lib().rvcmi_gru_destroy(h)
            except Exception:  # noqa  (interpreter shutdown)
                pass

    def flatten_parameters(self):  # (nn.GRU API the callers of a loaded checkpoint may use)
        pass

    def forward(self, x: torch.Tensor, hx=None):
        if hx is not None:
            raise _lib.RvcmiError("GRUHIP: h_0 must be None (zeros), as in rvc/f0/e2e.py:66")
        if x.dim() != 3 or x.shape[-1] != self.input_size:
            raise ValueError("GRUHIP: expected [B, T, %d], got %s" % (self.input_size, tuple(x.shape)))
        if x.device.type != "cuda":
            raise _lib.RvcmiError("GRUHIP input must live on the GPU (got %s)" % x.device)
        B, T = int(x.shape[0]), int(x.shape[1])
        H = self.hidden_size
        y = torch.empty(B, T, 2 * H, device=x.device, dtype=torch.float32)
        hn = torch.empty(2, B, H, device=x.device, dtype=torch.float32)
        if B and T:
            x16 = x.detach().to(torch.float16).contiguous()
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib().rvcmi_gru_forward(self._h, B, T, C.c_void_p(x16.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(hn.data_ptr()),
                                                        C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        return y.to(x.dtype), hn.to(x.dtype)


def supports(m) -> bool:
    return (isinstance(m, torch.nn.GRU) and m.num_layers == 1 and m.bidirectional and m.batch_first and m.bias and m.hidden_size == 256
            and m.input_size >= 16 and m.input_size % 16 == 0 and getattr(m, "proj_size", 0) == 0)


def accelerate_rmvpe(model: torch.nn.Module) -> int:
    """Replace every supported ``nn.GRU`` inside ``model`` (on a GPU) by ``GRUHIP`` with the same weights, in place.  -> how many were
    replaced (0: nothing matched, the model is untouched and keeps running on PyTorch)."""
    n = 0
    for parent in list(model.modules()):
        for name, child in list(parent.named_children()):
            if supports(child) and next(child.parameters()).device.type == "cuda":
                setattr(parent, name, GRUHIP(child))
                n += 1
    return n


def accelerate_f0_rmvpe(rmvpe) -> int:
    """What the rebound ``Pipeline.pipeline`` / ``RVC.infer`` do with the ``RMVPE`` object (rvc/f0/rmvpe.py) of their f0 generator, once:
    ``accelerate_rmvpe(rmvpe.model)`` unless ``RVCMI_RMVPE_GRU=0``.  The count is remembered on the object (``_rvcmi_gru``)."""
    import os

    n = getattr(rmvpe, "_rvcmi_gru", None)
    if n is None:
        n = 0
        net = getattr(rmvpe, "model", None)
        if os.environ.get("RVCMI_RMVPE_GRU", "1") != "0" and isinstance(net, torch.nn.Module):
            n = accelerate_rmvpe(net)
        try:
            rmvpe._rvcmi_gru = n
        except Exception:  # noqa  (an object without a __dict__: try again next time)
            pass
    return n

