"""``rvc.synthesizer`` surface (rvc/synthesizer.py:10-35) with the synthesizer's compute swapped for the HIP path.

    get_synthesizer(cpt, device)      -> (net_g, cpt)     rvc/synthesizer.py:10
    load_synthesizer(pth_path, device) -> (net_g, cpt)    rvc/synthesizer.py:31

This module calls the reference loader (which must therefore be importable -- it is when this package is dropped
into an RVC checkout), then replaces ``net_g.dec`` with the HIP generator and, with ``front=True`` (default), routes
``net_g.infer`` through the HIP encoder / flow as well (``rvc_amd.front``): the only torch op left inside ``infer`` is
the ``emb_g`` row lookup.  The reference's own modules stay attached to ``net_g`` (``enc_p``, ``flow``), unused.
"""
from __future__ import annotations

import functools

import torch

from .front import FrontHIP, infer_hip
from .nsf import GeneratorHIP, NSFGeneratorHIP


def accelerate_synthesizer(net_g: torch.nn.Module, device=None, operand: str = "fp16", max_B: int = 1, max_T: int = 256,
                           front: bool = True):
    """Swap ``net_g.dec`` (already weight-norm-folded, rvc/synthesizer.py:27) for the HIP generator.
    ``net_g.infer`` (rvc/layers/synthesizers.py:160-203) keeps working unchanged: it type-switches on
    ``isinstance(self.dec, NSFGenerator)`` / ``Generator`` so the replacement classes are registered as
    virtual subclasses of those when they are importable."""
    dec = net_g.dec
    if device is None:
        device = next(net_g.parameters()).device
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("accelerate_synthesizer needs the synthesizer on a GPU (got %s)" % device)
    use_f0 = hasattr(dec, "m_source")
    cls = NSFGeneratorHIP if use_f0 else GeneratorHIP
    new = cls.from_reference(dec, device=device, operand=operand, max_B=max_B, max_T=max_T)
    net_g.dec = _as_reference_subclass(new, dec)
    if front and operand in ("fp16", "f16", "bf16") and hasattr(net_g, "enc_p") and hasattr(net_g, "flow"):
        fr = FrontHIP.from_reference(net_g, device=device, operand=operand, max_B=max_B, max_T=max_T)
        object.__setattr__(net_g, "_rvcmi_front", fr)  # not a registered submodule: net_g.half()/.to() must not touch it
        # same signature as SynthesizerTrnMsNSFsid.infer (synthesizers.py:160-170); instance attribute shadows the method
        object.__setattr__(net_g, "infer", functools.partial(infer_hip, net_g, fr))
    return net_g


def _as_reference_subclass(new, old):
    """``SynthesizerTrnMsNSFsid.infer`` dispatches with isinstance(self.dec, NSFGenerator) (synthesizers.py:190-199).
    Give the replacement a dynamic subclass that also inherits the reference class so that dispatch still
    routes to it; no reference code is copied, only its type identity is reused."""
    ref_cls = type(old)
    name = type(new).__name__
    try:
        # __call__: the reference classes override it with a fixed signature (rvc/layers/nsf.py:136-143) that would come
        # first in the MRO and reject the HIP module's extra keyword (``noise=``); route calls straight to nn.Module's.
        dyn = type(name, (type(new), ref_cls), {"__init__": lambda self, *a, **k: None, "__call__": torch.nn.Module.__call__})
        new.__class__ = dyn
    except TypeError:
        pass
    return new


def _reference_get_synthesizer():
    """The reference's own loader (rvc/synthesizer.py:10); after ``rvc_amd.install()`` rebinds that name, the original."""
    import rvc.synthesizer as rs

    fn = rs.get_synthesizer
    return getattr(fn, "_rvcmi_original", fn)


def get_synthesizer(cpt, device=torch.device("cpu"), operand: str = "fp16", front: bool = True):
    net_g, cpt = _reference_get_synthesizer()(cpt, device)
    if torch.device(device).type == "cuda":
        accelerate_synthesizer(net_g, device, operand, front=front)
    return net_g, cpt


def load_synthesizer(pth_path, device=torch.device("cpu"), operand: str = "fp16", front: bool = True):
    return get_synthesizer(torch.load(pth_path, map_location=torch.device("cpu"), weights_only=True), device, operand, front)
