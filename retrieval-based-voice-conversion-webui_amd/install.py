"""Zero-edit installation behind the stock WebUI / CLI / realtime GUI.

    import rvc_amd; rvc_amd.install()        # once, at process start (e.g. in sitecustomize or the first line of web.py)

rebinds the two names the reference resolves its hot path through:

* ``rvc.synthesizer.get_synthesizer`` / ``load_synthesizer`` (rvc/synthesizer.py:10,31) -- also inside every module that took
  them with ``from rvc.synthesizer import ...`` before the call (infer/modules/vc/modules.py:12, infer/modules/vc/hash.py:11,
  infer/lib/rtrvc.py:15), so ``VC.get_vc`` and ``RVC.__init__`` receive a ``net_g`` whose ``dec`` / ``infer`` run on HIP;
* the name ``faiss`` as seen by ``infer.modules.vc.pipeline`` (pipeline.py:11,214-215,126) and ``infer.lib.rtrvc``
  (rtrvc.py:7,56-57,130-131,172): ``faiss.read_index`` returns an ``IVFFlatHIP`` (``.ntotal``, ``.reconstruct_n``, ``.search``);
  every other attribute is forwarded to the real ``faiss`` module when one is installed (index training in web.py keeps
  working), and raises a clear error when it is not.

Nothing of the reference is edited; ``uninstall()`` restores the original bindings.
"""
from __future__ import annotations

import importlib
import sys
import types
from typing import Optional

import torch

from . import ivf as _ivf
from . import synthesizer as _syn

_state: dict = {}
_LOADER_USERS = ("infer.modules.vc.modules", "infer.modules.vc.hash", "infer.lib.rtrvc")
_FAISS_USERS = ("infer.modules.vc.pipeline", "infer.lib.rtrvc")


class _FaissShim(types.ModuleType):
    """``faiss`` as the inference path sees it; ``read_index`` is served by the HIP index."""

    def __init__(self, real: Optional[types.ModuleType], device):
        super().__init__("faiss")
        self.__dict__["_rvcmi_real"] = real
        self.__dict__["_rvcmi_device"] = device
        self.__dict__["__doc__"] = "rvc_amd faiss shim (read_index -> IVFFlatHIP); other names forwarded to the real faiss"

    def read_index(self, path, *a, **k):
        return _ivf.read_index(path, device=self._rvcmi_device)

    def write_index(self, index, path):
        if isinstance(index, _ivf.IVFFlatHIP):
            return _ivf.write_index(index, path)
        return self.__getattr__("write_index")(index, path)

    def __getattr__(self, name):
        real = self.__dict__.get("_rvcmi_real")
        if real is None:
            raise AttributeError("faiss.%s: faiss is not installed and rvc_amd only replaces read_index / write_index / "
                                 "the index object's search / reconstruct_n / ntotal" % name)
        return getattr(real, name)


def install(operand: str = "fp16", front: bool = True, device="cuda:0", patch_faiss: bool = True) -> None:
    """Route the reference's loader and index reader through the HIP path (idempotent)."""
    if _state.get("installed"):
        return
    import rvc.synthesizer as rs  # the reference package must be importable: this IS the plug-in boundary

    ref_get, ref_load = rs.get_synthesizer, rs.load_synthesizer

    def get_synthesizer(cpt, device=torch.device("cpu")):
        return _syn.get_synthesizer(cpt, device, operand=operand, front=front)

    def load_synthesizer(pth_path, device=torch.device("cpu")):
        return _syn.load_synthesizer(pth_path, device, operand=operand, front=front)

    get_synthesizer._rvcmi_original = ref_get
    load_synthesizer._rvcmi_original = ref_load
    rebound = []
    for mod in [rs] + [sys.modules[m] for m in _LOADER_USERS if m in sys.modules]:
        for name, old, new in (("get_synthesizer", ref_get, get_synthesizer), ("load_synthesizer", ref_load, load_synthesizer)):
            if getattr(mod, name, None) is old:
                setattr(mod, name, new)
                rebound.append((mod, name, old))
    _state.update(installed=True, rebound=rebound)
    if patch_faiss:
        try:
            real = importlib.import_module("faiss")
            if isinstance(real, _FaissShim):
                real = real._rvcmi_real
        except ImportError:
            real = None
        shim = _FaissShim(real, device)
        _state["faiss_prev"] = sys.modules.get("faiss")
        sys.modules["faiss"] = shim  # modules imported from now on bind the shim with their `import faiss`
        for m in _FAISS_USERS:  # ... and the ones already imported are rebound
            mod = sys.modules.get(m)
            if mod is not None and hasattr(mod, "faiss"):
                rebound.append((mod, "faiss", mod.faiss))
                mod.faiss = shim


def uninstall() -> None:
    if not _state.get("installed"):
        return
    for mod, name, old in reversed(_state.get("rebound", [])):
        setattr(mod, name, old)
    if "faiss_prev" in _state:
        if _state["faiss_prev"] is None:
            sys.modules.pop("faiss", None)
        else:
            sys.modules["faiss"] = _state["faiss_prev"]
    _state.clear()
