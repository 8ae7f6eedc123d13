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

* ``Pipeline.vc`` / ``Pipeline.pipeline`` (infer/modules/vc/pipeline.py:76,186) and ``RVC.infer`` (infer/lib/rtrvc.py:134): replaced
  ON THE REFERENCE'S OWN CLASSES by the device-resident versions (``rvc_amd.pipeline.vc_hip`` / ``pipeline_hip``,
  ``rvc_amd.realtime.rvc_infer_hip``), so the unmodified ``VC.vc_single`` and the realtime GUI loop keep the HuBERT features on
  the GPU through retrieval, x2, protect mix, ``net_g.infer``, RMS mix and int16 scaling (one copy to the host at the end).
  The originals stay reachable (an index object the HIP reader cannot serve is handed back to them).

Nothing of the reference is edited; ``uninstall()`` restores the original bindings.
"""
from __future__ import annotations

import importlib
import sys
import types
from typing import Optional

import torch

from . import _lib
from . import ivf as _ivf
from . import synthesizer as _syn

_state: dict = {}
_LOADER_USERS = ("infer.modules.vc.modules", "infer.modules.vc.hash", "infer.lib.rtrvc")
_PIPELINE_MODULE, _RTRVC_MODULE = "infer.modules.vc.pipeline", "infer.lib.rtrvc"
_FAISS_USERS = ("infer.modules.vc.pipeline", "infer.lib.rtrvc")


class _FaissShim(types.ModuleType):
    """``faiss`` as the inference path sees it; ``read_index`` is served by the HIP index."""

    def __init__(self, real: Optional[types.ModuleType], device):
        super().__init__("faiss")
        self.__dict__["_rvcmi_real"] = real
        self.__dict__["_rvcmi_device"] = device
        self.__dict__["__doc__"] = "rvc_amd faiss shim (read_index -> IVFFlatHIP); other names forwarded to the real faiss"

    def read_index(self, path, *a, **k):
        dev = self._rvcmi_device
        if dev is None:  # the process's current GPU (the reference passes config.device to the loader, not to faiss)
            dev = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        try:
            return _ivf.read_index(path, device=dev)
        except _lib.RvcmiError as e:
            # ONLY "this file is not something the HIP reader serves" (not IVF-Flat / L2, hashed direct map ...: RVCMI_ERR_IO)
            # goes to real faiss; out-of-memory, HIP errors and everything else propagate -- they are not faiss' to paper over
            real = self.__dict__.get("_rvcmi_real")
            if e.code != _lib.ERR_IO or real is None:
                raise
            return real.read_index(path, *a, **k)  # the rebound vc / pipeline / infer hand such an index to the reference code

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


_ABSENT = object()  # `old` of an attribute install() ADDED: uninstall() deletes it


def _rebind_methods(rebound) -> None:
    """``Pipeline.vc`` / ``Pipeline.pipeline`` / ``RVC.infer`` -> the device-resident versions.  The modules are imported here
    (after the faiss name has been taken care of) when the checkout provides them; a checkout without them is left alone."""
    from . import pipeline as _pl
    from . import realtime as _rt

    def grab(modname):
        mod = sys.modules.get(modname)
        if mod is None:
            try:
                mod = importlib.import_module(modname)
            except Exception:  # noqa  (a trimmed checkout, or one whose optional dependencies are missing)
                return None
        return mod

    mod = grab(_PIPELINE_MODULE)
    cls = getattr(mod, "Pipeline", None) if mod is not None else None
    if cls is not None:
        for name, new in (("vc", _pl.vc_hip), ("pipeline", _pl.pipeline_hip)):
            old = cls.__dict__.get(name)
            if old is not None and old is not new:
                new._rvcmi_original = old
                setattr(cls, name, new)
                rebound.append((cls, name, old))
        if "convert_files" not in cls.__dict__:  # a NEW method: Pipeline.pipeline for several inputs at once (vc_multi's loop, batched)
            cls.convert_files = _pl.convert_files
            rebound.append((cls, "convert_files", _ABSENT))
    mod = grab(_RTRVC_MODULE)
    cls = getattr(mod, "RVC", None) if mod is not None else None
    if cls is not None:
        old = cls.__dict__.get("infer")
        if old is not None and old is not _rt.rvc_infer_hip:
            _rt.rvc_infer_hip._rvcmi_original = old
            cls.infer = _rt.rvc_infer_hip
            rebound.append((cls, "infer", old))


def install(operand: str = "fp16", front: bool = True, device=None, patch_faiss: bool = True, patch_pipeline: bool = True) -> None:
    """Route the reference's loader, index reader and conversion methods through the HIP path (idempotent)."""
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
    # TorchScript export (rvc/synthesizer.py:38-64, used by rtrvc.py set_jit_model) scripts the REFERENCE modules: it resolves
    # ``load_synthesizer`` at call time, so it gets the original loader back for the duration of the call.
    ref_export = getattr(rs, "synthesizer_jit_export", None)
    if ref_export is not None:
        def synthesizer_jit_export(*a, **k):
            cur = rs.load_synthesizer
            rs.load_synthesizer = ref_load
            try:
                return ref_export(*a, **k)
            finally:
                rs.load_synthesizer = cur

        synthesizer_jit_export._rvcmi_original = ref_export
        for mod in [rs] + [sys.modules[m] for m in _LOADER_USERS if m in sys.modules]:
            if getattr(mod, "synthesizer_jit_export", None) is ref_export:
                mod.synthesizer_jit_export = synthesizer_jit_export
                rebound.append((mod, "synthesizer_jit_export", ref_export))
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
    if patch_pipeline:
        _rebind_methods(rebound)


def uninstall() -> None:
    if not _state.get("installed"):
        return
    for mod, name, old in reversed(_state.get("rebound", [])):
        if old is _ABSENT:
            delattr(mod, name)
        else:
            setattr(mod, name, old)
    if "faiss_prev" in _state:
        if _state["faiss_prev"] is None:
            sys.modules.pop("faiss", None)
        else:
            sys.modules["faiss"] = _state["faiss_prev"]
    _state.clear()
