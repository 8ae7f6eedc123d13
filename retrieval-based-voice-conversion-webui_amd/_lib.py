"""ctypes binding of ``include/rvcmi.h`` (librvcmi.so).

There is deliberately NO fallback: if the shared library has not been built, or a call fails,
this module raises.  A silent CPU/PyTorch fallback would void every parity and performance claim.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from typing import List

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RVCMI_LIB") or os.path.join(_HERE, "librvcmi.so")  # RVCMI_LIB: dev A/B builds
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["nsf.hip", "rb_stream.hip", "ivf.hip", "front.hip", "glue.hip", "gru.hip"]

RVCMI_MAX_UPS, RVCMI_MAX_RB, RVCMI_MAX_DIL = 8, 4, 4
RVCMI_VERSION = 2  # include/rvcmi.h; the argument lists of SYMBOLS below are those of this ABI version
OPERANDS = {"fp32": 0, "f32": 0, "bf16": 1, "fp16": 2, "f16": 2}


class NsfConfig(C.Structure):
    _fields_ = [
        ("inter_channels", C.c_int),
        ("upsample_initial_channel", C.c_int),
        ("gin_channels", C.c_int),
        ("sr", C.c_int),
        ("use_f0", C.c_int),
        ("n_ups", C.c_int),
        ("upsample_rates", C.c_int * RVCMI_MAX_UPS),
        ("upsample_kernel_sizes", C.c_int * RVCMI_MAX_UPS),
        ("n_resblock_kernels", C.c_int),
        ("resblock_kernel_sizes", C.c_int * RVCMI_MAX_RB),
        ("n_dilations", C.c_int * RVCMI_MAX_RB),
        ("resblock_dilation_sizes", (C.c_int * RVCMI_MAX_DIL) * RVCMI_MAX_RB),
        ("operand", C.c_int),
    ]


class FrontConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "in_channels", "inter_channels", "hidden_channels", "filter_channels", "n_heads", "n_layers", "kernel_size",
        "window_size", "gin_channels", "use_f0", "flow_n_flows", "flow_n_layers", "flow_kernel_size", "flow_dilation_rate",
        "operand")]


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int), ("shape", C.c_int64 * 4)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


# every symbol include/rvcmi.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("rvcmi_last_error", C.c_char_p, []),
    ("rvcmi_version", C.c_int, []),
    ("rvcmi_nsf_create", C.c_int, [C.POINTER(NsfConfig), C.POINTER(Tensor), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    ("rvcmi_nsf_destroy", C.c_int, [_P]),
    ("rvcmi_nsf_forward", C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_int, _P, _P]),
    ("rvcmi_nsf_upp", C.c_int, [_P]),
    ("rvcmi_nsf_workspace_bytes", C.c_size_t, [_P]),
    ("rvcmi_nsf_debug_forward", C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, C.c_char_p, _P, C.c_size_t,
                                          C.POINTER(C.c_int64), _P]),
    ("rvcmi_nsf_set_option", C.c_int, [_P, C.c_char_p, C.c_double]),
    ("rvcmi_nsf_profile_enable", C.c_int, [_P, C.c_int]),
    ("rvcmi_nsf_profile_read", C.c_int, [_P, C.POINTER(KernelStat), C.c_int, C.POINTER(C.c_int), C.c_int]),
    ("rvcmi_front_create", C.c_int, [C.POINTER(FrontConfig), C.POINTER(Tensor), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    ("rvcmi_front_destroy", C.c_int, [_P]),
    ("rvcmi_front_forward", C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_int, _P, _P]),
    ("rvcmi_front_workspace_bytes", C.c_size_t, [_P]),
    ("rvcmi_front_debug_forward", C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_char_p, _P, C.c_size_t,
                                            C.POINTER(C.c_int64), _P]),
    ("rvcmi_front_set_option", C.c_int, [_P, C.c_char_p, C.c_double]),
    ("rvcmi_front_profile_enable", C.c_int, [_P, C.c_int]),
    ("rvcmi_front_profile_read", C.c_int, [_P, C.POINTER(KernelStat), C.c_int, C.POINTER(C.c_int), C.c_int]),
    ("rvcmi_ivf_create_from_file", C.c_int, [C.c_char_p, C.c_int, C.POINTER(_P)]),
    ("rvcmi_ivf_build", C.c_int, [C.c_int, C.c_int64, _P, C.c_int64, C.c_int, C.c_uint64, C.c_int, _P, C.POINTER(_P)]),
    ("rvcmi_kmeans", C.c_int, [C.c_int, C.c_int64, _P, C.c_int64, C.c_int, C.c_uint64, C.c_int, _P, _P]),
    ("rvcmi_ivf_write_file", C.c_int, [_P, C.c_char_p]),
    ("rvcmi_ivf_create", C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int, _P, _P, _P, _P, C.c_int, C.POINTER(_P)]),
    ("rvcmi_ivf_destroy", C.c_int, [_P]),
    ("rvcmi_ivf_d", C.c_int, [_P]),
    ("rvcmi_ivf_ntotal", C.c_int64, [_P]),
    ("rvcmi_ivf_nlist", C.c_int64, [_P]),
    ("rvcmi_ivf_nprobe", C.c_int, [_P]),
    ("rvcmi_ivf_set_nprobe", C.c_int, [_P, C.c_int]),
    ("rvcmi_ivf_reserve", C.c_int, [_P, C.c_int64]),
    ("rvcmi_ivf_search", C.c_int, [_P, C.c_int64, _P, C.c_int, _P, _P, _P]),
    ("rvcmi_ivf_search_blend", C.c_int, [_P, C.c_int64, _P, C.c_float, C.c_int, C.c_int, _P]),
    ("rvcmi_ivf_search_blend_expand", C.c_int, [_P, C.c_int64, _P, C.c_float, C.c_int, C.c_int, _P, C.c_float, C.c_int64, _P, _P]),
    ("rvcmi_ivf_reconstruct_n", C.c_int, [_P, C.c_int64, C.c_int64, _P]),
    ("rvcmi_ivf_blob", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    ("rvcmi_ivf_centroids", C.c_int, [_P, _P]),
    ("rvcmi_ivf_blob_copy", C.c_int, [_P, _P, C.c_size_t, _P]),
    ("rvcmi_ivf_create_from_blob", C.c_int, [_P, C.c_size_t, C.c_int, C.c_int, C.POINTER(_P)]),
    ("rvcmi_ivf_set_option", C.c_int, [_P, C.c_char_p, C.c_double]),
    ("rvcmi_ivf_profile_enable", C.c_int, [_P, C.c_int]),
    ("rvcmi_ivf_profile_read", C.c_int, [_P, C.POINTER(KernelStat), C.c_int, C.POINTER(C.c_int), C.c_int]),
    ("rvcmi_glue_expand_protect", C.c_int, [_P, C.c_int64, C.c_int, C.c_int, _P, C.c_float, C.c_int64, _P, _P]),
    ("rvcmi_glue_rmvpe_f0", C.c_int, [_P, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _P, _P, _P, _P]),
    ("rvcmi_glue_f0_post", C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P]),
    ("rvcmi_glue_change_rms", C.c_int, [_P, C.c_int64, C.c_int, _P, C.c_int64, C.c_int, C.c_float, _P, _P]),
    ("rvcmi_glue_scale_int16_range", C.c_int, [_P, C.c_int64, _P, _P]),
    ("rvcmi_glue_resample_poly", C.c_int, [_P, C.c_int64, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int64, _P]),
    ("rvcmi_glue_sola", C.c_int, [_P, C.c_int64, _P, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, _P]),
    ("rvcmi_gru_create", C.c_int, [C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, C.POINTER(_P)]),
    ("rvcmi_gru_destroy", C.c_int, [_P]),
    ("rvcmi_gru_forward", C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P]),
]


class RvcmiError(RuntimeError):
    """``code``: the C ABI's return value (include/rvcmi.h: -1 invalid, -2 HIP, -3 IO / not an IVF-Flat L2 file, -4 no memory,
    -5 missing weight); None for errors raised on the Python side."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


ERR_IO = -3


_lib = None


def _unit_deps(src: str) -> List[str]:
    """`src` and every local header it includes, transitively (`#include "x.hpp"` next to it, `rvcmi.h` under include/)."""
    import re

    seen, todo = [], [src]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(f):
            continue
        seen.append(f)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(f, errors="replace").read(), re.M):
            for base in (os.path.dirname(f), os.path.join(_HERE, "..", "include")):
                cand = os.path.normpath(os.path.join(base, inc))
                if os.path.exists(cand):
                    todo.append(cand)
                    break
    return seen


def build(verbose: bool = True, force: bool = False) -> str:
    """Compile librvcmi.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps.append(os.path.join(_HERE, "..", "include", "rvcmi.h"))
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    flags_key = " ".join([os.environ.get(k, "") for k in ("RVCMI_DEFINES", "RVCMI_DEV_STAMPS", "RVCMI_NO_MFMA_VGPR_FORM", "HIPCC")])
    stamp = os.path.join(CSRC, ".build_flags")
    same_flags = os.path.exists(stamp) and open(stamp).read() == flags_key
    for s in srcs:  # compile the translation units in parallel; a unit whose object is newer than it and every header it includes is kept
        o = os.path.join(CSRC, os.path.basename(s) + ".o")
        objs.append(o)
        if not force and same_flags and os.path.exists(o) and all(os.path.getmtime(o) >= os.path.getmtime(d) for d in _unit_deps(s)):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o]
        if not os.environ.get("RVCMI_NO_MFMA_VGPR_FORM"):
            # MFMA results in VGPRs wherever the 256 architectural VGPRs allow it: by default LLVM puts every accumulator in an
            # AGPR and copies it out with v_accvgpr_read for each VALU use (publish, bias, residual) -- 21 000 such copies in
            # nsf.hip alone, none with this option; the 512-register kernels keep their overflow in AGPRs
            cmd[1:1] = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
        for d_ in os.environ.get("RVCMI_DEFINES", "").split():
            cmd.insert(1, "-D" + d_)  # dev: geometry experiments, e.g. RVCMI_DEFINES="RBF32_NWV=8"
        if os.environ.get("RVCMI_DEV_STAMPS"):
            cmd.insert(1, "-DRVCMI_DEV_STAMPS")  # per-phase s_memtime stamps in the fused kernels (RVCMI_DBG=32)
        if verbose:
            print("[rvcmi build]", " ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RvcmiError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
    with open(stamp, "w") as f:
        f.write(flags_key)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print("[rvcmi build]", " ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode:
        raise RvcmiError("link failed:\n" + r.stdout.decode(errors="replace"))
    return LIB_PATH


def lib() -> C.CDLL:
    """The loaded library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RvcmiError(
                "librvcmi.so is not built (%s).  Run `python __graft_entry__.py build`.  "
                "There is no CPU fallback for the HIP hot path." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(l, name)  # AttributeError if the header and the library drift apart
            f.restype = res
            f.argtypes = args
        v = l.rvcmi_version()
        if v != RVCMI_VERSION:  # a stale .so against a newer header shifts pointer arguments: garbage reads, not an error
            raise RvcmiError("%s implements ABI version %d, this binding is written against %d (include/rvcmi.h); rebuild with "
                             "`python __graft_entry__.py build`" % (LIB_PATH, v, RVCMI_VERSION))
        _lib = l
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().rvcmi_last_error()
        raise RvcmiError("rvcmi error %d: %s" % (rc, msg.decode(errors="replace") if msg else "?"), code=int(rc))


def device_index(dev) -> int:
    """The ordinal of a CUDA (ROCm) device; ``cuda`` without an index means the current device."""
    import torch

    dev = torch.device(dev)
    return dev.index if dev.index is not None else torch.cuda.current_device()


def set_option(fn, handle, key: str, value) -> None:
    """Dev / test option of one handle (include/rvcmi.h rvcmi_*_set_option); ``value=None`` restores the default."""
    check(fn(handle, key.encode(), float("nan") if value is None else float(value)))


def read_stats(read_fn, handle, reset: bool = True) -> List[dict]:
    n = C.c_int(0)
    buf = (KernelStat * 64)()
    check(read_fn(handle, buf, 64, C.byref(n), 1 if reset else 0))
    return [dict(name=buf[i].name.decode(), launches=int(buf[i].launches), ms=float(buf[i].ms),
                 flops=float(buf[i].flops), bytes=float(buf[i].bytes)) for i in range(min(n.value, 64))]
