"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the retrieval half of the hot path.

PARITY UNPINNED.  The arithmetic the reference uses lives in ``faiss-cpu`` (requirements/cpu.txt:8,
version not pinned, not vendored, not installable offline); the reference has no test, fixture
or golden vector at this boundary.  This file restates the *published*
IndexIVFFlat algorithm anchored on the reference's call sites:

* ``index.search(npy, k=8)``           infer/modules/vc/pipeline.py:126, infer/lib/rtrvc.py:172
* ``index.reconstruct_n(0, ntotal)``   infer/modules/vc/pipeline.py:215
* index recipe (IVF{nlist},Flat, L2, nprobe=1, sequential ids)   web.py:544-563
* weights / gather / blend              infer/modules/vc/pipeline.py:129-138

Numerics are *defined* here (and mirrored by the HIP path) as: squared L2 in exact arithmetic on
the fp32 inputs, evaluated in fp64 as sum((q-v)^2); coarse = the ``nprobe`` nearest centroids;
results sorted ascending by (distance, id); lists shorter than k padded with id -1 and distance
FLT_MAX (faiss' heap sentinel for METRIC_L2).  faiss itself evaluates these in fp32 with either a
BLAS ``|x|^2+|y|^2-2xy`` expansion or direct differences depending on batch size, so on exact
near-ties "the" faiss answer is not unique either.

The on-disk layout read/written below follows faiss ``impl/index_write.cpp`` / ``index_read.cpp``
as recalled (IwFl / IxF2 / ilar); it could not be validated against a real ``.index`` file offline.
"""
from __future__ import annotations

import struct
from typing import Tuple

import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)


def coarse_assign(index: dict, q: np.ndarray, nprobe: int) -> np.ndarray:
    """The ``nprobe`` nearest centroids per query, ascending (distance, id).  fp64 direct differences."""
    c = index["centroids"].astype(np.float64)
    out = np.empty((q.shape[0], nprobe), dtype=np.int64)
    for i in range(q.shape[0]):
        diff = c - q[i].astype(np.float64)[None, :]
        dist = np.einsum("ij,ij->i", diff, diff)
        order = np.lexsort((np.arange(dist.shape[0]), dist))
        out[i] = order[:nprobe]
    return out


def search(index: dict, q: np.ndarray, k: int = 8, nprobe: int | None = None) -> Tuple[np.ndarray, np.ndarray]:
    """IndexIVFFlat.search restatement -> (D [nq,k] float32 squared-L2 ascending, I [nq,k] int64)."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    nprobe = int(index.get("nprobe", 1) if nprobe is None else nprobe)
    nprobe = min(nprobe, index["nlist"])
    nq = q.shape[0]
    D = np.full((nq, k), FLT_MAX, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    if nq == 0:
        return D, I
    lists = coarse_assign(index, q, nprobe)
    off, ids, vecs = index["list_offsets"], index["ids"], index["vecs"]
    for i in range(nq):
        cand_d, cand_i = [], []
        for l in lists[i]:
            a, b = int(off[l]), int(off[l + 1])
            if b > a:
                diff = vecs[a:b].astype(np.float64) - q[i].astype(np.float64)[None, :]
                cand_d.append(np.einsum("ij,ij->i", diff, diff))
                cand_i.append(ids[a:b])
        if not cand_d:
            continue
        cd = np.concatenate(cand_d)
        ci = np.concatenate(cand_i)
        order = np.lexsort((ci, cd))[:k]
        D[i, :order.shape[0]] = cd[order].astype(np.float32)
        I[i, :order.shape[0]] = ci[order]
    return D, I


def _c_lib():
    """oracle/libivf_oracle.so (oracle/ivf_scan.c: the plain-C second restatement; OpenMP over queries), built on demand."""
    import ctypes as C
    import os
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    so, src = os.path.join(here, "libivf_oracle.so"), os.path.join(here, "ivf_scan.c")
    if not os.path.exists(so):
        subprocess.run(["gcc", "-O3", "-mavx2", "-mfma", "-fopenmp", "-shared", "-fPIC", "-o", so, src, "-lm"], check=True)
    return C.CDLL(so)


def search_c(index: dict, q: np.ndarray, k: int = 8, nprobe: int | None = None, f32: bool = False):
    """``search`` through the C restatement (fp64 definition unless ``f32``): the checker for query counts the numpy loops above
    would take minutes on (BASELINE configs[2] / [3]: 38 336 queries per call).  -> (D, I, P): P = list-major row positions."""
    import ctypes as C

    lib = _c_lib()
    q = np.ascontiguousarray(q, dtype=np.float32)
    nprobe = int(index.get("nprobe", 1) if nprobe is None else nprobe)
    nq, d = q.shape
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    P = np.empty((nq, k), np.int64)
    arrs = [np.ascontiguousarray(index["centroids"], np.float32), np.ascontiguousarray(index["list_offsets"], np.int64),
            np.ascontiguousarray(index["ids"], np.int64), np.ascontiguousarray(index["vecs"], np.float32)]
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.ivf_search(vp(q), C.c_int64(nq), C.c_int(d), vp(arrs[0]), C.c_int64(index["nlist"]), C.c_int(nprobe), vp(arrs[1]), vp(arrs[2]),
                   vp(arrs[3]), C.c_int(k), vp(D), vp(I), vp(P), C.c_int(1 if f32 else 0))
    return D, I, P


def blend_c(index: dict, feats: np.ndarray, D: np.ndarray, P: np.ndarray, index_rate: float) -> np.ndarray:
    """pipeline.py:129-138 through the C restatement (numpy's fp32 operation order) -> the blended copy of ``feats``."""
    import ctypes as C

    lib = _c_lib()
    out = np.ascontiguousarray(feats, dtype=np.float32).copy()
    nq, d = out.shape
    vecs = np.ascontiguousarray(index["vecs"], np.float32)
    pos_last = int(np.nonzero(index["ids"] == index["ntotal"] - 1)[0][0])  # numpy's big_npy[-1] for an id of -1
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.ivf_blend(vp(out), C.c_int64(nq), C.c_int(d), vp(np.ascontiguousarray(D)), vp(np.ascontiguousarray(P)), C.c_int(D.shape[1]), vp(vecs),
                  C.c_int64(pos_last), C.c_float(index_rate), C.c_float(1.0 - index_rate))
    return out


def reconstruct_n(index: dict, i0: int = 0, n: int | None = None) -> np.ndarray:
    """``index.reconstruct_n(0, ntotal)``: rows back in id order (pipeline.py:215)."""
    n = index["ntotal"] - i0 if n is None else n
    out = np.empty((index["ntotal"], index["d"]), dtype=np.float32)
    out[index["ids"]] = index["vecs"]
    return out[i0:i0 + n]


def blend(feats: np.ndarray, score: np.ndarray, ix: np.ndarray, big_npy: np.ndarray, index_rate: float) -> np.ndarray:
    """pipeline.py:129-138 in the same numpy operations and dtype (fp32):

        weight = np.square(1 / score); weight /= weight.sum(axis=1, keepdims=True)
        npy = np.sum(big_npy[ix] * np.expand_dims(weight, axis=2), axis=1)
        feats = npy * index_rate + (1 - index_rate) * feats

    Edge semantics kept as-is: id -1 indexes ``big_npy[-1]`` with weight (1/FLT_MAX)^2 = 0; an
    exact hit (score 0) gives inf/inf = NaN weights for that row.
    The torch expression at :135-138 evaluates ``npy * index_rate`` and ``(1 - index_rate) * feats``
    in fp32 with a python-float scalar, then adds.
    """
    with np.errstate(divide="ignore", invalid="ignore", over="ignore", under="ignore"):
        weight = np.square(1 / score)
        weight /= weight.sum(axis=1, keepdims=True)
        npy = np.sum(big_npy[ix] * np.expand_dims(weight, axis=2), axis=1)
        r = np.float32(index_rate)
        omr = np.float32(1 - index_rate)
        return (npy.astype(np.float32) * r + omr * feats.astype(np.float32)).astype(np.float32)


def search_blend(index: dict, feats: np.ndarray, index_rate: float, k: int = 8) -> np.ndarray:
    big_npy = reconstruct_n(index)
    score, ix = search(index, feats, k)
    return blend(feats, score, ix, big_npy, index_rate)


# ----------------------------------------------------------------------------------------------
# faiss on-disk format (IndexIVFFlat, METRIC_L2) -- independent python reader/writer used to
# cross-check the C++ reader in the product (csrc/ivf_io.cpp).
# ----------------------------------------------------------------------------------------------

def _index_header(d: int, ntotal: int, metric: int = 1) -> bytes:
    # write_index_header: int d; idx_t ntotal; idx_t dummy x2 (1<<20); bool is_trained; int metric_type
    return struct.pack("<iqqqBi", d, ntotal, 1 << 20, 1 << 20, 1, metric)


def write_index(index: dict, path: str, sparse: bool | None = None) -> None:
    d, n, nlist = index["d"], index["ntotal"], index["nlist"]
    off = index["list_offsets"]
    sizes = (off[1:] - off[:-1]).astype(np.uint64)
    with open(path, "wb") as f:
        f.write(b"IwFl")
        f.write(_index_header(d, n))
        f.write(struct.pack("<QQ", nlist, int(index.get("nprobe", 1))))
        # quantizer: IndexFlatL2 -> "IxF2", header, then the raw codes as a float vector
        f.write(b"IxF2")
        f.write(_index_header(d, nlist))
        f.write(struct.pack("<Q", nlist * d))
        f.write(np.ascontiguousarray(index["centroids"], dtype="<f4").tobytes())
        # direct map: type NoMap (0) + empty int64 vector
        f.write(struct.pack("<bQ", 0, 0))
        # ArrayInvertedLists
        f.write(b"ilar")
        f.write(struct.pack("<QQ", nlist, 4 * d))
        nonzero = int((sizes > 0).sum())
        if sparse is None:
            sparse = not (nonzero > nlist // 2)
        if not sparse:
            f.write(b"full")
            f.write(struct.pack("<Q", nlist))
            f.write(sizes.astype("<u8").tobytes())
        else:
            f.write(b"sprs")
            pairs = np.stack([np.nonzero(sizes)[0].astype(np.uint64), sizes[sizes > 0]], axis=1)
            f.write(struct.pack("<Q", pairs.size))
            f.write(pairs.astype("<u8").tobytes())
        for l in range(nlist):
            a, b = int(off[l]), int(off[l + 1])
            if b > a:
                f.write(np.ascontiguousarray(index["vecs"][a:b], dtype="<f4").tobytes())
                f.write(np.ascontiguousarray(index["ids"][a:b], dtype="<i8").tobytes())


def read_index(path: str) -> dict:
    buf = open(path, "rb").read()
    p = 0

    def take(fmt):
        nonlocal p
        v = struct.unpack_from(fmt, buf, p)
        p += struct.calcsize(fmt)
        return v

    def header():
        d, ntotal, _d1, _d2, trained, metric = take("<iqqqBi")
        if metric > 1:
            take("<f")
        return d, ntotal, metric

    assert buf[p:p + 4] == b"IwFl", buf[:4]
    p += 4
    d, ntotal, metric = header()
    assert metric == 1, "only METRIC_L2 indices are produced by web.py:547"
    nlist, nprobe = take("<QQ")
    assert buf[p:p + 4] in (b"IxF2", b"IxFl"), buf[p:p + 4]
    p += 4
    qd, qn, _ = header()
    (nfl,) = take("<Q")
    cent = np.frombuffer(buf, dtype="<f4", count=nfl, offset=p).reshape(qn, qd).copy()
    p += 4 * nfl
    (dm_type,) = take("<b")
    (dm_n,) = take("<Q")
    p += 8 * dm_n
    assert buf[p:p + 4] == b"ilar", buf[p:p + 4]
    p += 4
    nl2, code_size = take("<QQ")
    assert nl2 == nlist and code_size == 4 * d
    tag = buf[p:p + 4]
    p += 4
    sizes = np.zeros(nlist, dtype=np.int64)
    (cnt,) = take("<Q")
    if tag == b"full":
        sizes[:] = np.frombuffer(buf, dtype="<u8", count=cnt, offset=p)
        p += 8 * cnt
    else:
        assert tag == b"sprs", tag
        pairs = np.frombuffer(buf, dtype="<u8", count=cnt, offset=p).reshape(-1, 2)
        p += 8 * cnt
        sizes[pairs[:, 0].astype(np.int64)] = pairs[:, 1]
    off = np.zeros(nlist + 1, dtype=np.int64)
    np.cumsum(sizes, out=off[1:])
    vecs = np.empty((int(off[-1]), d), dtype=np.float32)
    ids = np.empty(int(off[-1]), dtype=np.int64)
    for l in range(nlist):
        n = int(sizes[l])
        if n:
            vecs[off[l]:off[l + 1]] = np.frombuffer(buf, dtype="<f4", count=n * d, offset=p).reshape(n, d)
            p += 4 * n * d
            ids[off[l]:off[l + 1]] = np.frombuffer(buf, dtype="<i8", count=n, offset=p)
            p += 8 * n
    assert p == len(buf), (p, len(buf))
    return dict(d=d, ntotal=ntotal, nlist=nlist, nprobe=nprobe, centroids=cent, list_offsets=off, ids=ids, vecs=vecs)
