"""TEST INFRASTRUCTURE ONLY -- what faiss-cpu's OWN fp32 arithmetic would return, as a bracket.

PARITY UNPINNED (faiss is not installable offline; see ivf_oracle.py).  The HIP path and ``ivf_oracle.search`` return the
EXACT answer: squared L2 evaluated in fp64 on the fp32 inputs, ties by id.  faiss evaluates in fp32, and which fp32
expression it uses depends on the build and on the batch size, so "the" faiss answer on a near-tie is not unique.  This
module restates the published fp32 code paths of ``IndexIVFFlat.search`` (faiss 1.7.x, the versions ``faiss-cpu`` on PyPI
shipped while the reference pinned none, requirements/cpu.txt:8) so that the distance between "exact" and "what faiss would
return" can be MEASURED instead of assumed (``faisslike_report.py``):

* list scan (``IVFFlatScanner::scan_codes``): ``fvec_L2sqr(x, y, d)`` = direct differences accumulated in fp32 SIMD lanes,
  lanes added at the end.  Lane count and fused multiply-add depend on the build (hand-written AVX: 8 lanes, mul + add,
  ``distances_simd.cpp``; auto-vectorised loop since 1.7.3: 8..32 lanes, possibly FMA) -> ``lanes`` in {1, 8, 16, 32},
  ``fma`` in {False, True}.  ``lanes=1`` is the scalar reference loop of the same file.
* coarse quantizer (``IndexFlatL2.search`` -> ``knn_L2sqr``): ``nq < 20`` (``distance_compute_blas_threshold``) the same
  direct-difference kernel; ``nq >= 20`` the BLAS expansion ``|x|^2 + |y|^2 - 2 x.y`` with fp32 ``sgemm`` and fp32 norms,
  negative results clamped to 0 (``exhaustive_L2sqr_blas``).  The sgemm's own summation order is the BLAS library's; numpy's
  float32 matmul (OpenBLAS here) stands in for it.
* top-k: a max-heap that replaces its top only on a strictly smaller distance (``dis < simi[0]``), results sorted by
  distance; on equal distances faiss >= 1.7.3 breaks ties by id (``CMax::cmp2``).  Restated as a stable sort by
  (distance, id): the *set* of k results is what the comparison below scores, the tie order separately.

Nothing here is used by the product; ``tests/`` and ``faisslike_report.py`` import it.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)

VARIANTS: Dict[str, dict] = {
    "scalar": dict(lanes=1, fma=False),         # reference loop, one fp32 accumulator
    "avx8": dict(lanes=8, fma=False),           # hand-written AVX kernel (faiss <= 1.7.2 fvec_L2sqr)
    "avx8_fma": dict(lanes=8, fma=True),
    "avx16_fma": dict(lanes=16, fma=True),      # auto-vectorised, 2 x 8 lanes or one AVX-512 register
    "avx32_fma": dict(lanes=32, fma=True),      # auto-vectorised, 4 x 8 lanes interleaved
}


def _hsum(acc: np.ndarray) -> np.ndarray:
    """Add the lanes of ``acc`` [..., W] in fp32 the way the SIMD epilogues do: registers of 8 lanes pairwise
    ((r0 + r1) + (r2 + r3)), then the upper 128-bit half onto the lower, then two horizontal adds ((a0 + a1) + (a2 + a3))."""
    w = acc.shape[-1]
    a = acc.astype(np.float32)
    if w == 1:
        return a[..., 0]
    while w > 8:  # combine 8-lane registers pairwise
        regs = a.reshape(a.shape[:-1] + (w // 8, 8))
        regs = (regs[..., 0::2, :] + regs[..., 1::2, :]).astype(np.float32)
        a = regs.reshape(a.shape[:-1] + (w // 2,))
        w //= 2
    lo = (a[..., 0:4] + a[..., 4:8]).astype(np.float32)
    h1 = (lo[..., 0::2] + lo[..., 1::2]).astype(np.float32)
    return (h1[..., 0] + h1[..., 1]).astype(np.float32)


def l2sqr_fp32(x: np.ndarray, y: np.ndarray, lanes: int = 8, fma: bool = False) -> np.ndarray:
    """``fvec_L2sqr`` of every row of ``y`` [n, d] against ``x`` [d] in fp32 with ``lanes`` SIMD accumulators.
    d must be a multiple of ``lanes`` (768 and 256 are; faiss handles a remainder with masked lanes)."""
    x = np.asarray(x, dtype=np.float32)
    y = np.asarray(y, dtype=np.float32)
    n, d = y.shape
    assert d % lanes == 0
    diff = (y - x[None, :]).astype(np.float32).reshape(n, d // lanes, lanes)
    acc = np.zeros((n, lanes), dtype=np.float32)
    if fma:
        d64 = diff.astype(np.float64)
        for s in range(d // lanes):  # one rounding per step: fp32(acc + diff*diff), the product exact in fp64
            acc = (acc.astype(np.float64) + d64[:, s, :] * d64[:, s, :]).astype(np.float32)
    else:
        sq = (diff * diff).astype(np.float32)
        for s in range(d // lanes):
            acc = (acc + sq[:, s, :]).astype(np.float32)
    return _hsum(acc)


def norms_fp32(a: np.ndarray, lanes: int, fma: bool) -> np.ndarray:
    """``fvec_norms_L2sqr``: |a_i|^2 in fp32 with the same lane structure."""
    return l2sqr_fp32(np.zeros(a.shape[1], np.float32), a, lanes=lanes, fma=fma)


def coarse_fp32(cent: np.ndarray, q: np.ndarray, lanes: int, fma: bool, blas_threshold: int = 20) -> Tuple[np.ndarray, np.ndarray]:
    """Nearest centroid per query (nprobe = 1, web.py:552) as ``IndexFlatL2.search`` evaluates it -> (list id, fp32 distance).
    First minimum wins (a heap of size 1 is replaced only by a strictly smaller distance)."""
    nq = q.shape[0]
    out = np.empty(nq, dtype=np.int64)
    dmin = np.empty(nq, dtype=np.float32)
    if nq >= blas_threshold:
        xn = norms_fp32(q, lanes, fma)
        yn = norms_fp32(cent, lanes, fma)
        ip = (q.astype(np.float32) @ cent.astype(np.float32).T).astype(np.float32)
        dis = (xn[:, None] + yn[None, :]).astype(np.float32) - (np.float32(2) * ip).astype(np.float32)
        dis = np.maximum(dis.astype(np.float32), np.float32(0))
        out[:] = np.argmin(dis, axis=1)
        dmin[:] = dis[np.arange(nq), out]
    else:
        for i in range(nq):
            dis = l2sqr_fp32(q[i], cent, lanes, fma)
            out[i] = int(np.argmin(dis))
            dmin[i] = dis[out[i]]
    return out, dmin


def search_faisslike(index: dict, q: np.ndarray, k: int = 8, variant: str = "avx8", blas_threshold: int = 20) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """``IndexIVFFlat.search`` (nprobe = 1) in faiss' fp32 arithmetic, variant ``VARIANTS[variant]``.
    -> (D [nq, k] float32, I [nq, k] int64, probed list [nq])."""
    v = VARIANTS[variant]
    q = np.ascontiguousarray(q, dtype=np.float32)
    nq = q.shape[0]
    D = np.full((nq, k), FLT_MAX, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    if nq == 0:
        return D, I, np.zeros(0, np.int64)
    lists, _ = coarse_fp32(index["centroids"], q, v["lanes"], v["fma"], blas_threshold)
    off, ids, vecs = index["list_offsets"], index["ids"], index["vecs"]
    for i in range(nq):
        a, b = int(off[lists[i]]), int(off[lists[i] + 1])
        if b <= a:
            continue
        dis = l2sqr_fp32(q[i], vecs[a:b], v["lanes"], v["fma"])
        order = np.lexsort((ids[a:b], dis))[:k]
        D[i, : order.shape[0]] = dis[order]
        I[i, : order.shape[0]] = ids[a:b][order]
    return D, I, lists


def compare(exact: Tuple[np.ndarray, np.ndarray], exact_lists: np.ndarray, got: Tuple[np.ndarray, np.ndarray, np.ndarray]) -> dict:
    """Flip statistics of one fp32 variant against the exact (fp64) answer the HIP path reproduces."""
    De, Ie = exact
    Dg, Ig, Lg = got
    nq, k = Ie.shape
    top1 = int((Ie[:, 0] != Ig[:, 0]).sum())
    seq = int((Ie != Ig).any(axis=1).sum())
    sets = int(sum(set(Ie[i].tolist()) != set(Ig[i].tolist()) for i in range(nq)))
    coarse = int((exact_lists != Lg).sum())
    same_list = exact_lists == Lg
    top1_same_list = int(((Ie[:, 0] != Ig[:, 0]) & same_list).sum())
    sets_same_list = int(sum(set(Ie[i].tolist()) != set(Ig[i].tolist()) for i in range(nq) if same_list[i]))
    fin = (De < FLT_MAX) & (Dg < FLT_MAX)
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.abs(Dg.astype(np.float64) - De.astype(np.float64)) / np.maximum(De.astype(np.float64), 1e-30)
    rel = np.where(fin & same_list[:, None] & (De > 0), rel, 0.0)
    return {"queries": int(nq), "coarse_list_flips": coarse, "top1_flips": top1, "top1_flips_same_list": top1_same_list,
            "top%d_set_flips" % k: sets, "top%d_set_flips_same_list" % k: sets_same_list, "top%d_sequence_flips" % k: seq,
            "max_rel_dD_same_list": float(rel.max()) if rel.size else 0.0}
