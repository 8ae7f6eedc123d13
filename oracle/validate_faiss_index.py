"""TEST INFRASTRUCTURE -- pin the retrieval path to a REAL faiss artefact the day one is available.

faiss is not installable offline, so the IwFl/IxF2/ilar reader and the search semantics of this repo are restatements
("parity unpinned").  Given an index file written by faiss itself (``faiss.write_index``, e.g. any
``added_IVF*_Flat_nprobe_1_*.index`` of an RVC model, web.py:554-571) and optionally a dump of what faiss answered on some
queries, this tool reports exactly where the restatement and faiss disagree:

    python -m oracle.validate_faiss_index added.index                       # walk the file layout, field by field
    python -m oracle.validate_faiss_index added.index --dump qdi.npz        # + compare search results
                                                                            #   qdi.npz: q [nq,d] f32, D [nq,k] f32, I [nq,k] i64
    # how to make the dump on a machine that has faiss (6 lines):
    #   import faiss, numpy as np; ix = faiss.read_index("added.index"); big = ix.reconstruct_n(0, ix.ntotal)
    #   q = (big[np.random.default_rng(0).integers(0, ix.ntotal, 599)] + 0.05).astype("float32")
    #   D, I = ix.search(q, 8); np.savez("qdi.npz", q=q, D=D, I=I, big_head=big[:64])

Checks: (1) layout -- every fourcc / header field / vector length against what csrc/ivf.hip and oracle/ivf_oracle.py expect,
without stopping at the first surprise; (2) ``reconstruct_n`` rows vs ``big_head`` when present; (3) search -- the CPU oracle
(fp64 distances, ties -> lowest id) and, on a GPU box, the HIP index read by the product's own C++ reader, against faiss'
(D, I): identical top-1, identical top-k sequence, same top-k SET (order flips inside exact/near ties are benign: faiss
computes fp32 distances), genuine mismatches, max relative distance error.
"""
from __future__ import annotations

import argparse
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def walk_layout(path: str) -> list:
    """Field-by-field description of an IndexIVFFlat file; returns a list of problem strings (empty = as expected)."""
    buf = open(path, "rb").read()
    p, problems = 0, []

    def take(fmt, what):
        nonlocal p
        if p + struct.calcsize(fmt) > len(buf):
            problems.append("file ends inside %s (offset %d)" % (what, p))
            raise EOFError
        v = struct.unpack_from(fmt, buf, p)
        p += struct.calcsize(fmt)
        return v

    def fourcc(expect, what):
        nonlocal p
        cc = buf[p:p + 4]
        p += 4
        ok = cc in expect
        print("  %-28s %r%s" % (what, cc, "" if ok else "   <-- expected one of %s" % (expect,)))
        if not ok:
            problems.append("%s is %r, expected %s" % (what, cc, expect))
        return cc

    def header(what):
        d, ntotal, d1, d2, trained, metric = take("<iqqqBi", what)
        print("  %-28s d=%d ntotal=%d dummies=(%d,%d) is_trained=%d metric=%d" % (what, d, ntotal, d1, d2, trained, metric))
        if metric != 1:
            problems.append("%s: metric_type %d (only METRIC_L2 = 1 is produced by web.py:547)" % (what, metric))
            if metric > 1:
                take("<f", "metric_arg")
        return d, ntotal

    try:
        print("%s: %d bytes" % (path, len(buf)))
        fourcc((b"IwFl",), "index fourcc")
        d, ntotal = header("IndexIVF header")
        nlist, nprobe = take("<QQ", "nlist/nprobe")
        print("  %-28s nlist=%d nprobe=%d" % ("ivf", nlist, nprobe))
        fourcc((b"IxF2", b"IxFl"), "quantizer fourcc")
        qd, qn = header("quantizer header")
        if qd != d or qn != nlist:
            problems.append("quantizer is %d x %d, expected nlist x d = %d x %d" % (qn, qd, nlist, d))
        (nfl,) = take("<Q", "centroid vector length")
        print("  %-28s %d floats (%s)" % ("centroids", nfl, "ok" if nfl == nlist * d else "expected %d" % (nlist * d)))
        if nfl != nlist * d:
            problems.append("centroid vector holds %d floats, expected %d" % (nfl, nlist * d))
        p += 4 * nfl
        (dm_type,) = take("<b", "direct map type")
        (dm_n,) = take("<Q", "direct map length")
        print("  %-28s type=%d entries=%d" % ("direct map", dm_type, dm_n))
        if dm_type == 2:
            problems.append("direct map is a Hashtable (type 2): not supported by the readers")
        p += 8 * dm_n
        fourcc((b"ilar",), "inverted lists fourcc")
        nl2, code_size = take("<QQ", "ilar nlist/code_size")
        print("  %-28s nlist=%d code_size=%d (%s)" % ("array inverted lists", nl2, code_size, "ok" if code_size == 4 * d else "expected 4*d = %d" % (4 * d)))
        if nl2 != nlist or code_size != 4 * d:
            problems.append("ilar nlist/code_size = %d/%d, expected %d/%d" % (nl2, code_size, nlist, 4 * d))
        tag = fourcc((b"full", b"sprs"), "list sizes encoding")
        (cnt,) = take("<Q", "sizes vector length")
        sizes = np.zeros(int(nlist), np.int64)
        if tag == b"full":
            sizes[:] = np.frombuffer(buf, "<u8", int(cnt), p)
        else:
            pr = np.frombuffer(buf, "<u8", int(cnt), p).reshape(-1, 2)
            sizes[pr[:, 0].astype(np.int64)] = pr[:, 1]
        p += 8 * int(cnt)
        print("  %-28s sum=%d (ntotal %d) min=%d max=%d empty=%d" % ("list sizes", sizes.sum(), ntotal, sizes.min(), sizes.max(), int((sizes == 0).sum())))
        if sizes.sum() != ntotal:
            problems.append("list sizes sum to %d, ntotal is %d" % (sizes.sum(), ntotal))
        p += int(sizes.sum()) * (4 * d + 8)
        print("  %-28s %d (file has %d)%s" % ("expected end of file", p, len(buf), "" if p == len(buf) else "   <-- MISMATCH"))
        if p != len(buf):
            problems.append("payload ends at %d but the file has %d bytes" % (p, len(buf)))
    except EOFError:
        pass
    return problems


def compare(name, D, I, Dr, Ir):
    nq, k = Ir.shape
    top1 = float((I[:, 0] == Ir[:, 0]).mean())
    seq = float((I == Ir).all(1).mean())
    sets = np.array([set(I[i]) == set(Ir[i]) for i in range(nq)])
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.abs(D - Dr) / np.maximum(np.abs(Dr), 1e-30)
    rel = rel[np.isfinite(rel) & (Ir >= 0)]
    # a position where ids differ but both distances agree to fp32 rounding is a tie-order flip, not an error
    flips = (I != Ir) & (np.abs(D - Dr) <= 4e-6 * np.maximum(np.abs(Dr), 1e-30))
    genuine = int(((I != Ir) & ~flips).sum())
    print("  %-10s top-1 identical %.4f | top-%d sequence identical %.4f | same set %.4f | tie-order flips %d | genuine id mismatches %d | max rel D err %.2e"
          % (name, top1, k, seq, float(sets.mean()), int(flips.sum()), genuine, float(rel.max()) if rel.size else 0.0))
    return genuine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("index")
    ap.add_argument("--dump", help="npz with q, D, I from faiss (and optionally big_head)")
    a = ap.parse_args()
    problems = walk_layout(a.index)
    print("layout: %s" % ("as expected" if not problems else "%d problem(s):" % len(problems)))
    for pr in problems:
        print("   - " + pr)
    if problems:
        return 1
    from oracle import ivf_oracle

    idx = ivf_oracle.read_index(a.index)
    hip = None
    try:
        import torch

        if torch.cuda.is_available():
            import rvc_amd

            hip = rvc_amd.read_index(a.index)
            ok = np.array_equal(hip.reconstruct_n(0, hip.ntotal), ivf_oracle.reconstruct_n(idx))
            print("HIP reader (csrc/ivf.hip) vs python reader: reconstruct_n %s" % ("identical" if ok else "DIFFERS"))
    except Exception as e:  # noqa
        print("HIP index not available here (%s): CPU oracle only" % e)
    if not a.dump:
        return 0
    z = np.load(a.dump)
    q, Dr, Ir = np.ascontiguousarray(z["q"], np.float32), z["D"], z["I"].astype(np.int64)
    k = Ir.shape[1]
    if "big_head" in z:
        big = ivf_oracle.reconstruct_n(idx, 0, z["big_head"].shape[0])
        print("reconstruct_n head vs faiss: %s" % ("identical" if np.array_equal(big, z["big_head"]) else "DIFFERS (ids are not the add order?)"))
    bad = 0
    D, I = ivf_oracle.search(idx, q, k)
    bad += compare("oracle", D, I, Dr, Ir)
    if hip is not None and k <= 8:
        Dh, Ih = hip.search(q, k)
        bad += compare("HIP", Dh, Ih, Dr, Ir)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
