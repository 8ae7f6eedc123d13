"""TEST INFRASTRUCTURE ONLY -- quantifies the UNPINNED retrieval risk: how often would faiss-cpu's own
fp32 arithmetic (oracle/ivf_faisslike.py, five plausible build variants) return something else than the exact fp64 answer
that ``ivf_oracle.search`` defines and the HIP path reproduces bit for bit?

    python -m oracle.faisslike_report [--big] > profiles/r03_faisslike_flips.json

Cases: the BASELINE index (10000 x 768 i.i.d. rows, 599 queries = bench.py's workload), the clustered 10000 x 768 index of
bench.py's side leg, the real-HuBERT fixture (mute.npy rows x 40 jittered copies + exact duplicates, queries = the rows
themselves and near hits), a small batch (nq = 16 < 20: the direct coarse path, realtime chunks), and with --big the
200000 x 256 / 1M x 256 stress indices (minutes of numpy).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ivf_faisslike as fl  # noqa: E402
from oracle import ivf_oracle, synth  # noqa: E402


def run_case(name: str, idx: dict, q: np.ndarray, k: int = 8) -> dict:
    t0 = time.time()
    De, Ie = ivf_oracle.search(idx, q, k)
    Le = ivf_oracle.coarse_assign(idx, q, 1)[:, 0]
    sizes = np.diff(idx["list_offsets"])
    out = {"index": "%d x %d, nlist %d" % (idx["ntotal"], idx["d"], idx["nlist"]), "queries": int(q.shape[0]),
           "rows_scanned_per_query": float(sizes[Le].mean()), "variants": {}}
    # how close are the exact answers themselves to a tie?  (relative gap between the best and the second-best distance)
    ok = (De[:, 1] < fl.FLT_MAX) & (De[:, 0] > 0)
    gap = (De[ok, 1].astype(np.float64) - De[ok, 0]) / De[ok, 0]
    out["exact_top1_rel_gap"] = {"min": float(gap.min()) if gap.size else None, "p01": float(np.quantile(gap, 0.01)) if gap.size else None,
                                 "median": float(np.median(gap)) if gap.size else None,
                                 "exact_ties_at_top1": int(((De[:, 0] == De[:, 1]) & (De[:, 1] < fl.FLT_MAX)).sum())}
    for v in fl.VARIANTS:
        out["variants"][v] = fl.compare((De, Ie), Le, fl.search_faisslike(idx, q, k, v))
    out["seconds"] = round(time.time() - t0, 1)
    print("[faisslike] %s done in %.1fs" % (name, out["seconds"]), file=sys.stderr)
    return out


def cases(big: bool):
    idx = synth.make_ivf(10000, 768, seed=4321, kmeans_iters=1)
    q = synth.make_phone(1, 599, 768)[0].numpy()
    yield "baseline_10000x768_iid_599q", idx, q
    yield "baseline_10000x768_iid_16q_direct_coarse", idx, q[:16]
    xr, cen = synth.make_clustered_rows(10000 + 599, 768, synth.ivf_nlist(10000), return_centres=True)
    yield "clustered_10000x768_599q", synth.make_ivf_from_rows(xr[:10000], kmeans_iters=2, init=cen), xr[10000:]
    gold = os.path.join(ROOT, "tests", "golden", "mute_hubert.npz")
    if os.path.exists(gold):
        g = np.load(gold)
        for dim in (768, 256):
            feats = g["f%d" % dim]
            x = synth.make_mute_rows(feats)
            qq = np.concatenate([feats, (feats[:60] + np.float32(1e-3)).astype(np.float32)])
            yield "mute_hubert_%d_real_rows_with_exact_duplicates" % dim, synth.make_ivf_from_rows(x), qq
    if big:
        for n in (200000, 1000000):
            nl = synth.ivf_nlist(n)
            xr, cen = synth.make_clustered_rows(n + 599, 256, nl, return_centres=True, seed=78)
            yield "stress_%dx256_clustered_599q" % n, synth.make_ivf_from_rows(xr[:n], nlist=nl, kmeans_iters=0, init=cen), xr[n:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    a = ap.parse_args()
    rep = {"what": "flips of faiss-like fp32 arithmetic (oracle/ivf_faisslike.py) against the exact fp64 answer = the HIP path's answer; "
                   "k = 8, nprobe = 1", "variants": {k: v for k, v in fl.VARIANTS.items()}, "cases": {}}
    for name, idx, q in cases(a.big):
        rep["cases"][name] = run_case(name, idx, q)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
