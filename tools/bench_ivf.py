"""IVF micro-benchmark: per-kernel HIP-event times for a few index shapes (optionally with RVCMI_IVF_DBG ablations)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rvc_amd

dev = torch.device("cuda:0")
shapes = [(10000, 768, 256, 599), (10000, 256, 256, 599), (200000, 768, 5128, 599), (1000000, 256, 16000, 599)]
if len(sys.argv) > 1:
    shapes = shapes[: int(sys.argv[1])]
for n, d, nlist, nq in shapes:
    rng = np.random.default_rng(1)
    vecs = rng.standard_normal((n, d), dtype=np.float32)
    cent = vecs[rng.choice(n, nlist, replace=False)].copy()
    sizes = np.full(nlist, n // nlist, dtype=np.int64)
    sizes[: n - sizes.sum()] += 1
    off = np.zeros(nlist + 1, np.int64)
    np.cumsum(sizes, out=off[1:])
    ids = np.arange(n, dtype=np.int64)
    h = rvc_amd.IVFFlatHIP.from_arrays(cent, off, ids, vecs, device=dev)
    q = torch.from_numpy(rng.standard_normal((nq, d), dtype=np.float32)).to(dev)
    for _ in range(3):
        h.search(q, 8)
    h.profile(True)
    for _ in range(5):
        f = q.clone()
        h.search_blend(f, 0.75)
    torch.cuda.synchronize()
    st = h.profile_read()
    print("n=%d d=%d nlist=%d nq=%d dbg=%s:" % (n, d, nlist, nq, os.environ.get("RVCMI_IVF_DBG", "0")),
          " ".join("%s=%.1fus" % (s["name"], 1e3 * s["ms"] / s["launches"]) for s in st))
