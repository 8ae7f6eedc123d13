"""GPU bring-up diagnostic for the IVF path."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rvc_amd
from oracle import ivf_oracle, synth

dev = torch.device("cuda:0")
for (n, d, nq) in [(2000, 256, 100), (10000, 768, 599)]:
    idx = synth.make_ivf(n, d, seed=4321, dup=5)
    h = rvc_amd.IVFFlatHIP.from_arrays(idx["centroids"], idx["list_offsets"], idx["ids"], idx["vecs"], device=dev)
    q = synth.make_phone(1, nq, d)[0].numpy()
    q[:3] = idx["xb"][:3]  # exact hits
    t0 = time.time(); D, I = h.search(q, 8); t1 = time.time()
    Dr, Ir = ivf_oracle.search(idx, q, 8)
    print("n=%d d=%d nq=%d nlist=%d: ids equal %s (%d mismatches), D max rel err %.2e, %.1f ms" % (
        n, d, nq, idx["nlist"], np.array_equal(I, Ir), int((I != Ir).sum()),
        float(np.max(np.abs(D - Dr) / np.maximum(np.abs(Dr), 1e-30))), 1e3 * (t1 - t0)))
    big = h.reconstruct_n(0, h.ntotal)
    print("  reconstruct equal:", np.array_equal(big, idx["xb"]))
    feats = torch.from_numpy(q[3:].copy()).to(dev)
    out = h.search_blend(feats.clone(), 0.75).cpu().numpy()
    exp = ivf_oracle.search_blend(idx, q[3:], 0.75)
    print("  blend max abs err %.3e rms %.3e" % (np.abs(out - exp).max(), np.sqrt(np.mean((out - exp) ** 2))))
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "a.index")
        rvc_amd.write_index(h, p)
        r = ivf_oracle.read_index(p)
        print("  file round trip (C++ writer -> py reader):", np.array_equal(r["vecs"], idx["vecs"]), np.array_equal(r["ids"], idx["ids"]))
        ivf_oracle.write_index(idx, p)
        h2 = rvc_amd.read_index(p, dev)
        D2, I2 = h2.search(q, 8)
        print("  file round trip (py writer -> C++ reader):", np.array_equal(I2, I))
    b = h.blob()
    h3 = rvc_amd.IVFFlatHIP.from_blob(b)
    D3, I3 = h3.search(q, 8)
    print("  blob round trip:", np.array_equal(I3, I), "blob MB", b.numel() / 1e6)
