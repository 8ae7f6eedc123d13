"""Dev tool: index build time (GPU k-means + add), web.py:544-563 recipe sizes."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rvc_amd
for n, d in ((10000, 768), (200000, 768), (200000, 256)):
    x = np.random.default_rng(1).standard_normal((n, d), dtype=np.float32)
    t0 = time.perf_counter()
    idx, obj = rvc_amd.IVFFlatHIP.train(x, niter=10, device="cuda:0", return_objective=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("N=%d d=%d nlist=%d: build %.2f s (10 k-means iterations + add), objective %.4g -> %.4g" % (n, d, idx.nlist, dt, obj[0], obj[-1]))
