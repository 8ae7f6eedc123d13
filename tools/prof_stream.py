"""Dev: per-kernel HIP-event times of the realtime chunk's hot path (v1/40k generator at T=31 + 16-query retrieval)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rvc_amd
from oracle import nsf_oracle, synth

dev = torch.device("cuda", 0)
cfg = nsf_oracle.CONFIGS["v1_40k"]
w = synth.make_dec_weights(cfg, 1234)
T, NQ = 31, 16
z, f0, g = synth.make_dec_inputs(cfg, 1, T)
noise = nsf_oracle.reference_noise(1, T, cfg.upp)
idx = synth.make_ivf(10000, 768, seed=4321, kmeans_iters=1)
index = rvc_amd.IVFFlatHIP.from_arrays(idx["centroids"], idx["list_offsets"], idx["ids"], idx["vecs"], device=dev).reserve(NQ)
gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=dev, operand="fp16", max_B=1, max_T=64)
zd, fd, gd, nd = z.to(dev), f0.to(dev), g.to(dev), noise.to(dev)
feats = synth.make_phone(1, NQ, 768)[0].to(dev).contiguous()
fb = torch.empty_like(feats)
def hot():
    fb.copy_(feats)
    index.search_blend(fb, 0.75, 8, skip_if_short=True)
    return gen(zd, fd, gd, noise=nd)
for _ in range(5): hot()
gen.profile(True); index.profile(True)
for _ in range(20): hot()
torch.cuda.synchronize()
st = gen.profile_read() + index.profile_read()
tot = 0
for s in st:
    us = 1e3 * s["ms"] / 20
    tot += us
    print("%-18s %3d launches/chunk  %7.1f us/chunk  %6.1f us/launch" % (s["name"], s["launches"] // 20, us, 1e3 * s["ms"] / s["launches"]))
print("sum %.1f us" % tot)
