"""dev: per-phase cycle stamps of k_rb_full on the benchmark clip (needs a library built with RVCMI_DEV_STAMPS=1; RVCMI_LIB selects it):
RVCMI_LIB=.../librvcmi_stamps.so python tools/stamp_rbf.py
Prints, per kernel size, the mean cycles between consecutive stamps of a tile (nsf_kernels.hpp k_rb_full: 1 x loaded, 2 X published,
3 barrier, 4 conv1, 5 H published, 6 barrier, 7 conv2, 8 X' published, 9 barrier (pair 0), 10 all pairs, 11 stores issued)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import rvc_amd
from oracle import nsf_oracle, synth

dev = torch.device("cuda", 0)
cfg = nsf_oracle.CONFIGS["v2_48k"]
w = synth.make_dec_weights(cfg, 1234)
z, f0, g = synth.make_dec_inputs(cfg, 1, 1198)
noise = nsf_oracle.reference_noise(1, 1198, cfg.upp)
gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=dev, operand="fp16", max_B=1, max_T=1198)
zd, fd, gd, nd = z.to(dev), f0.to(dev), g.to(dev), noise.to(dev)
for _ in range(3):
    gen(zd, fd, gd, noise=nd)
torch.cuda.synchronize()
gen.set_option("DBG", 32)
gen(zd, fd, gd, noise=nd)
torch.cuda.synchronize()
