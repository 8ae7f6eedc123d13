"""GPU bring-up diagnostic: per-tap error of the HIP generator against the oracle in every operand mode.
Usage (on a GPU box): python tools/diag_nsf.py [config] [T] [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rvc_amd
from oracle import nsf_oracle, synth

name = sys.argv[1] if len(sys.argv) > 1 else "v2_48k"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = nsf_oracle.CONFIGS[name]
w = synth.make_dec_weights(cfg, 1234)
z, f0, g = synth.make_dec_inputs(cfg, B, T)
noise = nsf_oracle.reference_noise(B, T, cfg.upp)
taps = {}
with torch.no_grad():
    ref = nsf_oracle.generator_forward(cfg, w, z, f0, g, noise, taps=taps)
dev = torch.device("cuda:0")
nk = cfg.num_kernels
for op in ("fp32", "bf16", "fp16"):
    gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=dev, operand=op, max_B=B, max_T=T)
    print("== operand", op, "workspace MB", gen.workspace_bytes / 1e6)
    for k, v in taps.items():
        try:
            got = gen.debug_tap(k, z.to(dev), f0.to(dev), g.to(dev), noise=noise.to(dev))
        except Exception as e:  # noqa
            print("  tap", k, "FAILED", e)
            continue
        exp = v * nk if k.startswith("stage") else v
        err = (got - exp).float()
        print("  tap %-7s shape %-18s rms_ref %.3f  rms_err %.3e  max_err %.3e" % (
            k, tuple(got.shape), exp.pow(2).mean().sqrt().item(), err.pow(2).mean().sqrt().item(), err.abs().max().item()))
    t0 = time.time()
    out = gen(z.to(dev), f0.to(dev), g.to(dev), noise=noise.to(dev))
    torch.cuda.synchronize()
    out = out.cpu()
    err = out - ref
    print("  FINAL rms_err %.3e max_err %.3e (ref rms %.3f) nan=%s  [%.1f ms]" % (
        err.pow(2).mean().sqrt().item(), err.abs().max().item(), ref.pow(2).mean().sqrt().item(),
        bool(torch.isnan(out).any()), 1e3 * (time.time() - t0)))
