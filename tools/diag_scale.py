"""GPU diagnostic (round 6): where does the HIP generator stop being scale-invariant?  The network rescaled homogeneously by a power of
two G (tests/test_gpu_front.py:_scale_homogeneously) has a bit-identical fp32 reference; tap by tap, HIP(scaled) / G against HIP(unscaled),
and HIP(unscaled) twice (run-to-run determinism).  python tools/diag_scale.py [T]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rvc_amd
from oracle import nsf_oracle, synth
from test_gpu_front import _scale_homogeneously

T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
G = 512.0
cfg = nsf_oracle.CONFIGS["v2_48k"]
wd = synth.make_dec_weights(cfg, 1234)
ws = _scale_homogeneously(wd, G)
z, f0, g = synth.make_dec_inputs(cfg, 1, T)
noise = nsf_oracle.reference_noise(1, T, cfg.upp)
dev = torch.device("cuda:0")
a = (z.to(dev), f0.to(dev), g.to(dev))
rms = lambda x, y: float((x.double() - y.double()).pow(2).mean().sqrt())
for op in ("fp16", "fp32"):
    g0 = rvc_amd.NSFGeneratorHIP(vars(cfg), wd, device=dev, operand=op, max_B=1, max_T=T)
    g1 = rvc_amd.NSFGeneratorHIP(vars(cfg), ws, device=dev, operand=op, max_B=1, max_T=T)
    o0, o0b, o1 = g0(*a, noise=noise.to(dev)).cpu(), g0(*a, noise=noise.to(dev)).cpu(), g1(*a, noise=noise.to(dev)).cpu()
    print("== %s: final: unscaled twice bit-equal %s; scaled vs unscaled rms %.3e (waveform rms %.3f)" % (op, torch.equal(o0, o0b), rms(o0, o1), float(o0.pow(2).mean().sqrt())))
    for tap in ["pre"] + [x for i in range(4) for x in ("up%d" % i, "stage%d" % i)]:
        t0 = g0.debug_tap(tap, *a, noise=noise.to(dev))
        t1 = g1.debug_tap(tap, *a, noise=noise.to(dev)) / G
        print("   tap %-7s rms %.3f  scaled/G vs unscaled: rms %.3e  max %.3e  bit-equal %s" % (tap, float(t0.pow(2).mean().sqrt()), rms(t0, t1), float((t0 - t1).abs().max()), torch.equal(t0, t1)))
