"""Dev tool: front (enc_p + z_p + flow^-1) errors vs the golden fixtures and per-kernel times at clip size."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rvc_amd
from conftest import load_golden, rms
from oracle import synth
from oracle.front_oracle import FrontConfig

gpu = torch.device("cuda:0")
for name in ("front_v2_B2_T50", "front_v2_B1_T100_head6"):
    d = load_golden(name)
    fcfg = FrontConfig()
    wf = synth.make_front_weights(fcfg, int(d["seed"]))
    for op in ("fp16", "bf16"):
        fr = rvc_amd.FrontHIP(vars(fcfg), wf, device=gpu, operand=op, max_B=2, max_T=128)
        t = lambda k: torch.from_numpy(d[k]).to(gpu)
        fh = max(int(d["flow_head"]), 0)
        args = (t("phone"), t("pitch"), t("lengths"), t("g"))
        z = fr(*args, fh, noise=t("noise")).cpu()
        line = "%s %s: z rms err %.2e" % (name, op, rms(z, d["z"]))
        for tap in ("emb", "attn0", "layer0", "layer5", "z_p"):
            line += "  %s %.2e" % (tap, rms(fr.debug_tap(tap, *args, fh, noise=t("noise")), d[tap]))
        print(line)

B = int(os.environ.get("B", "1")); T = 1198
fcfg = FrontConfig(); wf = synth.make_front_weights(fcfg, 1)
fr = rvc_amd.FrontHIP(vars(fcfg), wf, device=gpu, operand="fp16", max_B=B, max_T=T)
phone = synth.make_phone(1, T, 768, 1).repeat(B, 1, 1).to(gpu); pitch = synth.make_pitch(synth.make_f0(B, T)).to(gpu)
g = wf["emb_g.weight"][:B].to(gpu); nz = torch.randn(B, 192, T, device=gpu)
for _ in range(3): fr(phone, pitch, None, g, 0, noise=nz)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): fr(phone, pitch, None, g, 0, noise=nz)
torch.cuda.synchronize(); print("B=%d T=%d eager: %.3f ms per call" % (B, T, (time.perf_counter() - t0) / 20 * 1e3))
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr): zz = fr(phone, pitch, None, g, 0, noise=nz)
gr.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): gr.replay()
torch.cuda.synchronize(); print("B=%d T=%d hipGraph: %.3f ms per call" % (B, T, (time.perf_counter() - t0) / 50 * 1e3))
fr.profile(True)
for _ in range(5): fr(phone, pitch, None, g, 0, noise=nz)
torch.cuda.synchronize()
tot = 0
for s in fr.profile_read():
    tot += s["ms"] / 5
    print("  %-14s %3d launches  %.4f ms/call  %7.1f TFLOP/s" % (s["name"], s["launches"] // 5, s["ms"] / 5, s["flops"] / max(s["ms"], 1e-9) / 1e9))
print("  sum %.3f ms" % tot)
