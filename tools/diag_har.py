"""dev: where does the HIP `har` tap differ from the oracle on a voiced full-size clip?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rvc_amd
from oracle import nsf_oracle, synth

cfg = nsf_oracle.CONFIGS["v2_48k"]
w = synth.make_dec_weights(cfg, 1234)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1198
z, f0, g = synth.make_dec_inputs(cfg, 1, T, 1234)
noise = nsf_oracle.reference_noise(1, T, cfg.upp, 114514)
dev = torch.device("cuda:0")
gen = rvc_amd.NSFGeneratorHIP(vars(cfg), w, device=dev, operand="fp16", max_B=1, max_T=T)
har = gen.debug_tap("har", z.to(dev), f0.to(dev), g.to(dev), noise=noise.to(dev))[0, 0].numpy().astype(np.float64)
upp, sr = cfg.upp, cfg.sr
f = f0[0].numpy()
w32 = (np.fmod((f[:-1] / np.float32(sr) * np.float32(upp)).astype(np.float32) + np.float32(0.5), np.float32(1)) - np.float32(0.5)).astype(np.float32)
pref = np.concatenate([[0.0], np.cumsum(w32.astype(np.float64))]).astype(np.float32)
phase = np.fmod(pref, np.float32(1.0)).astype(np.float32)
n = np.arange(1, upp + 1, dtype=np.float32)
base = ((f / np.float32(sr)).astype(np.float32)[:, None] * n[None, :]).astype(np.float32)
rad = (base + phase[:, None]).astype(np.float32)
arg = (np.float32(6.2831855) * rad).astype(np.float32)
uv = (f > 0).astype(np.float32)[:, None]
amp = np.where(uv > 0, np.float32(0.003), np.float32(0.1) / np.float32(3.0)).astype(np.float32)
nz = noise[0].numpy().reshape(T, upp)
lw, lb = np.float32(2.5), np.float32(0.1)
def finish(sinv):
    s = (sinv.astype(np.float32) * np.float32(0.1)).astype(np.float32)
    v = (s * uv + (amp * nz).astype(np.float32)).astype(np.float32)
    return np.tanh((v * lw).astype(np.float32) + lb).astype(np.float64)
variants = {
    "sin32(arg32)": finish(np.sin(arg)),
    "sin64(arg32)": finish(np.sin(arg.astype(np.float64))),
    "sin64(2pi64*rad32)": finish(np.sin(2 * np.pi * rad.astype(np.float64))),
    "oracle": nsf_oracle.har_source(w, f0, upp, sr, noise)[0, 0].numpy().reshape(T, upp).astype(np.float64),
}
H = har.reshape(T, upp)
ref = variants["oracle"]
print("rms(har) %.4f" % np.sqrt((ref ** 2).mean()))
for k, v in variants.items():
    e = H - v
    print("%-22s rms %.3e  voiced rms %.3e  unvoiced rms %.3e  max %.3e" % (k, np.sqrt((e ** 2).mean()), np.sqrt((e[f > 0] ** 2).mean()),
          np.sqrt((e[f == 0] ** 2).mean()) if (f == 0).any() else 0, np.abs(e).max()))
e = np.abs(H - ref).max(axis=1)
bad = np.nonzero(e > 1e-6)[0]
print("frames with |err| > 1e-6: %d of %d; first %s" % (len(bad), T, bad[:20]))
for t in bad[:6]:
    j = int(np.argmax(np.abs(H[t] - ref[t])))
    print(" frame %d f0 %.2f phase %.7f prefix %.6f  worst n=%d arg %.5f err %.3e  implied dphase %.3e" % (
        t, f[t], phase[t], pref[t], j + 1, arg[t, j], H[t, j] - ref[t, j], (H[t, j] - ref[t, j]) / (0.25 * 2 * np.pi * max(1e-9, abs(np.cos(arg[t, j]))))))
