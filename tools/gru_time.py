import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import conftest  # noqa
import rvc_amd
dev = torch.device("cuda:0")
torch.manual_seed(0)
ref = torch.nn.GRU(384, 256, batch_first=True, bidirectional=True).eval()
m = rvc_amd.GRUHIP(ref, device=dev)
tg = ref.to(dev).half()
for B, T in ((1, 32), (1, 64), (1, 1216), (8, 1216), (64, 1216)):  # (32 / 64 frames: the f0 window of a realtime chunk, rtrvc.py:203-207)
    x = torch.randn(B, T, 384, device=dev).half()
    for name, f in (("hip", lambda: m(x)), ("torch/MIOpen fp16", lambda: tg(x))):
        with torch.no_grad():
            for _ in range(2): f()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 20
            for _ in range(n): f()
            torch.cuda.synchronize()
        print("B=%d T=%d %s: %.3f ms" % (B, T, name, 1e3 * (time.perf_counter() - t0) / n), flush=True)
