"""CPU model of the streaming fused-ResBlock schedule of k_rb_stream (csrc/nsf_kernels.hpp): the SAME buffer layout,
row arithmetic, masks, history copies and residual tile shift, in numpy fp32 -- checked against a direct evaluation of
ResBlock1 (rvc/layers/residuals.py:68-85).  Run: python tools/model_rb_stream.py"""
import numpy as np
import torch
import torch.nn.functional as F

HEAD = 62      # first row of the new X rows in M
HROW = 10      # first row of the new H rows in M
SLACK = 8


def lrelu(x):
    return np.maximum(x, 0.1 * x)


def run_strip(x, W1, B1, W2, B2, k, dils, S0, S1, R, out):
    """x [L, C] fp32; produces out[S0:S1].  One block's work."""
    L, C = x.shape
    nd = len(dils)
    p2 = (k - 1) // 2
    p1 = [d * (k - 1) // 2 for d in dils]
    Hx = [32 + p - p2 for p in p1]
    HL = sum(p1) + nd * p2
    r0 = S0 - HL
    nsteps = -(-(S1 - r0 + 32 * nd) // R)
    M = np.zeros((HEAD + R + SLACK, C), np.float32)
    sideX = [np.zeros((Hx[m], C), np.float32) for m in range(nd)]
    sideH = [[np.zeros((2 * p2, C), np.float32) for _ in range(2)] for m in range(nd)]
    carry = [np.zeros((32, C), np.float32) for m in range(nd)]
    NJ = R // 32
    for i in range(nsteps):
        par = i & 1
        # global load (clamped + masked)
        rows = r0 + i * R + np.arange(R)
        xin = np.where(((rows >= 0) & (rows < L))[:, None], x[np.clip(rows, 0, L - 1)], 0).astype(np.float32)
        for m in range(nd):
            wm = r0 - 32 * m + i * R  # first row of xin
            # phase A
            M[HEAD - Hx[m]:HEAD] = sideX[m]
            rows = wm + np.arange(R)
            M[HEAD:HEAD + R] = np.where(((rows >= 0) & (rows < L))[:, None], lrelu(xin), 0)
            res = np.concatenate([carry[m], xin[:R - 32]])
            carry[m] = xin[R - 32:].copy()
            # conv1: h row q reads M[HEAD-Hx + q + j*dil]
            base = HEAD - Hx[m]
            h = np.tile(B1[m][None, :], (R, 1)).astype(np.float32)
            for j in range(k):
                h += M[base + j * dils[m]: base + j * dils[m] + R] @ W1[m][:, :, j].T
            # phase B
            sideX[m] = M[HEAD + R - Hx[m]:HEAD + R].copy()
            M[HROW - 2 * p2:HROW] = sideH[m][par]
            am = wm - 32 + p2
            rows = am + np.arange(R)
            hp = np.where(((rows >= 0) & (rows < L))[:, None], lrelu(h), 0).astype(np.float32)
            M[HROW:HROW + R] = hp
            sideH[m][par ^ 1] = hp[R - 2 * p2:].copy()
            # conv2: out row q reads M[HROW-2p2 + q + j]
            base = HROW - 2 * p2
            acc = (res + B2[m][None, :]).astype(np.float32)
            for j in range(k):
                acc += M[base + j: base + j + R] @ W2[m][:, :, j].T
            xin = acc
        wout = r0 - 32 * nd + i * R
        rows = wout + np.arange(R)
        ok = (rows >= S0) & (rows < S1)
        out[rows[ok]] = xin[ok]


def reference(x, W1, B1, W2, B2, k, dils):
    t = torch.from_numpy(x.T[None].copy())
    for m, d in enumerate(dils):
        xt = F.leaky_relu(t, 0.1)
        xt = F.conv1d(xt, torch.from_numpy(W1[m]), torch.from_numpy(B1[m]), dilation=d, padding=d * (k - 1) // 2)
        xt = F.leaky_relu(xt, 0.1)
        xt = F.conv1d(xt, torch.from_numpy(W2[m]), torch.from_numpy(B2[m]), padding=(k - 1) // 2)
        t = xt + t
    return t[0].numpy().T


def main():
    rng = np.random.default_rng(0)
    C = 8
    for k in (3, 7, 11):
        for dils in ([1, 3, 5], [1], [5], [3, 5]):
            for (L, R, strips) in ((1000, 128, 1), (1000, 128, 3), (777, 256, 2), (300, 256, 1), (90, 128, 2)):
                nd = len(dils)
                W1 = [rng.standard_normal((C, C, k), dtype=np.float32) / np.float32(np.sqrt(C * k)) for _ in range(nd)]
                W2 = [rng.standard_normal((C, C, k), dtype=np.float32) / np.float32(np.sqrt(C * k)) for _ in range(nd)]
                B1 = [(rng.standard_normal(C, dtype=np.float32) * np.float32(0.1)).astype(np.float32) for _ in range(nd)]
                B2 = [(rng.standard_normal(C, dtype=np.float32) * np.float32(0.1)).astype(np.float32) for _ in range(nd)]
                x = rng.standard_normal((L, C), dtype=np.float32)
                out = np.full((L, C), np.nan, np.float32)
                sl = -(-L // strips)
                for s in range(strips):
                    run_strip(x, W1, B1, W2, B2, k, dils, s * sl, min(L, (s + 1) * sl), R, out)
                ref = reference(x, W1, B1, W2, B2, k, dils)
                err = np.abs(out - ref).max()
                assert np.isfinite(out).all() and err < 2e-4, (k, dils, L, R, strips, err)
    print("streaming ResBlock schedule == direct ResBlock1 on every case")


if __name__ == "__main__":
    main()
