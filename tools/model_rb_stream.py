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


HEAD2 = 52     # k_rb_stream2: X and H share the rows of M; new rows at HEAD2, histories in front
SLACK2 = 3


def run_strip2(x, W1, B1, W2, B2, k, dils, S0, S1, R, out, swizzle=True):
    """k_rb_stream2 (csrc/rb_stream2_kernels.hpp): same strip walk, but H is published over X in place, no tail is copied
    (a publish writes its last rows twice: tile + history buffer; the history is restored in front of the new rows by the
    same wave, reads before its own dual writes), rows are 256 B with the 16-byte chunk c of the row of time t stored at
    chunk c ^ (t & 15).  `swizzle` models that storage permutation on a 16-chunk row (C = 16 * 8 channels in the kernel;
    here a chunk = C // 16 channels) through the same address expressions as the kernel."""
    L, C = x.shape
    assert C % 16 == 0
    cw = C // 16
    nd = len(dils)
    p2 = (k - 1) // 2
    p1 = [d * (k - 1) // 2 for d in dils]
    Hx = [32 + p - p2 for p in p1]
    assert max(Hx) <= HEAD2 and R % 16 == 0
    HL = sum(p1) + nd * p2
    r0 = S0 - HL
    nsteps = -(-(S1 - r0 + 32 * nd) // R)
    MROWS = HEAD2 + R + SLACK2
    M = np.zeros((MROWS, C), np.float32)          # STORAGE order (swizzled chunks)
    sideX = [np.zeros((Hx[m], C), np.float32) for m in range(nd)]
    sideH = [np.zeros((2 * p2, C), np.float32) for m in range(nd)]
    carry = [np.zeros((32, C), np.float32) for m in range(nd)]

    def store_rows(buf, row0, vals, t0):
        """publish: logical rows `vals` whose first row has time t0 -> storage rows row0.. of buf"""
        for i in range(vals.shape[0]):
            key = (t0 + i) & 15 if swizzle else 0
            for c in range(16):
                pos = c ^ key
                buf[row0 + i, pos * cw:(pos + 1) * cw] = vals[i, c * cw:(c + 1) * cw]

    def load_rows(buf, row0, n, t0):
        """what the K loop reads: storage rows -> logical rows (reader knows the time of each row)"""
        o = np.empty((n, C), np.float32)
        for i in range(n):
            key = (t0 + i) & 15 if swizzle else 0
            for c in range(16):
                pos = c ^ key
                o[i, c * cw:(c + 1) * cw] = buf[row0 + i, pos * cw:(pos + 1) * cw]
        return o

    for i in range(nsteps):
        rows = r0 + i * R + np.arange(R)
        xin = np.where(((rows >= 0) & (rows < L))[:, None], x[np.clip(rows, 0, L - 1)], 0).astype(np.float32)
        for m in range(nd):
            wm = r0 - 32 * m + i * R
            assert (wm - r0) % 16 == 0
            # phase A: restore reads (raw rows, no re-swizzle), publish, restore writes, dual write of the tail
            hist = sideX[m].copy()
            rows = wm + np.arange(R)
            xp = np.where(((rows >= 0) & (rows < L))[:, None], lrelu(xin), 0).astype(np.float32)
            store_rows(M, HEAD2, xp, r0)                       # kernel key: (r0 + lrow) & 15  (wm == r0 mod 16)
            M[HEAD2 - Hx[m]:HEAD2] = hist
            store_rows(sideX[m], 0, xp[R - Hx[m]:], r0 + R - Hx[m])
            res = np.concatenate([carry[m], xin[:R - 32]])
            carry[m] = xin[R - 32:].copy()
            # conv1: the lane whose output row is q reads storage row HEAD2 - Hx + q + j*dil, time key (r0 - Hx + q + j*dil)
            Xl = load_rows(M, HEAD2 - Hx[m], Hx[m] + R, r0 - Hx[m])
            h = np.tile(B1[m][None, :], (R, 1)).astype(np.float32)
            for j in range(k):
                h += Xl[j * dils[m]: j * dils[m] + R] @ W1[m][:, :, j].T
            # phase B: H published over X in place
            hist = sideH[m].copy()
            am = wm - 32 + p2
            rows = am + np.arange(R)
            hp = np.where(((rows >= 0) & (rows < L))[:, None], lrelu(h), 0).astype(np.float32)
            store_rows(M, HEAD2, hp, r0 + p2)                  # kernel key: (r0 + p2 + lrow) & 15
            M[HEAD2 - 2 * p2:HEAD2] = hist
            store_rows(sideH[m], 0, hp[R - 2 * p2:], r0 + p2 + R - 2 * p2)
            Hl = load_rows(M, HEAD2 - 2 * p2, 2 * p2 + R, r0 - p2)  # kernel key at tap 0: (r0 - p2 + lrow) & 15
            acc = (res + B2[m][None, :]).astype(np.float32)
            for j in range(k):
                acc += Hl[j: j + R] @ W2[m][:, :, j].T
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
    C = 16
    for k in (3, 7, 11):
        for dils in ([1, 3, 5], [5], [3, 5]):
            for (L, R, strips) in ((1000, 96, 1), (1000, 96, 3), (777, 96, 2), (300, 96, 1), (90, 96, 2), (1001, 96, 7)):
                nd = len(dils)
                W1 = [rng.standard_normal((C, C, k), dtype=np.float32) / np.float32(np.sqrt(C * k)) for _ in range(nd)]
                W2 = [rng.standard_normal((C, C, k), dtype=np.float32) / np.float32(np.sqrt(C * k)) for _ in range(nd)]
                B1 = [(rng.standard_normal(C, dtype=np.float32) * np.float32(0.1)).astype(np.float32) for _ in range(nd)]
                B2 = [(rng.standard_normal(C, dtype=np.float32) * np.float32(0.1)).astype(np.float32) for _ in range(nd)]
                x = rng.standard_normal((L, C), dtype=np.float32)
                out = np.full((L, C), np.nan, np.float32)
                sl = -(-L // strips)
                for s in range(strips):
                    run_strip2(x, W1, B1, W2, B2, k, dils, s * sl, min(L, (s + 1) * sl), R, out)
                ref = reference(x, W1, B1, W2, B2, k, dils)
                err = np.abs(out - ref).max()
                assert np.isfinite(out).all() and err < 2e-4, ("v2", k, dils, L, R, strips, err)
    print("k_rb_stream2 schedule (shared X/H rows, dual-written histories, time-keyed swizzle) == direct ResBlock1 on every case")


if __name__ == "__main__":
    main()
    main3()


# ---------------------------------------------------------------------------------------------------------------------------
# k_rb_stream3 (csrc/rb_stream3_kernels.hpp): one block per CU, R = 192 rows per step handled as two HALVES a / b of 96 rows;
# every K loop (a "slot") carries the publish / history work of the OTHER half as fillers in its MFMA shadow:
#   slot 1  C1(m, a)   fillers  PX(m, b) (+ X tail -> sideX_m),  RH(m)
#   slot 2  C1(m, b)   fillers  PH(m, a),  RX(next pair)
#   slot 3  C2(m, a)   fillers  PH(m, b) (+ H tail -> sideH_m)
#   slot 4  C2(m, b)   fillers  PX(next pair, a)
# X and H live in SEPARATE buffers (a filler writes the buffer the running K loop does not read), rows are 256 B with the
# time-keyed chunk swizzle of k_rb_stream2.  The model runs the slots in order and checks, for every slot, that its fillers do
# not touch a row its K loop reads (the K loop is evaluated before AND after the fillers).
XHEAD3, HHEAD3, XSLACK3 = 52, 10, 3


def run_strip3(x, W1, B1, W2, B2, k, dils, S0, S1, out, swizzle=True):
    R, HALF = 192, 96
    L, C = x.shape
    assert C % 16 == 0
    cw = C // 16
    nd = len(dils)
    p2 = (k - 1) // 2
    p1 = [d * (k - 1) // 2 for d in dils]
    Hx = [32 + p - p2 for p in p1]
    assert max(Hx) <= XHEAD3 and 2 * p2 <= HHEAD3 and max(p1) + p2 <= 32
    HL = sum(p1) + nd * p2
    r0 = S0 - HL
    nsteps = -(-(S1 - r0 + 32 * nd) // R)
    XB = np.zeros((XHEAD3 + R + XSLACK3, C), np.float32)   # storage order
    HB = np.zeros((HHEAD3 + R + 1, C), np.float32)
    sideX = [np.zeros((Hx[m], C), np.float32) for m in range(nd)]
    sideH = [np.zeros((2 * p2, C), np.float32) for m in range(nd)]
    carry = [np.zeros((32, C), np.float32) for m in range(nd)]

    def store_rows(buf, row0, vals, t0):
        for i in range(vals.shape[0]):
            key = (t0 + i) & 15 if swizzle else 0
            for c in range(16):
                pos = c ^ key
                buf[row0 + i, pos * cw:(pos + 1) * cw] = vals[i, c * cw:(c + 1) * cw]

    def load_rows(buf, row0, n, t0):
        o = np.empty((n, C), np.float32)
        for i in range(n):
            key = (t0 + i) & 15 if swizzle else 0
            for c in range(16):
                pos = c ^ key
                o[i, c * cw:(c + 1) * cw] = buf[row0 + i, pos * cw:(pos + 1) * cw]
        return o

    def load_x(step):
        rows = r0 + step * R + np.arange(R)
        return np.where(((rows >= 0) & (rows < L))[:, None], x[np.clip(rows, 0, L - 1)], 0).astype(np.float32)

    def wm_of(m, step):
        return r0 - 32 * m + step * R

    def PX(m, step, half, xv):
        """publish the X (input of pair m at `step`) rows of one half from registers xv [R, C] (fp32 stream)"""
        wm = wm_of(m, step)
        lo = half * HALF
        rows = wm + lo + np.arange(HALF)
        xp = np.where(((rows >= 0) & (rows < L))[:, None], lrelu(xv[lo:lo + HALF]), 0).astype(np.float32)
        store_rows(XB, XHEAD3 + lo, xp, wm + lo)
        if half == 1:  # dual write: new rows [R - Hx, R) are the next step's X history (Hx <= 52: tiles 4 and 5 only)
            store_rows(sideX[m], 0, xp[HALF - Hx[m]:], wm + R - Hx[m])

    def RX(m):
        XB[XHEAD3 - Hx[m]:XHEAD3] = sideX[m]       # raw rows: the time of history row i is wm - Hx + i in both places

    def PH(m, step, half, hv):
        am = wm_of(m, step) - 32 + p2
        lo = half * HALF
        rows = am + lo + np.arange(HALF)
        hp = np.where(((rows >= 0) & (rows < L))[:, None], lrelu(hv[lo:lo + HALF]), 0).astype(np.float32)
        store_rows(HB, HHEAD3 + lo, hp, am + lo)
        if half == 1:
            store_rows(sideH[m], 0, hp[HALF - 2 * p2:], am + R - 2 * p2)

    def RH(m):
        HB[HHEAD3 - 2 * p2:HHEAD3] = sideH[m]

    def C1(m, step, half):
        wm = wm_of(m, step)
        lo = half * HALF
        n = HALF + 2 * p1[m]
        Xl = load_rows(XB, XHEAD3 - Hx[m] + lo, n, wm - Hx[m] + lo)
        h = np.tile(B1[m][None, :], (HALF, 1)).astype(np.float32)
        for j in range(k):
            h += Xl[j * dils[m]: j * dils[m] + HALF] @ W1[m][:, :, j].T
        return h

    def C2(m, step, half, res):
        am = wm_of(m, step) - 32 + p2
        lo = half * HALF
        Hl = load_rows(HB, HHEAD3 - 2 * p2 + lo, HALF + 2 * p2, am - 2 * p2 + lo)
        acc = (res[lo:lo + HALF] + B2[m][None, :]).astype(np.float32)
        for j in range(k):
            acc += Hl[j: j + HALF] @ W2[m][:, :, j].T
        return acc

    def slot(kfun, fillers):
        before = kfun()
        for f in fillers:
            f()
        after = kfun()
        assert np.array_equal(before, after), "a filler wrote a row the running K loop reads"
        return before

    xin = load_x(0)
    PX(0, 0, 0, xin)           # prologue (exposed): first half of the first step; histories start as zeros
    for step in range(nsteps):
        xin_next = None
        for m in range(nd):
            res = np.concatenate([carry[m], xin[:R - 32]])
            carry[m] = xin[R - 32:].copy()
            h = np.empty((R, C), np.float32)
            h[:HALF] = slot(lambda: C1(m, step, 0), [lambda: PX(m, step, 1, xin), lambda: RH(m)])
            mn = (m + 1) % nd
            h[HALF:] = slot(lambda: C1(m, step, 1), [lambda: PH(m, step, 0, h), lambda: RX(mn)])
            if m == nd - 1:
                xin_next = load_x(step + 1)
            xo = np.empty((R, C), np.float32)
            xo[:HALF] = slot(lambda: C2(m, step, 0, res), [lambda: PH(m, step, 1, h)])
            if m + 1 < nd:
                xo[HALF:] = slot(lambda: C2(m, step, 1, res), [lambda: PX(m + 1, step, 0, xo)])
            else:
                xo[HALF:] = slot(lambda: C2(m, step, 1, res), [lambda: PX(0, step + 1, 0, xin_next)])
            xin = xo
        wout = r0 - 32 * nd + step * R
        rows = wout + np.arange(R)
        ok = (rows >= S0) & (rows < S1)
        out[rows[ok]] = xin[ok]
        xin = xin_next


def main3():
    rng = np.random.default_rng(1)
    C = 16
    for k in (3, 7, 11):
        for dils in ([1, 3, 5], [5], [3, 5]):
            for (L, strips) in ((1000, 1), (1000, 3), (777, 2), (300, 1), (90, 2), (1001, 5), (2500, 2)):
                nd = len(dils)
                W1 = [rng.standard_normal((C, C, k), dtype=np.float32) / np.float32(np.sqrt(C * k)) for _ in range(nd)]
                W2 = [rng.standard_normal((C, C, k), dtype=np.float32) / np.float32(np.sqrt(C * k)) for _ in range(nd)]
                B1 = [(rng.standard_normal(C, dtype=np.float32) * np.float32(0.1)).astype(np.float32) for _ in range(nd)]
                B2 = [(rng.standard_normal(C, dtype=np.float32) * np.float32(0.1)).astype(np.float32) for _ in range(nd)]
                x = rng.standard_normal((L, C), dtype=np.float32)
                out = np.full((L, C), np.nan, np.float32)
                sl = -(-L // strips)
                for s in range(strips):
                    run_strip3(x, W1, B1, W2, B2, k, dils, s * sl, min(L, (s + 1) * sl), out)
                ref = reference(x, W1, B1, W2, B2, k, dils)
                err = np.abs(out - ref).max()
                assert np.isfinite(out).all() and err < 2e-4, ("v3", k, dils, L, strips, err)
    print("k_rb_stream3 schedule (half-step slots with fillers, separate X / H buffers) == direct ResBlock1 on every case")
