"""Evidence for csrc/exact_fp.hpp:div3_exact -- the three-instruction quotient  q = x*y; r = fma(-3, q, x); q' = fma(r, y, q)  with
y = RN32(1/3) equals the IEEE fp32 quotient x / 3 for EVERY mantissa (scaling by 2^e is exact, so one exponent covers all x whose
quotient is a normal number; several are run anyway).  The fma's are evaluated exactly: products of two fp32 are exact in fp64,
the residual is exactly representable in fp32 (asserted), and the final sum is re-done in rational arithmetic wherever the fp64
addition's rounding error could move it across an fp32 rounding boundary.   python tools/check_div3.py [exponents...]"""
import fractions
import sys

import numpy as np


def mismatches(exp: int, sign: float) -> int:
    y = np.float32(1.0) / np.float32(3.0)
    A = np.arange(2 ** 23, 2 ** 24, dtype=np.int64)
    a64 = sign * A.astype(np.float64) * 2.0 ** (exp - 23)
    a32 = a64.astype(np.float32)
    assert np.array_equal(a32.astype(np.float64), a64)
    ref = a32 / np.float32(3.0)                                        # IEEE, correctly rounded
    q0 = (a64 * np.float64(y)).astype(np.float32)                      # RN32(x * y)
    r64 = a64 - 3.0 * q0.astype(np.float64)                            # exact
    r = r64.astype(np.float32)
    assert np.array_equal(r.astype(np.float64), r64), "the residual must be exact in fp32"
    p = r64 * np.float64(y)                                            # exact (24 x 24 bits)
    q064 = q0.astype(np.float64)
    s = q064 + p
    bb = s - q064
    err = (q064 - (s - bb)) + (p - bb)                                 # TwoSum: exact rounding error of the fp64 addition
    q1 = s.astype(np.float32)
    up = np.nextafter(q1, np.float32(np.inf)).astype(np.float64)
    dn = np.nextafter(q1, np.float32(-np.inf)).astype(np.float64)
    sus = (np.abs(s - (q1 + up) / 2) <= np.abs(err) * 4) | (np.abs(s - (q1 + dn) / 2) <= np.abs(err) * 4)
    for i in np.nonzero(sus)[0]:
        ex = fractions.Fraction(float(q0[i])) + fractions.Fraction(float(r[i])) * fractions.Fraction(float(y))
        c = np.float32(float(ex))
        cands = (np.nextafter(c, np.float32(-np.inf)), c, np.nextafter(c, np.float32(np.inf)))
        q1[i] = min(cands, key=lambda v: (abs(fractions.Fraction(float(v)) - ex), int(np.float32(v).view(np.uint32)) & 1))
    return int((q1 != ref).sum())


if __name__ == "__main__":
    exps = [int(v) for v in sys.argv[1:]] or [0, 1, -1, 5, -20, 60, -100]
    total = 0
    for e in exps:
        for sg in (1.0, -1.0):
            n = mismatches(e, sg)
            total += n
            print("exponent %4d sign %+d: 2^23 mantissas, %d mismatches against x / 3" % (e, int(sg), n))
    print("total mismatches:", total)
    sys.exit(1 if total else 0)
