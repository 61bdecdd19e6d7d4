#!/usr/bin/env python3
"""Independent high-precision check of the modular-reduction polynomial (dev-time tool, not product, not oracle).

The reference fits cos(2 pi (x - 1/4) / 2^r) on the union of the intervals [k - 2^-loge, k + 2^-loge],
k = -(K-1) .. K-1, by a multi-interval Remez iteration in 1000-bit NTL arithmetic until the error levels at the
reference points agree to 2^-120 (include/source/bootstrapping/common/Remez.cpp:577-586, RemezCos.h:11-16): its
result is the minimax polynomial of that function on that set, which is unique (Chebyshev's equioscillation theorem).
This script computes the same minimax polynomial with its own exchange iteration in mpmath at 400 bits and writes the
Chebyshev coefficients (basis T_j(x / K), as the reference stores them) to tests/golden/; the C++ product code
(seal_shim/bootstrapping/moai_remez.h, __float128) is tested against that file and against the equioscillation
property itself.

    python tools/remez_mpmath.py [K deg loge r] > tests/golden/remez_cos_K25_deg59_loge10_r2.json
"""
import json
import sys

import mpmath as mp

mp.mp.prec = 400


def cheb_eval(c, t):
    b1 = mp.mpf(0)
    b2 = mp.mpf(0)
    for a in reversed(c[1:]):
        b1, b2 = 2 * t * b1 - b2 + a, b1
    return t * b1 - b2 + c[0]


def remez(f, K, deg, w, tol=mp.mpf(2) ** -100, grid=64, log=None):
    centres = list(range(-(K - 1), K))
    # start: the extrema of a least-squares fit on Chebyshev points of every interval
    pts = []
    for k in centres:
        for j in range(8):
            pts.append(k + w * mp.cos(mp.pi * (2 * j + 1) / 16))
    A = mp.matrix(len(pts), deg + 1)
    for i, x in enumerate(pts):
        t = x / K
        A[i, 0] = 1
        A[i, 1] = t
        for j in range(2, deg + 1):
            A[i, j] = 2 * t * A[i, j - 1] - A[i, j - 2]
    b = mp.matrix([f(x) for x in pts])
    c = list(mp.qr_solve(A, b)[0])
    E = mp.mpf(0)
    for it in range(60):
        err = lambda x: cheb_eval(c, x / K) - f(x)
        # candidates: per interval, endpoints and interior local extrema of the error
        cand = []
        for k in centres:
            xs = [k - w + 2 * w * i / grid for i in range(grid + 1)]
            es = [err(x) for x in xs]
            for i in range(grid + 1):
                left = es[i - 1] if i > 0 else None
                right = es[i + 1] if i < grid else None
                a = abs(es[i])
                if (left is None or a >= abs(left)) and (right is None or a >= abs(right)):
                    x = xs[i]
                    if left is not None and right is not None:
                        # refine by golden-section on |err| inside [xs[i-1], xs[i+1]]
                        lo, hi = xs[i - 1], xs[i + 1]
                        sgn = 1 if es[i] > 0 else -1
                        for _ in range(60):
                            m1 = lo + (hi - lo) * mp.mpf("0.381966011250105")
                            m2 = lo + (hi - lo) * mp.mpf("0.618033988749895")
                            if sgn * err(m1) < sgn * err(m2):
                                lo = m1
                            else:
                                hi = m2
                        x = (lo + hi) / 2
                    cand.append((x, err(x)))
        # merge runs of equal sign, keep the largest of each run
        merged = []
        for x, e in cand:
            if merged and (merged[-1][1] > 0) == (e > 0):
                if abs(e) > abs(merged[-1][1]):
                    merged[-1] = (x, e)
            else:
                merged.append((x, e))
        while len(merged) > deg + 2:
            if (len(merged) - (deg + 2)) % 2 == 1:
                if abs(merged[0][1]) < abs(merged[-1][1]):
                    merged.pop(0)
                else:
                    merged.pop()
            else:
                i = min(range(len(merged) - 1), key=lambda i: max(abs(merged[i][1]), abs(merged[i + 1][1])))
                del merged[i:i + 2]
        if len(merged) < deg + 2:
            raise RuntimeError("only %d alternations" % len(merged))
        hi = max(abs(e) for _, e in merged)
        lo = min(abs(e) for _, e in merged)
        if log:
            log("iter %d: levelled error %s, max %s, min %s" % (it, mp.nstr(abs(E), 8), mp.nstr(hi, 8), mp.nstr(lo, 8)))
        if it > 0 and (hi - lo) / lo < tol:
            return c, hi, [x for x, _ in merged]
        M = mp.matrix(deg + 2, deg + 2)
        rhs = mp.matrix(deg + 2, 1)
        for i, (x, _) in enumerate(merged):
            t = x / K
            M[i, 0] = 1
            M[i, 1] = t
            for j in range(2, deg + 1):
                M[i, j] = 2 * t * M[i, j - 1] - M[i, j - 2]
            M[i, deg + 1] = (-1) ** i
            rhs[i] = f(x)
        sol = mp.lu_solve(M, rhs)
        c = [sol[j] for j in range(deg + 1)]
        E = sol[deg + 1]
    raise RuntimeError("no convergence")


def main():
    K, deg, loge, r = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (25, 59, 10, 2)
    sf = 1 << r
    w = mp.mpf(2) ** -loge
    f = lambda x: mp.cos(2 * mp.pi * (x - mp.mpf(1) / 4) / sf)
    c, e, ref = remez(f, K, deg, w, log=lambda s: print(s, file=sys.stderr))
    out = {
        "about": "minimax polynomial of cos(2 pi (x - 1/4) / %d) on the union of [k - 2^-%d, k + 2^-%d], |k| <= %d, degree %d; "
                 "Chebyshev coefficients in the basis T_j(x / %d).  Computed by tools/remez_mpmath.py (own exchange iteration, "
                 "400-bit mpmath); by uniqueness of the minimax polynomial this is what the reference's Remez converges to "
                 "(include/source/bootstrapping/common/Remez.cpp:577-586, RemezCos.h:11-16)." % (sf, loge, loge, K - 1, deg, K),
        "boundary_K": K, "deg": deg, "log_width": loge, "scale_factor": sf,
        "minimax_error": mp.nstr(e, 25),
        "chebcoeff": [mp.nstr(v, 40) for v in c],
        "reference_points": [mp.nstr(x, 25) for x in ref],
    }
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
