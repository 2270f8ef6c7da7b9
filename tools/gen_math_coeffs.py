#!/usr/bin/env python3
"""Derive the polynomial coefficients of the sbx math spec (docs in DESIGN.md §Math spec).

The reference (valentingalea/shaderbox) leaves sin/cos/exp/pow/acos/atan to its
environment (GLSL driver / HLSL / VML+libm, see src/def.h:1-42), so this build has
to *define* them once and use the same definition bit-for-bit in the CPU oracle
(oracle/sbx_math_ref.h) and in the HIP kernels (shaderbox_amd/csrc/sbx_math.h).

This script produces the fp32 minimax coefficients for

  sin(r)  = r + r^3 * P(r^2),      r in [-pi/2, pi/2],   P of degree 4
  exp(r)  = 1 + r + r^2 * Q(r),    r in [-ln2/2, ln2/2], Q of degree 4

by solving the discrete weighted minimax problem as a linear program, with
sequential rounding of the coefficients to binary32 (leading coefficient first).
Run:  python tools/gen_math_coeffs.py      (prints C initialisers)
"""
import numpy as np
import mpmath as mp
from scipy.optimize import linprog

mp.mp.prec = 120


def minimax_lp(A, g, w, fixed=None):
    """min_c max_i |w_i * (A c - g)_i| ; `fixed` = {index: value} pins coefficients."""
    n = A.shape[1]
    fixed = fixed or {}
    free = [j for j in range(n) if j not in fixed]
    g2 = g - sum(A[:, j] * v for j, v in fixed.items())
    Af = A[:, free] * w[:, None]
    gf = g2 * w
    m = len(free)
    # variables: c_free (m), E
    c = np.zeros(m + 1)
    c[-1] = 1.0
    A_ub = np.vstack([np.hstack([Af, -np.ones((len(gf), 1))]),
                      np.hstack([-Af, -np.ones((len(gf), 1))])])
    b_ub = np.concatenate([gf, -gf])
    res = linprog(c, A_ub=A_ub, b_ub=b_ub, bounds=[(None, None)] * m + [(0, None)],
                  method="highs")
    assert res.status == 0, res.message
    out = np.zeros(n)
    for j, v in fixed.items():
        out[j] = v
    for k, j in enumerate(free):
        out[j] = res.x[k]
    return out, res.x[-1]


def seq_round(A, g, w):
    """Sequentially round coefficients to fp32, lowest order first, re-solving the rest."""
    n = A.shape[1]
    fixed = {}
    for j in range(n):
        c, E = minimax_lp(A, g, w, fixed)
        fixed[j] = float(np.float32(c[j]))
    c = np.array([fixed[j] for j in range(n)])
    E = np.max(np.abs(w * (A @ c - g)))
    return c, E


def fit_sin():
    N = 4000
    # Chebyshev-distributed nodes in r on (0, pi/2]
    k = np.arange(1, N + 1)
    r = (np.pi / 2) * np.sin(0.5 * np.pi * k / N) ** 1.0
    r = np.unique(np.concatenate([r, np.linspace(1e-3, np.pi / 2, N)]))
    s = r * r
    # g(s) = (sin(r) - r) / r^3 evaluated in high precision
    g = np.array([float((mp.sin(mp.mpf(x)) - mp.mpf(x)) / mp.mpf(x) ** 3) for x in r])
    A = np.stack([s ** j for j in range(5)], axis=1)
    # error in sin = r^3 * dP ; relative to sin(r)
    w = r ** 3 / np.sin(r)
    c, E = seq_round(A, g, w)
    return c, E


def fit_exp():
    N = 4000
    h = float(mp.log(2) / 2) * 1.0001
    r = np.unique(np.concatenate([h * np.cos(np.pi * (np.arange(N) + 0.5) / N),
                                  np.linspace(-h, h, N)]))
    r = r[np.abs(r) > 1e-4]
    g = np.array([float((mp.exp(mp.mpf(x)) - 1 - mp.mpf(x)) / mp.mpf(x) ** 2) for x in r])
    A = np.stack([r ** j for j in range(5)], axis=1)
    w = r ** 2 / np.exp(r)
    c, E = seq_round(A, g, w)
    return c, E


def hexf(x):
    return float(np.float32(x)).hex()


if __name__ == "__main__" and not any(a.startswith("--") for a in __import__("sys").argv[1:]):
    c, E = fit_sin()
    print("// sin(r) = r + r^3*(S0 + S1 s + S2 s^2 + S3 s^3 + S4 s^4), s=r^2 ; max rel err (exact arith) = %.3g" % E)
    for j, v in enumerate(c):
        print("  S%d = %s  /* %.10e */" % (j, hexf(v), v))
    c, E = fit_exp()
    print("// exp(r) = 1 + r + r^2*(E0 + E1 r + E2 r^2 + E3 r^3 + E4 r^4) ; max rel err (exact arith) = %.3g" % E)
    for j, v in enumerate(c):
        print("  E%d = %s  /* %.10e */" % (j, hexf(v), v))
    # Cody-Waite splits
    pi = mp.pi
    hi = np.float32(float(pi))
    mid = np.float32(float(pi - mp.mpf(float(hi))))
    lo = np.float32(float(pi - mp.mpf(float(hi)) - mp.mpf(float(mid))))
    print("// pi = PI_HI + PI_MID + PI_LO")
    print("  PI_HI = %s, PI_MID = %s, PI_LO = %s" % (hexf(hi), hexf(mid), hexf(lo)))
    hp = pi / 2
    hi = np.float32(float(hp))
    mid = np.float32(float(hp - mp.mpf(float(hi))))
    lo = np.float32(float(hp - mp.mpf(float(hi)) - mp.mpf(float(mid))))
    print("  PIO2_HI = %s, PIO2_MID = %s, PIO2_LO = %s" % (hexf(hi), hexf(mid), hexf(lo)))
    print("  INV_PI = %s" % hexf(float(1 / pi)))
    ln2 = mp.log(2)
    hi = np.float32(float(ln2))
    lo = np.float32(float(ln2 - mp.mpf(float(hi))))
    print("  LN2_HI = %s, LN2_LO = %s, LOG2E = %s" % (hexf(hi), hexf(lo), hexf(float(1 / ln2))))
    print("  (double) 1/ln2 = %s ; ln2 = %s" % (float(1 / ln2).hex(), float(ln2).hex()))


def exp_table():
    """Constants of the table form of exp (sbx_math.h exp_, oracle m_exp): 2^(j/32) correctly rounded to
    binary64, 32/ln2, and ln2/32 split into a 38-bit high part (k*hi exact for |k| < 2^14) and the rest."""
    import struct
    tab = [float(mp.mpf(2) ** (mp.mpf(j) / 32)) for j in range(32)]
    c = mp.log(2) / 32
    b = struct.unpack("<Q", struct.pack("<d", float(c)))[0] & ~((1 << 15) - 1)
    hi = struct.unpack("<d", struct.pack("<Q", b))[0]
    lo = float(c - mp.mpf(hi))
    print("// 2^(j/32), j = 0..31")
    for i in range(0, 32, 4):
        print("    " + ", ".join(float.hex(v) for v in tab[i:i + 4]) + ",")
    print("// 32/ln2 = %s ; ln2/32 = %s + %s" % (float.hex(float(32 / mp.log(2))), float.hex(hi), float.hex(lo)))
    print("// 1/n!, n = 2..6: " + ", ".join(float.hex(float(1 / mp.factorial(n))) for n in range(2, 7)))


def exp_table_n(n):
    """2^(j/n), j = 0..n-1, correctly rounded to binary64, and n/ln2, ln2/n split as in exp_table(): the constants of exp_reg64_"""
    import struct
    tab = [float(mp.mpf(2) ** (mp.mpf(j) / n)) for j in range(n)]
    c = mp.log(2) / n
    b = struct.unpack("<Q", struct.pack("<d", float(c)))[0] & ~((1 << 15) - 1)
    hi = struct.unpack("<d", struct.pack("<Q", b))[0]
    for i in range(0, n, 4):
        print("    " + ", ".join(float.hex(v) for v in tab[i:i + 4]) + ",")
    print("// %d/ln2 = %s ; ln2/%d = %s + %s" % (n, float.hex(float(n / mp.log(2))), n, float.hex(hi), float.hex(float(c - mp.mpf(hi)))))


def exp_table_header(n, keep):
    """shaderbox_amd/csrc/sbx_exp4k_table.h: 2^(j/n) correctly rounded to binary64 as one macro, with the constants of exp_reg4k_"""
    import struct
    mp.mp.prec = 200
    tab = [float(mp.mpf(2) ** (mp.mpf(j) / n)) for j in range(n)]
    c = mp.log(2) / n
    b = struct.unpack("<Q", struct.pack("<d", float(c)))[0] & ~((1 << (53 - keep)) - 1)
    hi = struct.unpack("<d", struct.pack("<Q", b))[0]
    print("// shaderbox_amd/csrc/sbx_exp4k_table.h — GENERATED by `python tools/gen_math_coeffs.py --exp-table-header %d`: do not edit." % n)
    print("// 2^(j/%d), j = 0..%d, correctly rounded to binary64 (mpmath, 200 bits): the table of exp_reg4k_ (sbx_math.h)." % (n, n - 1))
    print("// %d/ln2 = %s ; ln2/%d = %s (high %d bits: k * hi is exact for |k| < 2^%d) + %s"
          % (n, float.hex(float(n / mp.log(2))), n, float.hex(hi), keep, 53 - keep, float.hex(float(c - mp.mpf(hi)))))
    print("#pragma once")
    print("#define SBX_EXP2_TAB%d_VALUES \\" % n)
    for i in range(0, n, 4):
        print("    " + ", ".join(float.hex(v) for v in tab[i:i + 4]) + ("," if i + 4 < n else "") + (" \\" if i + 4 < n else ""))
    print("#if defined(__HIP_DEVICE_COMPILE__)")
    print("static __device__ const double kExp2Tab%d[%d] = {SBX_EXP2_TAB%d_VALUES};      // one copy per translation unit that includes this" % (n, n, n))
    print("#else")
    print("static const double kExp2Tab%d[1] = {1.0};          // (host pass: kernel bodies only have to parse)" % n)
    print("#endif")


if __name__ == "__main__" and "--exp-table-header" in __import__("sys").argv:
    exp_table_header(int(__import__("sys").argv[__import__("sys").argv.index("--exp-table-header") + 1]), 32)
    raise SystemExit(0)


if __name__ == "__main__" and "--exp-table" in __import__("sys").argv:
    a = __import__("sys").argv
    i = a.index("--exp-table")
    if i + 1 < len(a) and a[i + 1].isdigit():
        mp.mp.prec = 200
        exp_table_n(int(a[i + 1]))
    else:
        exp_table()


def log2_table():
    """Constants of the table form of log2 used by pow (sbx_math.h d_log2, oracle d_log2).
    x = 2^k z with z in [0.6875, 1.375): tmp = bits(x) - OFF (OFF = bits(0.6875)), i = (tmp >> 45) & 127,
    k = tmp >> 52 (arithmetic), z = bits(x) - (tmp & 0xfff0000000000000).  Interval i of z has centre c_i; the two
    intervals next to 1.0 (i = 79, 80) take c = 1 so that log2 keeps its relative accuracy near 1.
    invc_i = RN64(1 / c_i), logc_i = RN64(-log2(invc_i)) computed from the ROUNDED invc_i, so that
    log2(z) = logc_i + log2(z * invc_i) holds up to the rounding of logc_i alone.  |z * invc_i - 1| < 2^-7."""
    import struct
    OFF = 0x3fe6000000000000

    def dbl(bits):
        return struct.unpack("<d", struct.pack("<Q", bits))[0]
    rows = []
    rmax = mp.mpf(0)
    for i in range(128):
        lo, hi = dbl(OFF + (i << 45)), dbl(OFF + ((i + 1) << 45))
        c = mp.mpf(1) if i in (79, 80) else (mp.mpf(lo) + mp.mpf(hi)) / 2
        invc = float(1 / c)
        logc = float(-mp.log(mp.mpf(invc), 2))
        rows.append((invc, logc))
        for z in (lo, hi):
            rmax = max(rmax, abs(mp.mpf(z) * mp.mpf(invc) - 1))
    print("// {invc, logc}, i = 0..127; max |z*invc - 1| = %s" % mp.nstr(rmax, 6))
    for i in range(0, 128, 2):
        print("    " + ", ".join("{%s, %s}" % (float.hex(a), float.hex(b)) for a, b in rows[i:i + 2]) + ",")
    print("// log2(1 + r) = r * (A0 + A1 r + ... + A7 r^7), A_j = (-1)^j / ((j + 1) ln 2):")
    print("    " + ", ".join(float.hex(float((-1) ** j / ((j + 1) * mp.log(2)))) for j in range(8)))


if __name__ == "__main__" and "--log2-table" in __import__("sys").argv:
    log2_table()


def exp_small(X="0.205", N=8):
    """Coefficients of exp_small_ (sbx_math.h): p(x) = 1 + c1 x + ... + cN x^N minimising the RELATIVE error against e^x on
    [-X, 0] (Remez exchange on a fine grid, 60 digits; p(0) = 1 is built in so that exp(-0) = exp(+0) = 1 exactly)."""
    mp.mp.dps = 60
    X = mp.mpf(X)
    a, b = -X, mp.mpf(0)
    m = N + 1

    def err(c, x):
        return (1 + sum(c[k] * x ** (k + 1) for k in range(N)) - mp.e ** x) / mp.e ** x
    ref = [a + (b - a) * (1 - mp.cos(mp.pi * i / m)) / 2 for i in range(m)]          # (x = 0 is never a reference: the error is 0 there)
    for _ in range(30):
        A, rhs = mp.matrix(m, m), mp.matrix(m, 1)
        for i, x in enumerate(ref):
            for k in range(N):
                A[i, k] = x ** (k + 1)
            A[i, N] = -((-1) ** i) * mp.e ** x
            rhs[i] = mp.e ** x - 1
        sol = mp.lu_solve(A, rhs)
        c, E = [sol[k] for k in range(N)], sol[N]
        G = 4000
        xs = [a + (b - a) * mp.mpf(i) / G for i in range(G + 1)]
        es = [err(c, x) for x in xs]
        ext = []
        for i in range(G + 1):
            if es[i] != 0 and (i == 0 or abs(es[i]) >= abs(es[i - 1])) and (i == G or abs(es[i]) >= abs(es[i + 1])):
                if ext and (ext[-1][1] > 0) == (es[i] > 0):
                    if abs(es[i]) > abs(ext[-1][1]):
                        ext[-1] = (xs[i], es[i])
                else:
                    ext.append((xs[i], es[i]))
        while len(ext) > m:
            ext.pop(0 if abs(ext[0][1]) < abs(ext[-1][1]) else -1)
        ref = [x for x, _ in ext]
        if max(abs(v) for _, v in ext) / min(abs(v) for _, v in ext) < mp.mpf("1.0001"):
            break
    cr = [mp.mpf(float(ck)) for ck in c]
    worst = max(abs(err(cr, a + (b - a) * mp.mpf(i) / 20000)) for i in range(20000))
    print("// exp on [-%s, 0], degree %d: minimax relative error 2^%s; with the binary64 coefficients below 2^%s"
          % (mp.nstr(X, 4), N, mp.nstr(mp.log(abs(E), 2), 5), mp.nstr(mp.log(worst, 2), 5)))
    print("    " + ", ".join("c%d = %s" % (k + 1, float.hex(float(c[k]))) for k in reversed(range(N))))


if __name__ == "__main__" and "--exp-small" in __import__("sys").argv:
    exp_small()


def sin15(N=7):
    """Coefficients of sin_b40_ (sbx_math.h): sin r ~ r + c3 r^3 + ... + c15 r^15 minimising the RELATIVE error on (0, pi/2 + 1e-6]
    (Remez exchange on a fine grid, 70 digits); printed from r^15 down to r^3."""
    mp.mp.dps = 70
    a, b = mp.mpf("1e-3"), mp.pi / 2 + mp.mpf("1e-6")
    m = N + 1

    def err(c, r):
        return (r + sum(c[k] * r ** (2 * k + 3) for k in range(N)) - mp.sin(r)) / mp.sin(r)
    ref = [a + (b - a) * (1 - mp.cos(mp.pi * (i + 0.5) / m)) / 2 for i in range(m)]
    for _ in range(40):
        A, rhs = mp.matrix(m, m), mp.matrix(m, 1)
        for i, r in enumerate(ref):
            for k in range(N):
                A[i, k] = r ** (2 * k + 3)
            A[i, N] = -((-1) ** i) * mp.sin(r)
            rhs[i] = mp.sin(r) - r
        sol = mp.lu_solve(A, rhs)
        c, E = [sol[k] for k in range(N)], sol[N]
        G = 6000
        xs = [a + (b - a) * mp.mpf(i) / G for i in range(G + 1)]
        es = [err(c, x) for x in xs]
        ext = []
        for i in range(G + 1):
            if es[i] != 0 and (i == 0 or abs(es[i]) >= abs(es[i - 1])) and (i == G or abs(es[i]) >= abs(es[i + 1])):
                if ext and (ext[-1][1] > 0) == (es[i] > 0):
                    if abs(es[i]) > abs(ext[-1][1]):
                        ext[-1] = (xs[i], es[i])
                else:
                    ext.append((xs[i], es[i]))
        while len(ext) > m:
            ext.pop(0 if abs(ext[0][1]) < abs(ext[-1][1]) else -1)
        if len(ext) < m:          # the error curve has flattened to the grid's resolution: the current solution stands
            break
        ref = [x for x, _ in ext]
        if max(abs(v) for _, v in ext) / min(abs(v) for _, v in ext) < mp.mpf("1.001"):
            break
    print("// sin r = r + c3 r^3 + ... + c%d r^%d on (0, pi/2]: levelled relative error 2^%s" % (2 * N + 1, 2 * N + 1, mp.nstr(mp.log(abs(E), 2), 5)))
    print("    " + ", ".join(float.hex(float(ck)) for ck in reversed(c)))


if __name__ == "__main__" and "--sin15" in __import__("sys").argv:
    sin15()
