"""Accuracy of the oracle's statement of the sbx math spec (oracle/sbx_math_ref.h).

Spec: transcendentals are evaluated in binary64 and rounded once to binary32, i.e. they are the
correctly rounded binary32 results (up to ties closer than ~2^-28 ulp).  sin on integer arguments
|n| <= 2^21 — the whole domain of the noise hash — is checked EXHAUSTIVELY; the others on dense samples.
Reference values: numpy float64 libm rounded to float32 (its error, <1 ulp of double, is 2^-29 ulp of
float, so it decides correct rounding except on near-ties, of which we allow a vanishing fraction)."""
import numpy as np
import pytest


def _mismatch(got, want64):
    want = want64.astype(np.float32)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    return int((~same).sum())


def test_sin_is_correctly_rounded_on_all_hash_arguments(oracle):
    n = np.arange(-(1 << 21), (1 << 21) + 1, dtype=np.float32)
    assert _mismatch(oracle.math("sin", n), np.sin(n.astype(np.float64))) == 0
    assert _mismatch(oracle.math("cos", n), np.cos(n.astype(np.float64))) == 0


def test_transcendentals_are_correctly_rounded_on_samples(oracle):
    rng = np.random.default_rng(11)
    x = (rng.standard_normal(2_000_000) * 2000).astype(np.float32)
    assert _mismatch(oracle.math("sin", x), np.sin(x.astype(np.float64))) <= 2
    assert _mismatch(oracle.math("cos", x), np.cos(x.astype(np.float64))) <= 2
    x = rng.uniform(-1.5, 1.5, 1_000_000).astype(np.float32)
    assert _mismatch(oracle.math("tan", x), np.tan(x.astype(np.float64))) <= 2
    x = rng.uniform(-87, 88, 2_000_000).astype(np.float32)
    assert _mismatch(oracle.math("exp", x), np.exp(x.astype(np.float64))) <= 2
    x = np.abs(rng.standard_normal(2_000_000)).astype(np.float32) * 3
    for y in (1 / 2.2, 1.5, 10.0, 30.0, 1500.0):
        yy = np.float32(y)
        w = np.power(x.astype(np.float64), np.float64(yy))
        ok = np.isfinite(w) & (w > 1e-37)
        assert _mismatch(oracle.math("pow", x, yy)[ok], w[ok]) <= 2
    x = rng.uniform(-1, 1, 1_000_000).astype(np.float32)
    assert _mismatch(oracle.math("acos", x), np.arccos(x.astype(np.float64))) <= 2
    a, b = rng.standard_normal(1_000_000).astype(np.float32), rng.standard_normal(1_000_000).astype(np.float32)
    assert _mismatch(oracle.math("atan2", a, b), np.arctan2(a.astype(np.float64), b.astype(np.float64))) <= 2


def test_special_values(oracle):
    f = np.float32
    assert oracle.math("exp", np.array([0, -np.inf, np.inf, 89, -104], f)).tolist() == [1.0, 0.0, np.inf, np.inf, 0.0]
    assert np.isnan(oracle.math("exp", np.array([np.nan], f))[0])
    assert np.isnan(oracle.math("sin", np.array([np.inf], f))[0])
    p = oracle.math("pow", np.array([0, 0, 2, -1, 1, np.nan, 0.5], f), np.array([1.5, 0, 0, .5, 1500, 1, 1500], f))
    assert p[0] == 0 and p[1] == 1 and p[2] == 1 and np.isnan(p[3]) and p[4] == 1 and np.isnan(p[5]) and p[6] == 0
    assert [float(v) for v in oracle.math("pow", np.array([2, 2, 2, 2, 2], f), np.array([1, 2, 3, 4, 5], f))] == [2, 4, 8, 16, 32]
    a = oracle.math("acos", np.array([1, -1, 1.5, -3.2], f))
    assert a[0] == 0 and a[1] == f(np.pi) and np.isnan(a[2]) and np.isnan(a[3])   # NaN dir -> black (App. B3)


def test_exp_is_correctly_rounded_against_mpmath(oracle):
    """m_exp (table form) against exp() in 200-bit arithmetic, rounded once to binary32 — over the whole result
    range including denormal results, and on the one input where the former 13-term form misrounded."""
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 200
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-104, 89, 6000), rng.uniform(-1, 1, 2000), rng.uniform(-104, -87, 2000)]).astype(np.float32)
    x = np.concatenate([x, np.array([0xc2b2e798], np.uint32).view(np.float32)])
    got = oracle.math("exp", x)

    def rn32(v):                      # round-to-nearest-even of a positive mpf to binary32, denormals included
        if v >= mp.mpf(2) ** 128:
            return np.float32(np.inf)
        e = max(int(mp.floor(mp.log(v, 2))), -126)
        q = v / mp.mpf(2) ** (e - 23)
        n = int(mp.floor(q)); f = q - n
        if f > 0.5 or (f == 0.5 and n & 1):
            n += 1
        return np.float32(float(mp.mpf(n) * mp.mpf(2) ** (e - 23)))
    want = np.array([rn32(mp.exp(mp.mpf(float(v)))) for v in x], np.float32)
    assert (got.view(np.uint32) == want.view(np.uint32)).all(), (x[got != want][:5], got[got != want][:5], want[got != want][:5])
    assert got[-1].view(np.uint32) == 0x000f6dce


def test_pow_is_correctly_rounded_against_mpmath(oracle):
    """m_pow (table forms of log2 and 2^t) against x**y in 200-bit arithmetic rounded once to binary32, for the
    exponents the apps use (1/2.2 of linear_to_srgb, 1.5 of the phase functions, 10/30/50/1500 of the sun lobes)."""
    mp = pytest.importorskip("mpmath")
    mp.mp.prec = 200
    rng = np.random.default_rng(5)

    def rn32(v):
        if v == 0:
            return np.float32(0)
        if v >= mp.mpf(2) ** 128:
            return np.float32(np.inf)
        e = max(int(mp.floor(mp.log(v, 2))), -126)
        q = v / mp.mpf(2) ** (e - 23)
        n = int(mp.floor(q)); f = q - n
        if f > 0.5 or (f == 0.5 and n & 1):
            n += 1
        return np.float32(float(mp.mpf(n) * mp.mpf(2) ** (e - 23)))
    for y in (1 / 2.2, 1.5, 10.0, 30.0, 50.0, 1500.0):
        yy = np.float32(y)
        x = np.concatenate([rng.uniform(0, 1.2, 700), rng.uniform(0.99, 1.01, 300), np.abs(rng.standard_normal(300)) * 5]).astype(np.float32)
        x = x[x > 0]
        got = oracle.math("pow", x, np.full_like(x, yy))
        want = np.array([rn32(mp.mpf(float(v)) ** mp.mpf(float(yy))) for v in x], np.float32)
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), (y, x[got != want][:3])


def test_math_tables_match_their_generator():
    """The 2^(j/32) and log2 {invc, logc} tables in oracle/sbx_math_ref.h and shaderbox_amd/csrc/sbx_math.h are the
    ones tools/gen_math_coeffs.py derives with mpmath (two independent statements of the spec, one provenance)."""
    mp = pytest.importorskip("mpmath")
    import os
    import re
    import struct
    mp.mp.prec = 200
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp_tab = [float(mp.mpf(2) ** (mp.mpf(j) / 32)) for j in range(32)]
    OFF = 0x3fe6000000000000

    def dbl(bits):
        return struct.unpack("<d", struct.pack("<Q", bits))[0]
    log_tab = []
    for i in range(128):
        lo, hi = dbl(OFF + (i << 45)), dbl(OFF + ((i + 1) << 45))
        c = mp.mpf(1) if i in (79, 80) else (mp.mpf(lo) + mp.mpf(hi)) / 2
        invc = float(1 / c)
        log_tab += [invc, float(-mp.log(mp.mpf(invc), 2))]
    hexf = r"-?0x[01]\.[0-9a-f]+p[+-]\d+"
    for path in ("oracle/sbx_math_ref.h", "shaderbox_amd/csrc/sbx_math.h"):
        txt = open(os.path.join(root, path)).read()
        e = txt[txt.index("0x1.0000000000000p+0, 0x1.059b0d3158574p+0"):]
        got = [float.fromhex(v) for v in re.findall(hexf, e)[:32]]
        assert got == exp_tab, path
        pairs = txt[txt.index("{0x1.734f0c541fe8dp+0"):]
        got = [float.fromhex(v) for v in re.findall(hexf, pairs)[:256]]
        assert got == log_tab, path
