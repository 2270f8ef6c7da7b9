"""exp_small_ (shaderbox_amd/csrc/sbx_math.h) against the oracle's m_exp on EVERY binary32 argument of its domain, on the host.

The kernel-internal form (degree-8 minimax polynomial on [-0.205, 0], one binary64 Horner chain, no argument reduction, no table)
is restated here in C with the coefficients READ from sbx_math.h, compiled with g++ (strict IEEE, hardware fma) and compared with
`sbxref::m_exp` of oracle/sbx_math_ref.h — the math spec — for all 1 045 556 103 arguments from -0.205f to -0 and for +0: the two
must round to the same binary32 value everywhere.  The same comparison runs on the GPU against exp_ in
tests/test_gpu_round3.py::test_exp_small_equals_exp_on_its_whole_domain; this one pins the claim without a GPU (about 10 s)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include "%(root)s/oracle/sbx_math_ref.h"
using namespace sbxref;
static inline float small8(float x) {
    const double xd = (double)x;
    double p = fma(%(c8)s, xd, %(c7)s);
    p = fma(p, xd, %(c6)s); p = fma(p, xd, %(c5)s); p = fma(p, xd, %(c4)s);
    p = fma(p, xd, %(c3)s); p = fma(p, xd, %(c2)s); p = fma(p, xd, %(c1)s);
    p = fma(p, xd, 1.0);
    return (float)p;
}
int main() {
    const uint32_t lim = f2u(%(lim)sf);
    long bad = 0;
    for (uint32_t b = 0; b <= lim; ++b) {
        const float x = u2f(b | 0x80000000u);
        if (f2u(small8(x)) != f2u(m_exp(x))) { if (bad < 5) printf("x=%%a\n", x); ++bad; }
    }
    if (f2u(small8(0.f)) != f2u(m_exp(0.f))) ++bad;
    printf("checked %%u mismatches %%ld\n", lim + 2, bad);
    return bad != 0;
}
"""


def test_exp_small_equals_the_spec_exp_on_its_whole_domain(tmp_path):
    text = open(os.path.join(ROOT, "shaderbox_amd", "csrc", "sbx_math.h")).read()
    body = text[text.index("float exp_small_(float x)"):]
    coef = dict(re.findall(r"\b(c[1-8]) = (-?0x[0-9a-f.]+p[-+]\d+)", body[:1200]))
    assert sorted(coef) == ["c%d" % i for i in range(1, 9)]
    lim = re.search(r"EXP_SMALL_MIN = -([0-9.]+)f", text).group(1)
    src = tmp_path / "exh.cpp"
    src.write_text(SRC % dict(coef, root=ROOT, lim=lim))
    exe = tmp_path / "exh"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-msse4.1", "-o", str(exe), str(src)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    assert "mismatches 0" in r.stdout and "checked 1045556103" in r.stdout, r.stdout


SRC4K = r"""
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include "%(root)s/oracle/sbx_math_ref.h"
#include "%(root)s/shaderbox_amd/csrc/sbx_exp4k_table.h"
using namespace sbxref;
static const double TAB[4096] = {SBX_EXP2_TAB4096_VALUES};
static inline float e4k(float x) {
    const double xd = (double)x;
    double kd = fma(xd, %(inv)s, SBX_D_MAGIC);
    const int32_t ki = (int32_t)(uint32_t)(d2u(kd) & 0xffffffffull);
    kd = kd - SBX_D_MAGIC;
    double r = fma(kd, %(hi)s, xd);
    r = fma(kd, %(lo)s, r);
    double p = fma(%(c3)s, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexpf((float)(p * TAB[ki & 4095]), ki >> 12);
}
int main() {
    long bad = 0;
    unsigned long n = 0;
    for (int sign = 0; sign < 2; ++sign) {
        const uint32_t lim = f2u(sign ? 80.0f : 262144.0f);      // [-80, 2^18]
        for (uint32_t b = 0; b <= lim; ++b, ++n) {
            const float x = u2f(b | (sign ? 0x80000000u : 0u));
            if (f2u(e4k(x)) != f2u(m_exp(x))) { if (bad < 5) printf("x=%%a\n", x); ++bad; }
        }
    }
    printf("checked %%lu mismatches %%ld\n", n, bad);
    return bad != 0;
}
"""


def test_exp_reg4k_equals_the_spec_exp_for_all_arguments_up_to_80(tmp_path):
    """exp_reg4k_ (sbx_math.h: 4096-entry table, degree 3 — k_atmosphere's density terms) restated in C with the constants read from
    sbx_math.h and the table of sbx_exp4k_table.h, against the oracle's m_exp on every binary32 argument in [-80, 2^18] (about 25 s):
    |x| <= 80 is what CLOUDS / CLOUDS_TEX / PLANET produce; ATMOSPHERE's view rays below the horizon reach +5300 (overflow to +inf)."""
    text = open(os.path.join(ROOT, "shaderbox_amd", "csrc", "sbx_math.h")).read()
    body = text[text.index("float exp_reg4k_(float x"):][:1200]
    inv = re.search(r"fma\(xd, (0x[0-9a-f.]+p[-+]\d+), D_MAGIC\)", body).group(1)
    hi, lo = re.findall(r"fma\(kd, (-0x[0-9a-f.]+p[-+]\d+),", body)
    c3 = re.search(r"fma\((0x[0-9a-f.]+p[-+]\d+), r, 0\.5\)", body).group(1)
    src = tmp_path / "exh4k.cpp"
    src.write_text(SRC4K % dict(root=ROOT, inv=inv, hi=hi, lo=lo, c3=c3))
    exe = tmp_path / "exh4k"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-msse4.1", "-o", str(exe), str(src)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "mismatches 0" in r.stdout and "checked 2334130178" in r.stdout, r.stdout


SRC_SIN = r"""
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <thread>
#include <vector>
#include <atomic>
#include "%(root)s/oracle/sbx_math_ref.h"
using namespace sbxref;
static const double C[7] = {%(coefs)s};
static inline float sin15(float xf) {
    const double x = (double)xf;
    double kd = fma(x, SBX_D_INV_PI, SBX_D_MAGIC);
    const uint64_t flip = d2u(kd) << 63;
    kd = kd - SBX_D_MAGIC;
    double r = fma(kd, -SBX_D_PI, x);
    r = fma(kd, -SBX_D_PI_LO, r);
    const double s = r * r;
    double p = C[0];
    for (int i = 1; i < 7; ++i) p = fma(p, s, C[i]);
    return (float)u2d(d2u(fma(r * s, p, r)) ^ flip);
}
int main() {
    const uint32_t lim = f2u(%(lim)s);
    std::atomic<long> bad{0};
    std::vector<std::thread> th;
    const int T = 8;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t]() {
        long b = 0;
        for (uint64_t i = t; i <= lim; i += T)
            for (uint32_t sign = 0; sign < 2; ++sign) {
                const float x = u2f((uint32_t)i | (sign << 31));
                if (f2u(sin15(x)) != f2u(m_sin(x))) ++b;
            }
        bad += b;
    });
    for (auto& t : th) t.join();
    printf("checked %%lu mismatches %%ld\n", 2ul * ((unsigned long)lim + 1), bad.load());
    return bad.load() != 0;
}
"""


def test_sin_b40_equals_the_spec_sin_up_to_2_pow_40(tmp_path):
    """sin_b40_ (sbx_math.h: the spec's argument reduction, a degree-15 minimax polynomial instead of the Taylor polynomial to r^21 —
    the hash passes' sin) restated in C with the coefficients read from sbx_math.h, against the oracle's m_sin on EVERY binary32
    argument with |x| <= 2^40 (2.8e9 values, 8 threads)."""
    text = open(os.path.join(ROOT, "shaderbox_amd", "csrc", "sbx_math.h")).read()
    coefs = re.search(r"#define SBX_SIN15_COEFS (.*?)/\*", text, re.S).group(1).replace("\\\n", " ")
    assert len(re.findall(r"0x", coefs)) == 7
    lim = re.search(r"SIN_B40_MAX = (0x1p\+40f)", text).group(1)
    src = tmp_path / "exhsin.cpp"
    src.write_text(SRC_SIN % dict(root=ROOT, coefs=coefs, lim=lim))
    exe = tmp_path / "exhsin"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-msse4.1", "-pthread", "-o", str(exe), str(src)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "mismatches 0" in r.stdout and "checked 2801795074" in r.stdout, r.stdout
