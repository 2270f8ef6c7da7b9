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
