/* oracle/sbx_math_ref.h — CPU statement of the sbx math spec.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under shaderbox_amd/ may include, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use
 * the oracle, and only as the checker.
 *
 * Why it exists: the reference leaves every scalar math function to its environment
 * (GLSL driver, HLSL intrinsics, or VML + libm: /root/reference/src/def.h:1-42 maps the
 * language, nothing in the tree defines sin/exp/pow...).  The noise hash
 * fract(sin(n)*753.5453123) (/root/reference/src/noise_iq.h:5-9) amplifies a 1-ulp
 * difference in sin by 753, so a CPU checker and a GPU kernel can only agree to 1e-4
 * per channel if both use ONE written-down definition.  That definition ("sbx math
 * spec") is restated here in plain scalar C++ from the spec text; the
 * HIP kernels carry their own statement of the same spec (shaderbox_amd/csrc/
 * sbx_math.h).  tests/test_math_parity.py checks that the two agree bit-for-bit and
 * tests/test_oracle_math.py checks this one against float64 libm within stated ulps.
 *
 * Rules of the spec:
 *   - every value is IEEE binary32 unless a function says "double inside";
 *   - + - * / sqrt are the IEEE correctly-rounded operations, evaluated in the written
 *     order, never contracted (build with -ffp-contract=off);
 *   - fmaf()/fma() appear only where written and mean the IEEE fused operation;
 *   - denormals are kept; NaN is data.
 * Coefficients come from tools/gen_math_coeffs.py.
 */
#ifndef SBX_MATH_REF_H
#define SBX_MATH_REF_H

#include <math.h>
#include <stdint.h>
#include <string.h>

namespace sbxref {

static inline uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
static inline uint64_t d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* ---- GLSL-style helpers (GLSL 4.x spec §8.3 formulas) ---------- */
static inline float m_min(float a, float b) { return (b < a) ? b : a; }
static inline float m_max(float a, float b) { return (a < b) ? b : a; }
static inline float m_clamp(float x, float lo, float hi) { return m_min(m_max(x, lo), hi); }
static inline float m_abs(float x) { return u2f(f2u(x) & 0x7fffffffu); }
static inline float m_floor(float x) { return floorf(x); }
static inline float m_fract(float x) { return x - floorf(x); }
static inline float m_mod(float x, float y) { return x - y * floorf(x / y); }
static inline float m_mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
static inline float m_step(float edge, float x) { return (x < edge) ? 0.0f : 1.0f; }
static inline float m_smoothstep(float e0, float e1, float x) {
    float t = m_clamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return (t * t) * (3.0f - 2.0f * t);
}
static inline float m_radians(float deg) { return deg * 0.017453292519943295f; }
static inline float m_sqrt(float x) { return sqrtf(x); }

/* ---- transcendental functions: "double inside" --------------------------------------
 * Every transcendental of the spec is evaluated in IEEE binary64 with the steps written
 * below and rounded once to binary32 at the end.  The binary64 value is accurate to
 * ~1 ulp of double, so the binary32 result is the CORRECTLY ROUNDED one unless the exact
 * value lies within ~2^-28 ulp of a rounding boundary.  For sin on integer arguments
 * |n| <= 2^21 (the whole domain of the noise hash) correct rounding is verified
 * exhaustively against mpmath (tests/test_oracle_math.py); that makes "correctly rounded
 * sin(n)" the spec on that domain, so an implementation may reach it by any route
 * (the HIP kernels use LDS trig tables + fp64 angle addition, checked exhaustively too).
 * Accuracy claims hold for |x| <= 2^22; beyond, results stay deterministic. */
static const double SBX_D_INV_LN2 = 0x1.71547652b82fep+0;
static const double SBX_D_LN2     = 0x1.62e42fefa39efp-1;
static const double SBX_D_SQRT2   = 0x1.6a09e667f3bcdp+0;
static const double SBX_D_MAGIC   = 6755399441055744.0; /* 1.5 * 2^52 */
static const double SBX_D_PI      = 0x1.921fb54442d18p+1;
static const double SBX_D_PI_LO   = 0x1.1a62633145c07p-53; /* pi - SBX_D_PI */
static const double SBX_D_PIO2    = 0x1.921fb54442d18p+0;
static const double SBX_D_INV_PI  = 0x1.45f306dc9c883p-2;

/* sin(r) for |r| <= pi/2 (+ a little): odd Taylor polynomial to r^21 */
static inline double d_sin_poly(double r) {
    double s = r * r;
    double p = 0x1.71b8ef6dcf572p-66;   /*  1/21! */
    p = fma(p, s, -0x1.2f49b46814157p-57);   /* -1/19! */
    p = fma(p, s, 0x1.952c77030ad4ap-49);   /*  1/17! */
    p = fma(p, s, -0x1.ae7f3e733b81fp-41);   /* -1/15! */
    p = fma(p, s, 0x1.6124613a86d09p-33);   /*  1/13! */
    p = fma(p, s, -0x1.ae64567f544e4p-26);   /* -1/11! */
    p = fma(p, s, 0x1.71de3a556c734p-19);   /*  1/9! */
    p = fma(p, s, -0x1.a01a01a01a01ap-13);   /* -1/7! */
    p = fma(p, s, 0x1.1111111111111p-7);   /*  1/5! */
    p = fma(p, s, -0x1.5555555555555p-3);   /* -1/3! */
    return fma(r * s, p, r);
}

/* k = rne(x/pi), r = x - k*pi (two-term pi), sin x = (-1)^k sin r */
static inline double d_sin(double x) {
    double kd = fma(x, SBX_D_INV_PI, SBX_D_MAGIC);
    uint64_t flip = d2u(kd) << 63;
    kd = kd - SBX_D_MAGIC;
    double r = fma(kd, -SBX_D_PI, x);
    r = fma(kd, -SBX_D_PI_LO, r);
    return u2d(d2u(d_sin_poly(r)) ^ flip);
}
/* cos x = sin(x + pi/2): k = rne(x/pi + 1/2), r = x - (k - 1/2)*pi */
static inline double d_cos(double x) {
    double kd = fma(x, SBX_D_INV_PI, 0.5) + SBX_D_MAGIC;
    uint64_t flip = d2u(kd) << 63;
    kd = kd - SBX_D_MAGIC;
    double m = kd - 0.5;
    double r = fma(m, -SBX_D_PI, x);
    r = fma(m, -SBX_D_PI_LO, r);
    return u2d(d2u(d_sin_poly(r)) ^ flip);
}
static inline float m_sin(float x) { return (float)d_sin((double)x); }
static inline float m_cos(float x) { return (float)d_cos((double)x); }
static inline float m_tan(float x) { return (float)(d_sin((double)x) / d_cos((double)x)); }

/* log2 of a finite positive double that came from a float */
static inline double d_log2(double x) {
    uint64_t b = d2u(x);
    int e = (int)(b >> 52) - 1023;
    double m = u2d((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > SBX_D_SQRT2) { m = m * 0.5; e = e + 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    /* ln(m) = 2s + s*z*(2/3 + 2/5 z + 2/7 z^2 + ... + 2/21 z^9) */
    double p = 0x1.8618618618618p-4;   /* 2/21 */
    p = fma(p, z, 0x1.af286bca1af28p-4);   /* 2/19 */
    p = fma(p, z, 0x1.e1e1e1e1e1e1ep-4);   /* 2/17 */
    p = fma(p, z, 0x1.1111111111111p-3);   /* 2/15 */
    p = fma(p, z, 0x1.3b13b13b13b14p-3);   /* 2/13 */
    p = fma(p, z, 0x1.745d1745d1746p-3);   /* 2/11 */
    p = fma(p, z, 0x1.c71c71c71c71cp-3);   /* 2/9 */
    p = fma(p, z, 0x1.2492492492492p-2);   /* 2/7 */
    p = fma(p, z, 0x1.999999999999ap-2);   /* 2/5 */
    p = fma(p, z, 0x1.5555555555555p-1);   /* 2/3 */
    double lnm = fma(s * z, p, 2.0 * s);
    return fma(lnm, SBX_D_INV_LN2, (double)e);
}

/* 2^t for t in [-160, 136], as a double */
static inline double d_exp2(double t) {
    double kd = t + SBX_D_MAGIC;
    int32_t ki = (int32_t)(uint32_t)(d2u(kd) & 0xffffffffull);
    kd = kd - SBX_D_MAGIC;
    double u = (t - kd) * SBX_D_LN2;
    /* exp(u), |u| <= 0.3466 : Taylor to u^13 */
    double p = 0x1.6124613a86d09p-33;   /* 1/13! */
    p = fma(p, u, 0x1.1eed8eff8d898p-29);   /* 1/12! */
    p = fma(p, u, 0x1.ae64567f544e4p-26);   /* 1/11! */
    p = fma(p, u, 0x1.27e4fb7789f5cp-22);   /* 1/10! */
    p = fma(p, u, 0x1.71de3a556c734p-19);   /* 1/9! */
    p = fma(p, u, 0x1.a01a01a01a01ap-16);   /* 1/8! */
    p = fma(p, u, 0x1.a01a01a01a01ap-13);   /* 1/7! */
    p = fma(p, u, 0x1.6c16c16c16c17p-10);   /* 1/6! */
    p = fma(p, u, 0x1.1111111111111p-7);   /* 1/5! */
    p = fma(p, u, 0x1.5555555555555p-5);   /* 1/4! */
    p = fma(p, u, 0x1.5555555555555p-3);   /* 1/3! */
    p = fma(p, u, 0x1.0000000000000p-1);   /* 1/2! */
    p = fma(p, u, 1.0);
    p = fma(p, u, 1.0);
    double sc = u2d((uint64_t)(int64_t)(ki + 1023) << 52);
    return p * sc;
}

/* exp: x = (k/32) ln2 + r, |r| <= ln2/64; exp(x) = 2^(k>>5) * 2^((k&31)/32) * exp(r).
 * k by the 1.5*2^52 trick on fma(x, 32/ln2, magic); r by a two-term Cody-Waite reduction (the high part of
 * ln2/32 has 38 significant bits, so k*hi is exact for |k| < 2^14); exp(r) by the degree-6 Taylor polynomial
 * (remainder < 3.5e-18 relative); 2^(j/32) from a table of correctly rounded doubles (tools/gen_math_coeffs.py).
 * The binary64 value is within ~1.2e-16 relative of exp(x) and is rounded once to binary32.
 * (The 13-term form in d_exp2, which this replaced as the definition of exp, gives the same binary32 result on
 * all 2^32 inputs but one, x = -89.45233 (0xc2b2e798), where it misrounds a denormal by an ulp and this one does
 * not — tests/test_gpu_parity.py::test_exp_table_vs_horner.)  pow keeps d_exp2. */
static const double SBX_EXP2_TAB[32] = {
    0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0,
    0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0, 0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0,
    0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0,
    0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0,
    0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0, 0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0,
    0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0,
    0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0,
    0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0, 0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0};
static inline float m_exp(float x) {
    if (x != x) return x;
    if (x < -104.0f) x = -104.0f;            /* below: rounds to 0 either way; above 89: overflows to +inf */
    if (x > 89.0f) x = 89.0f;
    const double xd = (double)x;
    double kd = fma(xd, 0x1.71547652b82fep+5 /* 32/ln2 */, SBX_D_MAGIC);
    const int32_t ki = (int32_t)(uint32_t)(d2u(kd) & 0xffffffffull);
    kd = kd - SBX_D_MAGIC;
    double r = fma(kd, -0x1.62e42fefa0000p-6 /* ln2/32, high 38 bits */, xd);
    r = fma(kd, -0x1.cf79abc9e3b3ap-45 /* ln2/32 - high */, r);
    double p = 0x1.6c16c16c16c17p-10;            /* 1/6! */
    p = fma(p, r, 0x1.1111111111111p-7);         /* 1/5! */
    p = fma(p, r, 0x1.5555555555555p-5);         /* 1/4! */
    p = fma(p, r, 0x1.5555555555555p-3);         /* 1/3! */
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double y = p * SBX_EXP2_TAB[ki & 31];
    return (float)u2d(d2u(y) + ((uint64_t)(int64_t)(ki >> 5) << 52));
}

static inline float m_pow(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x != x || y != y) return u2f(0x7fc00000u);
    if (x < 0.0f) return u2f(0x7fc00000u);
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : u2f(0x7f800000u);
    if (x == u2f(0x7f800000u)) return (y > 0.0f) ? x : 0.0f;
    double t = (double)y * d_log2((double)x);
    if (t < -160.0) t = -160.0;
    if (t > 136.0) t = 136.0;
    return (float)d_exp2(t);
}

/* atan of a non-negative double (inf allowed) */
static inline double d_atan_pos(double z) {
    bool inv = z > 1.0;
    if (inv) z = 1.0 / z;
    z = z / (1.0 + sqrt(fma(z, z, 1.0)));
    z = z / (1.0 + sqrt(fma(z, z, 1.0)));
    double w = z * z;
    /* atan z = z * (1 - w/3 + w^2/5 - ... - w^11/23), z <= tan(pi/16) */
    double p = -0x1.642c8590b2164p-5;   /* -1/23 */
    p = fma(p, w, 0x1.8618618618618p-5);   /*  1/21 */
    p = fma(p, w, -0x1.af286bca1af28p-5);   /* -1/19 */
    p = fma(p, w, 0x1.e1e1e1e1e1e1ep-5);   /*  1/17 */
    p = fma(p, w, -0x1.1111111111111p-4);   /* -1/15 */
    p = fma(p, w, 0x1.3b13b13b13b14p-4);   /*  1/13 */
    p = fma(p, w, -0x1.745d1745d1746p-4);   /* -1/11 */
    p = fma(p, w, 0x1.c71c71c71c71cp-4);   /*  1/9 */
    p = fma(p, w, -0x1.2492492492492p-3);   /* -1/7 */
    p = fma(p, w, 0x1.999999999999ap-3);   /*  1/5 */
    p = fma(p, w, -0x1.5555555555555p-2);   /* -1/3 */
    p = fma(p, w, 1.0);
    double a = 4.0 * (z * p);
    return inv ? (SBX_D_PIO2 - a) : a;
}

static inline double d_atan2(double y, double x) {
    if (x != x || y != y) return x + y;
    double ax = fabs(x), ay = fabs(y);
    if (ax == 0.0 && ay == 0.0) return 0.0;
    double a = d_atan_pos(ay / ax);
    if (x < 0.0) a = SBX_D_PI - a;
    if (y < 0.0) a = -a;
    return a;
}

/* GLSL atan(y, x) */
static inline float m_atan2(float y, float x) { return (float)d_atan2((double)y, (double)x); }

static inline float m_acos(float x) {
    if (!(x >= -1.0f && x <= 1.0f)) return u2f(0x7fc00000u);
    double xd = (double)x;
    return (float)d_atan2(sqrt((1.0 - xd) * (1.0 + xd)), xd);
}

#ifdef SBX_ORACLE_LIBM
/* Measurement-only variant (oracle/Makefile target libsbx_oracle_libm.so): route the
 * transcendental functions to glibc, i.e. what "VML + libm" would give, to quantify the
 * distance between the sbx math spec and a libm environment (tools/compare_libm.py). */
static inline float l_sin(float x) { return ::sinf(x); }
static inline float l_cos(float x) { return ::cosf(x); }
static inline float l_tan(float x) { return ::tanf(x); }
static inline float l_exp(float x) { return ::expf(x); }
static inline float l_pow(float x, float y) { return ::powf(x, y); }
static inline float l_acos(float x) { return ::acosf(x); }
static inline float l_atan2(float y, float x) { return ::atan2f(y, x); }
#define m_sin l_sin
#define m_cos l_cos
#define m_tan l_tan
#define m_exp l_exp
#define m_pow l_pow
#define m_acos l_acos
#define m_atan2 l_atan2
#endif

} /* namespace sbxref */
#endif
