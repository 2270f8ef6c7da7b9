// shaderbox_amd/csrc/sbx_math.h — the sbx math spec for the HIP kernels (and for the host code
// that precomputes per-frame constants), gfx950.
//
// The reference delegates sin/cos/exp/pow/acos/atan and the GLSL built-ins to its environment
// (/root/reference/src/def.h:1-42); its noise hash fract(sin(n)*753.5453123)
// (/root/reference/src/noise_iq.h:5-9) multiplies a sin ulp by 753, so GPU and CPU checker must
// share one definition.  DESIGN.md §3 is that definition: GLSL formulas for the built-ins, and
// transcendentals evaluated in binary64 by the fixed operation sequences below, rounded once to
// binary32 (= the correctly rounded value except within ~2^-28 ulp of a tie).  MI355X issues
// v_fma_f64 at ~0.6x the fp32 rate (profiles/r01_ubench_valu.txt), so "double inside" is cheap
// here, and every operation used (+ - * / sqrt fma, conversions) is IEEE-exact on both the
// device and the host, which makes results bit-identical without sharing code with the oracle.
//
// Build rules: -ffp-contract=off (hipcc contracts by default), no fast-math, fp32 denormals on,
// correctly rounded fp32 divide/sqrt (hipcc default).  fma appears only where written.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SBX_HD __host__ __device__ __forceinline__
#else
#define SBX_HD inline
#endif

namespace sbx {

SBX_HD uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
SBX_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
SBX_HD uint64_t d2u(double x) { return __builtin_bit_cast(uint64_t, x); }
SBX_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }

// ---- GLSL built-ins (GLSL spec formulas) -------------------------------
SBX_HD float fmin_(float a, float b) { return (b < a) ? b : a; }
SBX_HD float fmax_(float a, float b) { return (a < b) ? b : a; }
SBX_HD float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
SBX_HD float abs_(float x) { return u2f(f2u(x) & 0x7fffffffu); }
SBX_HD float floor_(float x) { return __builtin_floorf(x); }
SBX_HD float fract_(float x) { return x - __builtin_floorf(x); }
SBX_HD float mod_(float x, float y) { return x - y * __builtin_floorf(x / y); }
SBX_HD float mix_(float x, float y, float a) { return x * (1.0f - a) + y * a; }
SBX_HD float step_(float edge, float x) { return (x < edge) ? 0.0f : 1.0f; }
SBX_HD float smoothstep_(float e0, float e1, float x) {
    float t = clamp_((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return (t * t) * (3.0f - 2.0f * t);
}
SBX_HD float radians_(float deg) { return deg * 0.017453292519943295f; }

// Exact binary32 division by a denominator whose binary64 reciprocal is already known (a frame constant or
// a literal): n / d == (float)((double)n * RN64(1 / (double)d)) for ALL binary32 n, d, bit for bit with the
// IEEE quotient.  Proof sketch: the exact quotient Q of two 24-bit significands is never a rounding midpoint
// of binary32 (m * d with m an odd 25-bit integer has >= 25 significant bits) and, unless it is exactly
// representable, lies at relative distance >= 2^-49 from the nearest midpoint; the computed product carries a
// relative error < 2^-51.9 (one rounding in the reciprocal, one in the product), less than half that gap, so
// rounding it to binary32 gives RN32(Q).  Zeros, infinities, NaNs and denormals follow the same rules on
// both sides.  On MI355X this is v_cvt_f64_f32 + v_mul_f64 + v_cvt_f32_f64 (~13 issue cycles) against ~42
// for the v_div_scale/v_rcp/v_fma.../v_div_fixup sequence (profiles/r01_ubench_valu.txt).  The oracle keeps
// the plain division; tests/test_gpu_parity.py::test_division_by_reciprocal_is_exact checks the identity.
SBX_HD double recip64(float d) { return 1.0 / (double)d; }
SBX_HD float div_by(float n, double rd) { return (float)((double)n * rd); }
// smoothstep(e0, e1, x) with rd = recip64(e1 - e0)
SBX_HD float smoothstep_rd(float e0, double rd, float x) {
    float t = clamp_(div_by(x - e0, rd), 0.0f, 1.0f);
    return (t * t) * (3.0f - 2.0f * t);
}
SBX_HD float sqrt_(float x) { return __builtin_sqrtf(x); }

// ---- binary64 cores ------------------------------------------------------------------------
constexpr double D_INV_LN2 = 0x1.71547652b82fep+0;
constexpr double D_LN2 = 0x1.62e42fefa39efp-1;
constexpr double D_SQRT2 = 0x1.6a09e667f3bcdp+0;
constexpr double D_MAGIC = 6755399441055744.0;   // 1.5 * 2^52
constexpr double D_PI = 0x1.921fb54442d18p+1;
constexpr double D_PI_LO = 0x1.1a62633145c07p-53;
constexpr double D_PIO2 = 0x1.921fb54442d18p+0;
constexpr double D_INV_PI = 0x1.45f306dc9c883p-2;

// Polynomial coefficients of the binary64 cores.  On the device they are read from __constant__
// memory (scalar loads -> SGPR pairs, which v_fma_f64 takes directly as its addend): written as
// literals the compiler materialises every one of them in a VGPR pair and keeps ~40 VGPRs alive across
// the hot loops, which costs a wave of occupancy in k_clouds.  Host code reads the same list from a
// constexpr array; both arrays are initialised from the single list below, so the values are identical.
#define SBX_SIN_COEFS                                                                                  \
    0x1.71b8ef6dcf572p-66, -0x1.2f49b46814157p-57, 0x1.952c77030ad4ap-49, -0x1.ae7f3e733b81fp-41,     \
    0x1.6124613a86d09p-33, -0x1.ae64567f544e4p-26, 0x1.71de3a556c734p-19, -0x1.a01a01a01a01ap-13,      \
    0x1.1111111111111p-7, -0x1.5555555555555p-3
#define SBX_EXP_COEFS                                                                                  \
    0x1.6124613a86d09p-33, 0x1.1eed8eff8d898p-29, 0x1.ae64567f544e4p-26, 0x1.27e4fb7789f5cp-22,        \
    0x1.71de3a556c734p-19, 0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-13, 0x1.6c16c16c16c17p-10,        \
    0x1.1111111111111p-7, 0x1.5555555555555p-5, 0x1.5555555555555p-3, 0x1.0000000000000p-1
#if defined(__HIP_DEVICE_COMPILE__)
__constant__ const double kSinCoef[10] = {SBX_SIN_COEFS};
__constant__ const double kExpCoef[12] = {SBX_EXP_COEFS};
#else
constexpr double kSinCoef[10] = {SBX_SIN_COEFS};
constexpr double kExpCoef[12] = {SBX_EXP_COEFS};
#endif

// sin r, |r| <= pi/2: r + r^3 * (Taylor in r^2 up to r^21)
SBX_HD double d_sin_poly(double r) {
    double s = r * r;
    double p = kSinCoef[0];                           //  1/21!
    p = __builtin_fma(p, s, kSinCoef[1]);             // -1/19!
    p = __builtin_fma(p, s, kSinCoef[2]);             //  1/17!
    p = __builtin_fma(p, s, kSinCoef[3]);             // -1/15!
    p = __builtin_fma(p, s, kSinCoef[4]);             //  1/13!
    p = __builtin_fma(p, s, kSinCoef[5]);             // -1/11!
    p = __builtin_fma(p, s, kSinCoef[6]);             //  1/9!
    p = __builtin_fma(p, s, kSinCoef[7]);             // -1/7!
    p = __builtin_fma(p, s, kSinCoef[8]);             //  1/5!
    p = __builtin_fma(p, s, kSinCoef[9]);             // -1/3!
    return __builtin_fma(r * s, p, r);
}
SBX_HD double d_sin(double x) {
    double kd = __builtin_fma(x, D_INV_PI, D_MAGIC);
    uint64_t flip = d2u(kd) << 63;
    kd = kd - D_MAGIC;
    double r = __builtin_fma(kd, -D_PI, x);
    r = __builtin_fma(kd, -D_PI_LO, r);
    return u2d(d2u(d_sin_poly(r)) ^ flip);
}
SBX_HD double d_cos(double x) {
    double kd = __builtin_fma(x, D_INV_PI, 0.5) + D_MAGIC;
    uint64_t flip = d2u(kd) << 63;
    kd = kd - D_MAGIC;
    double m = kd - 0.5;
    double r = __builtin_fma(m, -D_PI, x);
    r = __builtin_fma(m, -D_PI_LO, r);
    return u2d(d2u(d_sin_poly(r)) ^ flip);
}
SBX_HD float sin_(float x) { return (float)d_sin((double)x); }
SBX_HD float cos_(float x) { return (float)d_cos((double)x); }
SBX_HD float tan_(float x) { return (float)(d_sin((double)x) / d_cos((double)x)); }

SBX_HD double d_log2(double x) {
    uint64_t b = d2u(x);
    int e = (int)(b >> 52) - 1023;
    double m = u2d((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > D_SQRT2) { m = m * 0.5; e = e + 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 0x1.8618618618618p-4;
    p = __builtin_fma(p, z, 0x1.af286bca1af28p-4);
    p = __builtin_fma(p, z, 0x1.e1e1e1e1e1e1ep-4);
    p = __builtin_fma(p, z, 0x1.1111111111111p-3);
    p = __builtin_fma(p, z, 0x1.3b13b13b13b14p-3);
    p = __builtin_fma(p, z, 0x1.745d1745d1746p-3);
    p = __builtin_fma(p, z, 0x1.c71c71c71c71cp-3);
    p = __builtin_fma(p, z, 0x1.2492492492492p-2);
    p = __builtin_fma(p, z, 0x1.999999999999ap-2);
    p = __builtin_fma(p, z, 0x1.5555555555555p-1);
    double lnm = __builtin_fma(s * z, p, 2.0 * s);
    return __builtin_fma(lnm, D_INV_LN2, (double)e);
}
SBX_HD double d_exp2(double t) {
    double kd = t + D_MAGIC;
    int32_t ki = (int32_t)(uint32_t)(d2u(kd) & 0xffffffffull);
    kd = kd - D_MAGIC;
    double u = (t - kd) * D_LN2;
    double p = kExpCoef[0];                           // 1/13!
    p = __builtin_fma(p, u, kExpCoef[1]);             // 1/12!
    p = __builtin_fma(p, u, kExpCoef[2]);
    p = __builtin_fma(p, u, kExpCoef[3]);
    p = __builtin_fma(p, u, kExpCoef[4]);
    p = __builtin_fma(p, u, kExpCoef[5]);
    p = __builtin_fma(p, u, kExpCoef[6]);
    p = __builtin_fma(p, u, kExpCoef[7]);
    p = __builtin_fma(p, u, kExpCoef[8]);
    p = __builtin_fma(p, u, kExpCoef[9]);
    p = __builtin_fma(p, u, kExpCoef[10]);            // 1/3!
    p = __builtin_fma(p, u, kExpCoef[11]);            // 1/2!
    p = __builtin_fma(p, u, 1.0);
    p = __builtin_fma(p, u, 1.0);
    double sc = u2d((uint64_t)(int64_t)(ki + 1023) << 52);
    return p * sc;
}
// The former definition of exp (13-term Taylor after a binary64 range reduction of x*log2e), kept as a test
// hook ("exp_h13"): the table form below gives the same binary32 result on all 2^32 inputs except
// x = -89.45233 (0xc2b2e798), where this one misrounds a denormal result by an ulp and the table form does not
// (tests/test_gpu_parity.py::test_exp_table_vs_horner).  pow still ends in d_exp2.
SBX_HD float exp_h13_(float x) {
    if (x != x) return x;
    double t = (double)x * D_INV_LN2;
    if (t < -160.0) t = -160.0;
    if (t > 136.0) t = 136.0;
    return (float)d_exp2(t);
}
// exp (oracle/sbx_math_ref.h m_exp): x = (k/32) ln2 + r, |r| <= ln2/64; exp(x) = 2^(k>>5) * 2^((k&31)/32) * exp(r)
// with k from the 1.5*2^52 trick, a two-term Cody-Waite r, the degree-6 Taylor polynomial of exp(r) and a table
// of the 32 correctly rounded 2^(j/32): 11 binary64 operations against 19 for the 13-term form (CLOUDS 4K
// 4.99 -> 4.64 ms, ATMOSPHERE 8K 6.06 -> 5.41 ms).  The table sits in __constant__ memory and is read per
// lane (one global_load_dwordx2 that hits in L1).  The binary32 clamp replaces the spec's NaN test: a NaN
// passes through clamp_ (comparisons false) and the arithmetic to a NaN result.
#define SBX_EXP2_TAB_VALUES                                                                                  \
    0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0,                  \
    0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0, 0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0,                  \
    0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0,                  \
    0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0,                  \
    0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0, 0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0,                  \
    0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0,                  \
    0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0,                  \
    0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0, 0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0
#if defined(__HIP_DEVICE_COMPILE__)
__constant__ const double kExp2Tab[32] = {SBX_EXP2_TAB_VALUES};
#else
constexpr double kExp2Tab[32] = {SBX_EXP2_TAB_VALUES};
#endif
SBX_HD float exp_(float x) {
    const double xd = (double)clamp_(x, -104.0f, 89.0f);
    double kd = __builtin_fma(xd, 0x1.71547652b82fep+5, D_MAGIC);          // 32/ln2
    const int32_t ki = (int32_t)(uint32_t)(d2u(kd) & 0xffffffffull);
    kd = kd - D_MAGIC;
    double r = __builtin_fma(kd, -0x1.62e42fefa0000p-6, xd);               // ln2/32, high 38 bits
    r = __builtin_fma(kd, -0x1.cf79abc9e3b3ap-45, r);                      // ln2/32 - high
    double p = 0x1.6c16c16c16c17p-10;                                      // 1/6!
    p = __builtin_fma(p, r, 0x1.1111111111111p-7);
    p = __builtin_fma(p, r, 0x1.5555555555555p-5);
    p = __builtin_fma(p, r, 0x1.5555555555555p-3);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double y = p * kExp2Tab[ki & 31];
    return (float)u2d(d2u(y) + ((uint64_t)(int64_t)(ki >> 5) << 52));
}
SBX_HD float pow_(float x, float y) {
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
SBX_HD double d_atan_pos(double z) {
    bool inv = z > 1.0;
    if (inv) z = 1.0 / z;
    z = z / (1.0 + __builtin_sqrt(__builtin_fma(z, z, 1.0)));
    z = z / (1.0 + __builtin_sqrt(__builtin_fma(z, z, 1.0)));
    double w = z * z;
    double p = -0x1.642c8590b2164p-5;
    p = __builtin_fma(p, w, 0x1.8618618618618p-5);
    p = __builtin_fma(p, w, -0x1.af286bca1af28p-5);
    p = __builtin_fma(p, w, 0x1.e1e1e1e1e1e1ep-5);
    p = __builtin_fma(p, w, -0x1.1111111111111p-4);
    p = __builtin_fma(p, w, 0x1.3b13b13b13b14p-4);
    p = __builtin_fma(p, w, -0x1.745d1745d1746p-4);
    p = __builtin_fma(p, w, 0x1.c71c71c71c71cp-4);
    p = __builtin_fma(p, w, -0x1.2492492492492p-3);
    p = __builtin_fma(p, w, 0x1.999999999999ap-3);
    p = __builtin_fma(p, w, -0x1.5555555555555p-2);
    p = __builtin_fma(p, w, 1.0);
    double a = 4.0 * (z * p);
    return inv ? (D_PIO2 - a) : a;
}
SBX_HD double d_atan2(double y, double x) {
    if (x != x || y != y) return x + y;
    double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
    if (ax == 0.0 && ay == 0.0) return 0.0;
    double a = d_atan_pos(ay / ax);
    if (x < 0.0) a = D_PI - a;
    if (y < 0.0) a = -a;
    return a;
}
SBX_HD float atan2_(float y, float x) { return (float)d_atan2((double)y, (double)x); }
SBX_HD float acos_(float x) {
    if (!(x >= -1.0f && x <= 1.0f)) return u2f(0x7fc00000u);
    double xd = (double)x;
    return (float)d_atan2(__builtin_sqrt((1.0 - xd) * (1.0 + xd)), xd);
}

}  // namespace sbx
