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
#if defined(SBX_ABLATE_MINMAX) && defined(__HIP_DEVICE_COMPILE__)     // timing experiment only: v_min_f32 / v_max_f32 (differ for NaN and for zeros of opposite sign)
SBX_HD float fmin_(float a, float b) { return __builtin_fminf(a, b); }
SBX_HD float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
#else
SBX_HD float fmin_(float a, float b) { return (b < a) ? b : a; }
SBX_HD float fmax_(float a, float b) { return (a < b) ? b : a; }
#endif
SBX_HD float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
SBX_HD float abs_(float x) { return u2f(f2u(x) & 0x7fffffffu); }
SBX_HD float floor_(float x) { return __builtin_floorf(x); }
SBX_HD float fract_(float x) { return x - __builtin_floorf(x); }
SBX_HD float mod_(float x, float y) { return x - y * __builtin_floorf(x / y); }
SBX_HD float mix_(float x, float y, float a) { return x * (1.0f - a) + y * a; }
// 3 - 2a, the second factor of every smoothstep weight a*a*(3 - 2a), in ONE instruction: 2a is exact in binary32 (a power-of-two
// scale; the weights' arguments are fracts or clamped to [0, 1]; beyond 2^127 both forms overflow to the same infinity), so the
// reference's RN(3 - RN(2a)) equals RN(3 - 2a), which is what fma(-2, a, 3) returns.  Same bits, one VALU instruction less per weight
// (round 3; k_clouds 2.745 -> 2.6 ms together with the pre-scaled light-march blends).  Written as an explicit fma: the build
// never contracts on its own (-ffp-contract=off).
SBX_HD float tm2_(float a) { return __builtin_fmaf(-2.0f, a, 3.0f); }
SBX_HD float step_(float edge, float x) { return (x < edge) ? 0.0f : 1.0f; }
SBX_HD float smoothstep_(float e0, float e1, float x) {
    float t = clamp_((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return (t * t) * tm2_(t);
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
    return (t * t) * tm2_(t);
}
// Division by a KNOWN divisor in three binary32 instructions: with r = RN(1 / d) computed once,
//     q0 = RN(a * r);   e = fma(-q0, d, a)  (the exact remainder);   q = fma(e, r, q0)
// is EQUAL to the IEEE quotient RN(a / d) for every pair of binary32 significands — all 2^47 pairs were run on the GPU
// (tools/div3_exhaustive.hip, profiles/r03_div3_exhaustive.txt: 0 divisors of 8 388 608 have a failing dividend; division is
// scale-free away from overflow and underflow).  6.75 issue cycles on gfx950 against 12.6 for div_by's cvt + binary64 multiply + cvt
// (and ~40 for the IEEE expansion).  Valid where nothing over- or underflows on the way and the sign of a zero quotient is not used:
// callers show a finite with 2^-100 <= |a| <= 2^100 or a == 0 (a zero dividend of either sign gives +0), and 2^-60 <= |d| <= 2^60.
// A NaN dividend gives NaN.
struct Div3 { float d, r; };
SBX_HD Div3 make_div3(float d) { return Div3{d, 1.0f / d}; }
SBX_HD float div3_(float a, float d, float r) {
    const float q0 = a * r;
    const float e = __builtin_fmaf(-q0, d, a);
    return __builtin_fmaf(e, r, q0);
}
// The same for a divisor that is NOT known in advance: v_rcp_f32 (1 ulp) made exact enough by one Newton step, then div3_'s three
// instructions — six in all (~20 issue cycles against ~36 for the compiler's IEEE expansion with its two v_div_scale, v_div_fmas and
// v_div_fixup).  EQUAL to the IEEE quotient for every pair of binary32 significands (tools/divv_exhaustive.hip, all 2^47 pairs on
// the GPU, profiles/r03_divv_exhaustive.txt; WITHOUT the Newton step 47 024 failures are reported).  Same domain as div3_, and the
// divisor must be finite and non-zero: callers show 2^-60 <= |b| <= 2^60 and a finite, zero or within 2^+-100.
SBX_HD float divn_(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r0, 1.0f);
    const float r = __builtin_fmaf(e, r0, r0);
    return div3_(a, b, r);
#else
    return a / b;
#endif
}
// smoothstep(e0, e0 + d, x) through div3_ (the clamp makes the sign of a zero quotient irrelevant: t * t is +0 either way)
SBX_HD float smoothstep_d3(float e0, float d, float r, float x) {
    const float t = clamp_(div3_(x - e0, d, r), 0.0f, 1.0f);
    return (t * t) * tm2_(t);
}
#if defined(__HIPCC__)
// smoothstep_d3 with the clamp as ONE v_med3_f32: equal for every FINITE x (a -0 quotient squares to +0 either way); a NaN x gives 0
// here and NaN there, so only for callers whose x cannot be a NaN
__device__ __forceinline__ float smoothstep_d3_med3(float e0, float d, float r, float x) {
    const float t = __builtin_amdgcn_fmed3f(div3_(x - e0, d, r), 0.0f, 1.0f);
    return (t * t) * tm2_(t);
}
__device__ __forceinline__ float x_smoothstep_d3_med3(float e0, float d, float r, float x) {      // x_smoothstep_rd_med3 with div3_
    const float t = __builtin_amdgcn_fmed3f(div3_(x - e0, d, r), 0.0f, 1.0f);
    return x * ((t * t) * tm2_(t));
}
// x * smoothstep(e0, e1, x) with the clamp done by ONE v_med3_f32 (half the issue cost of two compare/select pairs).
// v_med3_f32(t, 0, 1) differs from clamp_(t, 0, 1) in two cases only, and neither survives the operations around it:
//   t = -0 : clamp_ gives -0, med3 may give +0 — the next operation is t * t = +0 either way;
//   t NaN  : clamp_ gives NaN, med3 gives 0 — with e0 and rd finite t is NaN only if x is, and the result x * (...)
//            is NaN either way.
// Callers must have checked that e0 and rd are finite (kern_clouds.hip: the REG kernels).
__device__ __forceinline__ float x_smoothstep_rd_med3(float e0, double rd, float x) {
    const float t = __builtin_amdgcn_fmed3f(div_by(x - e0, rd), 0.0f, 1.0f);
    return x * ((t * t) * tm2_(t));
}
#endif
SBX_HD float sqrt_(float x) { return __builtin_sqrtf(x); }   // IEEE binary32, correctly rounded (the compiler's expansion on the device)
// The same function for callers that can show |x| >= 2^-96, x == 0, or x not finite — never a non-zero number of magnitude
// below 2^-96: v_sqrt_f32 (within 1 ulp) and the two-sided fix-up of the compiler's own expansion (the neighbours y -+ 1 ulp tested
// through exact fma residuals) WITHOUT that expansion's input scaling and class tests, which exist for arguments below 2^-96
// only.  ~35 instead of ~60 issue cycles.  Bit-identical to __builtin_sqrtf on every input with |x| outside (0, 2^-96)
// (tests/test_gpu_round2.py::test_sqrt_n_is_ieee_sqrt runs all 2^32).  As a general replacement behind a per-lane branch it
// LOST time in the SDF kernels (VINYL +20 %: the branch splits the straight-line SDF code), so only APP_ATMOSPHERE, whose
// operands are ~4e13 or differences of such (multiples of 4e6, or zero), uses it.
SBX_HD float sqrt_n_(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_sqrtf(x);
    const float ym = u2f(f2u(y) - 1u), yp = u2f(f2u(y) + 1u);
    const float tm = __builtin_fmaf(-ym, y, x);               // x - (y - 1 ulp) y, exact sign
    const float tp = __builtin_fmaf(-yp, y, x);               // x - (y + 1 ulp) y
    float r = (tm <= 0.0f) ? ym : y;
    r = (tp > 0.0f) ? yp : r;
    return r;
#else
    return __builtin_sqrtf(x);
#endif
}
SBX_HD float sqrt_ieee_(float x) { return __builtin_sqrtf(x); }
// The square root in FIVE instructions for callers that can show 2^-100 <= x < inf (or x NaN / negative: NaN either way) — never 0,
// never +inf, never a tiny number:  rs = v_rsq_f32(x) (1 ulp);  y0 = x rs;  h = rs / 2;  r = fma(-y0, y0, x) (the exact residual);
// y = fma(r, h, y0).  EQUAL to the IEEE square root for every binary32 x >= 2^-102: all 2^31 positive arguments were run on the GPU
// (tools/sqrt_rsq_exhaustive.hip, profiles/r03_sqrt_rsq_exhaustive.txt: the only differences are below 2^-102, where the residual
// underflows; 0 and +inf give NaN).  ~17 issue cycles against ~33 for sqrt_n_ and ~60 for the compiler's expansion.
SBX_HD float sqrt_rs_(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float rs = __builtin_amdgcn_rsqf(x);
    const float y0 = x * rs;
    const float h = .5f * rs;
    const float r = __builtin_fmaf(-y0, y0, x);
    return __builtin_fmaf(r, h, y0);
#else
    return __builtin_sqrtf(x);
#endif
}

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
// sin with a degree-15 minimax polynomial (Remez on [0, pi/2], relative error 2^-58; tools/gen_math_coeffs.py --sin15) in place of
// the spec's Taylor polynomial to r^21: the same argument reduction, three binary64 fma less.  A kernel-internal form like the exp
// forms below, admitted on the same ground: EQUAL to sin_ on every binary32 argument with |x| < 2^44.6 — all 2.9e9 of them were run
// (the first difference is at x = 0x1.95f654p+44, where the two-term reduction has long stopped delivering an r in [-pi/2, pi/2]);
// callers use it for |x| <= 2^40 (host: tests/test_exp_small.py against the oracle's m_sin; GPU: tests/test_gpu_round3.py).
// cos_ does NOT have this property (it differs at x = 0x1.8f219cp+5) and keeps the spec's polynomial.
constexpr float SIN_B40_MAX = 0x1p+40f;
#define SBX_SIN15_COEFS -0x1.a09498de90541p-41, 0x1.60f3b0ed85645p-33, -0x1.ae63ac7b0bc8cp-26, 0x1.71de3921ccf1fp-19, \
                         -0x1.a01a019dfb495p-13, 0x1.111111110f8bcp-7, -0x1.5555555555543p-3       /* r^15 ... r^3 */
#if defined(__HIP_DEVICE_COMPILE__)
__constant__ const double kSin15Coef[7] = {SBX_SIN15_COEFS};
#else
constexpr double kSin15Coef[7] = {SBX_SIN15_COEFS};
#endif
SBX_HD float sin_b40_(float x) {
    const double xd = (double)x;
    double kd = __builtin_fma(xd, D_INV_PI, D_MAGIC);
    const uint64_t flip = d2u(kd) << 63;
    kd = kd - D_MAGIC;
    double r = __builtin_fma(kd, -D_PI, xd);
    r = __builtin_fma(kd, -D_PI_LO, r);
    const double s = r * r;
    double p = kSin15Coef[0];                         // (read like kSinCoef: scalar loads at the use, no literals held in registers)
    p = __builtin_fma(p, s, kSin15Coef[1]);
    p = __builtin_fma(p, s, kSin15Coef[2]);
    p = __builtin_fma(p, s, kSin15Coef[3]);
    p = __builtin_fma(p, s, kSin15Coef[4]);
    p = __builtin_fma(p, s, kSin15Coef[5]);
    p = __builtin_fma(p, s, kSin15Coef[6]);
    return (float)u2d(d2u(__builtin_fma(r * s, p, r)) ^ flip);
}
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
// 2^(j/64), j = 0..63, correctly rounded to binary64 (tools/gen_math_coeffs.py --exp-table 64): the table of exp_reg64_
#define SBX_EXP2_TAB64_VALUES \
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,  \
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,  \
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,  \
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,  \
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,  \
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,  \
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,  \
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,  \
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,  \
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,  \
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,  \
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,  \
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,  \
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,  \
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,  \
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0
#if defined(__HIP_DEVICE_COMPILE__)
__constant__ const double kExp2Tab64[64] = {SBX_EXP2_TAB64_VALUES};
#else
constexpr double kExp2Tab64[64] = {SBX_EXP2_TAB64_VALUES};
#endif
// exp with the 2^(j/32) table read through `tab` (the __constant__ table, or a copy a kernel keeps in LDS: the per-lane
// table read is a dependent memory access in every call, ~4x shorter from LDS than from the vector L1)
// CLAMP = false leaves the binary32 range guard out: only for callers that have shown |x| <= 89 (or x NaN, which the
// guard passes through unchanged anyway) for every argument they can produce — then the guard is the identity.
template <bool CLAMP = true, class Tab>
SBX_HD float exp_tab_(float x, const Tab& tab) {
    const double xd = (double)(CLAMP ? clamp_(x, -104.0f, 89.0f) : x);
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
    const double y = p * tab[ki & 31];
    return (float)u2d(d2u(y) + ((uint64_t)(int64_t)(ki >> 5) << 52));
}
SBX_HD float exp_(float x) { return exp_tab_(x, kExp2Tab); }
#if defined(__HIPCC__)
// exp_tab_<false> for callers that have shown |x| <= 80 (or NaN) for every argument they can produce — the regular-frame
// k_clouds (|sigma * dt| <= 80 checked per launch) and APP_ATMOSPHERE's density terms (-height / H in [-50.1, 0.001]) — in
// the cheapest instruction sequence found on gfx950, same operations on the same operands:
//  * the middle of the Horner chain as three-address v_fma_f64: the compiler turns `p = fma(p, r, c)` with c in a VGPR pair
//    into v_mov_b64 (copy c) + v_fmac_f64 (two-address), one extra half-rate instruction per coefficient;
//  * the scaling by 2^(k >> 5) AFTER the rounding to binary32 (v_cvt + v_ashr + v_ldexp_f32 instead of shift, mask and a
//    64-bit add before the v_cvt): y * 2^e lies in [2^-117, 2^117], far inside binary32's normal range, where a
//    power-of-two scale commutes with the rounding; NaN stays NaN.
// tests/test_gpu_round2.py::test_regular_frame_exp_equals_exp_on_its_whole_domain compares it with exp_ on EVERY binary32
// argument with |x| <= 80 (sbx_math_eval "exp_reg"): identical.
__device__ __forceinline__ double fma64_3addr(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// ASM = false keeps the compiler's own Horner chain (APP_ATMOSPHERE: with its two unrolled sample loops the asm operands in VGPR
// pairs raise the kernel from 54 to 128 VGPRs and 4.53 to 4.72 ms).
template <bool ASM = true, class Tab>
__device__ __forceinline__ float exp_reg_(float x, const Tab& tab) {
    const double xd = (double)x;
    double kd = __builtin_fma(xd, 0x1.71547652b82fep+5, D_MAGIC);          // 32/ln2
    const int32_t ki = (int32_t)(uint32_t)(d2u(kd) & 0xffffffffull);
    kd = kd - D_MAGIC;
    double r = __builtin_fma(kd, -0x1.62e42fefa0000p-6, xd);               // ln2/32, high 38 bits
    r = __builtin_fma(kd, -0x1.cf79abc9e3b3ap-45, r);                      // ln2/32 - high
    double p, c6 = 0x1.6c16c16c16c17p-10, c5 = 0x1.1111111111111p-7;         // 1/6!, 1/5!
    const double c4 = 0x1.5555555555555p-5, c3 = 0x1.5555555555555p-3;
    if (ASM) {
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "s"(c6), "v"(r), "v"(c5));
        p = fma64_3addr(p, r, c4);
        p = fma64_3addr(p, r, c3);
    } else {
        p = __builtin_fma(c6, r, c5);
        p = __builtin_fma(p, r, c4);
        p = __builtin_fma(p, r, c3);
    }
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double y = p * tab[ki & 31];
    return __builtin_ldexpf((float)y, ki >> 5);
}
// The same with a 64-entry table: x = (k/64) ln2 + r, |r| <= ln2/128, and the degree-5 Taylor polynomial: ONE binary64 fma less
// per call.  Not the spec (its truncation error r^6/720 <= 3.5e-17 is 10x the degree-6 / 32-entry form's): a kernel-internal form,
// admitted because tests/test_gpu_round3.py::test_exp_reg64_equals_exp_on_its_whole_domain finds it equal to exp_ on EVERY
// binary32 argument with |x| <= 80 (2.2e9 values; so is the 128-entry form, which costs k_clouds a wave of occupancy in LDS).
// `tab` = kExp2Tab64 or a copy in LDS (512 B).
template <bool ASM = true, class Tab>
__device__ __forceinline__ float exp_reg64_(float x, const Tab& tab) {
    const double xd = (double)x;
    double kd = __builtin_fma(xd, 0x1.71547652b82fep+6, D_MAGIC);          // 64/ln2
    const int32_t ki = (int32_t)(uint32_t)(d2u(kd) & 0xffffffffull);
    kd = kd - D_MAGIC;
    double r = __builtin_fma(kd, -0x1.62e42fefa0000p-7, xd);               // ln2/64, high 38 bits (|k| < 2^13: k * hi exact)
    r = __builtin_fma(kd, -0x1.cf79abc9e3b3ap-46, r);                      // ln2/64 - high
    double p, c5 = 0x1.1111111111111p-7;                                     // 1/5!
    const double c4 = 0x1.5555555555555p-5, c3 = 0x1.5555555555555p-3;       // 1/4!, 1/3!
    if (ASM) {
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(p) : "s"(c5), "v"(r), "v"(c4));
        p = fma64_3addr(p, r, c3);
    } else {
        p = __builtin_fma(c5, r, c4);
        p = __builtin_fma(p, r, c3);
    }
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double y = p * tab[ki & 63];
    return __builtin_ldexpf((float)y, ki >> 6);
}
// The same with a 4096-entry table (sbx_exp4k_table.h, 32 KB, read from global memory through the vector L1): x = (k/4096) ln2 + r,
// |r| <= ln2/8192, and the degree-3 Taylor polynomial (truncation r^4/24 <= 2.1e-18): TWO binary64 fma less than exp_reg64_ and no
// coefficient to move into a register pair.  A kernel-internal form like exp_reg64_, admitted on the same ground: equal to exp_ on
// EVERY binary32 argument in [-80, 2^18] (2.3e9 values; beyond 88.7 all three overflow to +inf; host: tests/test_exp_small.py against
// the oracle's m_exp; GPU:
// tests/test_gpu_round3.py::test_exp_reg64_equals_exp_on_its_whole_domain).  (2048 entries: ONE argument of the 2.2e9 differs.)
template <class Tab>
__device__ __forceinline__ float exp_reg4k_(float x, const Tab& tab) {
    const double xd = (double)x;
    double kd = __builtin_fma(xd, 0x1.71547652b82fep+12, D_MAGIC);         // 4096/ln2
    const int32_t ki = (int32_t)(uint32_t)(d2u(kd) & 0xffffffffull);
    kd = kd - D_MAGIC;
    double r = __builtin_fma(kd, -0x1.62e42fee00000p-13, xd);              // ln2/4096, high 32 bits (|k| < 2^19: k * hi exact)
    r = __builtin_fma(kd, -0x1.a39ef35793c76p-45, r);                      // ln2/4096 - high
    double p = __builtin_fma(0x1.5555555555555p-3, r, 0.5);                // 1/3!
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double y = p * tab[ki & 4095];
    return __builtin_ldexpf((float)y, ki >> 12);
}
// exp on the SMALL NEGATIVE domain [-0.205, -0] (and +0): NO argument reduction, no table — the degree-8 minimax polynomial of
// e^x on that interval (relative error 2^-55.6 with these binary64 coefficients, tools/gen_math_coeffs.py --exp-small), one Horner
// chain: cvt + 8 fma + cvt = 10 half-rate instructions against the 17 of exp_reg64_ (magic-number rounding, two-step reduction,
// degree 5, table read, scaling).  Not the spec: a kernel-internal form, admitted because it is EQUAL to exp_ on EVERY binary32
// argument of its domain — 1 045 556 103 values, checked on the host against the oracle's m_exp (tests/test_exp_small.py, g++)
// and on the GPU against exp_ (tests/test_gpu_round3.py::test_exp_small_equals_exp_on_its_whole_domain).  The callers show
// -0.205 <= x <= 0 for every argument they can produce (APP_CLOUDS: x = -density * sigma * dt with 0 <= density <= .9375 * (1 + 1e-6)
// and the host's check .94 * sigma * dt <= .205, sigma >= 0, dt >= 0 — the default frame's .1758); NaN stays NaN.
constexpr float EXP_SMALL_MIN = -0.205f;
template <bool ASM = true>
__device__ __forceinline__ float exp_small_(float x) {
    const double xd = (double)x;
    const double c8 = 0x1.77c3d007a0225p-16, c7 = 0x1.9e349217bfc91p-13, c6 = 0x1.6c0a5d6258a41p-10, c5 = 0x1.1110e1e19112dp-7,
                 c4 = 0x1.5555548342347p-5, c3 = 0x1.555555534e819p-3, c2 = 0x1.fffffffffb251p-2, c1 = 0x1.fffffffffffc0p-1;
    double p;
    if (ASM) {
        // three-address v_fma_f64 with the coefficient as the one scalar operand (the compiler's own chain copies every coefficient
        // into a VGPR pair first: v_mov_b64 + two-address v_fmac_f64), in ONE asm statement (after each separate one the compiler
        // adds an s_nop)
        asm("v_fma_f64 %0, %1, %2, %3\n\t"
            "v_fma_f64 %0, %0, %2, %4\n\t"
            "v_fma_f64 %0, %0, %2, %5\n\t"
            "v_fma_f64 %0, %0, %2, %6\n\t"
            "v_fma_f64 %0, %0, %2, %7\n\t"
            "v_fma_f64 %0, %0, %2, %8\n\t"
            "v_fma_f64 %0, %0, %2, %9"
            : "=&v"(p) : "v"(c8), "v"(xd), "s"(c7), "s"(c6), "s"(c5), "s"(c4), "s"(c3), "s"(c2), "s"(c1));
    } else {
        p = __builtin_fma(c8, xd, c7);
        p = __builtin_fma(p, xd, c6);
        p = __builtin_fma(p, xd, c5);
        p = __builtin_fma(p, xd, c4);
        p = __builtin_fma(p, xd, c3);
        p = __builtin_fma(p, xd, c2);
        p = __builtin_fma(p, xd, c1);
    }
    p = __builtin_fma(p, xd, 1.0);
    return (float)p;
}
#endif
// The former pow (atanh-series log2 with a binary64 division, 13-term 2^t), kept as the test hook "pow_h": the table
// form below agrees with it except on a ~1e-8 fraction of inputs that sit on a binary32 rounding boundary to within
// the ~1e-16 accuracy of either binary64 value (tests/test_gpu_parity.py::test_pow_table_vs_series).
SBX_HD float pow_h_(float x, float y) {
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
// pow (oracle/sbx_math_ref.h m_pow): 2^(y log2 x) through two table forms, 23 binary64 operations against 45
// (a quarter of APP_RAYTRACER's and 4 % of APP_CLOUDS' time is linear_to_srgb's three pow).
//   log2 x: x = 2^k z, z in [0.6875, 1.375), 128 intervals of z with centre c_i (the two next to 1.0 use c = 1, which
//           keeps the relative accuracy near x = 1): k + logc_i + log2(1 + r), r = z * invc_i - 1 (one fma), |r| < 2^-7,
//           log2(1 + r) = r (A0 + ... + A7 r^7);
//   2^t   : t = k/32 + r exactly, 2^(k>>5) * 2^((k&31)/32) * exp(r ln2) with the degree-6 Taylor polynomial and the
//           table of exp_.
// Constants: tools/gen_math_coeffs.py --log2-table / --exp-table.
#define SBX_LOG2_TAB_VALUES \
    {0x1.734f0c541fe8dp+0, -0x1.12aceeefcd824p-1}, {0x1.713786d9c7c09p+0, -0x1.0e809617b46b5p-1}, \
    {0x1.6f26016f26017p+0, -0x1.0a5a3dc175219p-1}, {0x1.6d1a62681c861p+0, -0x1.0639d4c219d60p-1}, \
    {0x1.6b1490aa31a3dp+0, -0x1.021f4a37ecbfbp-1}, {0x1.691473a88d0c0p+0, -0x1.fc151b11b3641p-2}, \
    {0x1.6719f3601671ap+0, -0x1.f3f71cc1b629cp-2}, {0x1.6524f853b4aa3p+0, -0x1.ebe47960e3c08p-2}, \
    {0x1.63356b88ac0dep+0, -0x1.e3dd1156507dep-2}, {0x1.614b36831ae94p+0, -0x1.dbe0c58c3cff3p-2}, \
    {0x1.5f66434292dfcp+0, -0x1.d3ef776d43ff4p-2}, {0x1.5d867c3ece2a5p+0, -0x1.cc0908e19b7bcp-2}, \
    {0x1.5babcc647fa91p+0, -0x1.c42d5c4c688b2p-2}, {0x1.59d61f123ccaap+0, -0x1.bc5c5489254cbp-2}, \
    {0x1.5805601580560p+0, -0x1.b495d4e9185f7p-2}, {0x1.56397ba7c52e2p+0, -0x1.acd9c130dd540p-2}, \
    {0x1.54725e6bb82fep+0, -0x1.a527fd95fd8ffp-2}, {0x1.52aff56a8054bp+0, -0x1.9d806ebc9921dp-2}, \
    {0x1.50f22e111c4c5p+0, -0x1.95e2f9b51f04cp-2}, {0x1.4f38f62dd4c9bp+0, -0x1.8e4f83fa145f0p-2}, \
    {0x1.4d843bedc2c4cp+0, -0x1.86c5f36dea3dep-2}, {0x1.4bd3edda68fe1p+0, -0x1.7f462e58e1689p-2}, \
    {0x1.4a27fad76014ap+0, -0x1.77d01b66fbd36p-2}, {0x1.4880522014880p+0, -0x1.7063a1a5fb4f1p-2}, \
    {0x1.46dce34596066p+0, -0x1.6900a8836d0d4p-2}, {0x1.453d9e2c776cap+0, -0x1.61a717cac1983p-2}, \
    {0x1.43a2730abee4dp+0, -0x1.5a56d7a370dedp-2}, {0x1.420b5265e5951p+0, -0x1.530fd08f29fa6p-2}, \
    {0x1.40782d10e6566p+0, -0x1.4bd1eb680e548p-2}, {0x1.3ee8f42a5af07p+0, -0x1.449d115ef7d88p-2}, \
    {0x1.3d5d991aa75c6p+0, -0x1.3d712bf9c9df0p-2}, {0x1.3bd60d9232955p+0, -0x1.364e2511cc823p-2}, \
    {0x1.3a524387ac822p+0, -0x1.2f33e6d2120f0p-2}, {0x1.38d22d366088ep+0, -0x1.28225bb5e64a5p-2}, \
    {0x1.3755bd1c945eep+0, -0x1.21196e87473d2p-2}, {0x1.35dce5f9f2af8p+0, -0x1.1a190a5d674a0p-2}, \
    {0x1.34679ace01346p+0, -0x1.13211a9b38422p-2}, {0x1.32f5ced6a1dfap+0, -0x1.0c318aedff3c0p-2}, \
    {0x1.3187758e9ebb6p+0, -0x1.054a474bf0eb7p-2}, {0x1.301c82ac40260p+0, -0x1.fcd677e5ac81bp-3}, \
    {0x1.2eb4ea1fed14bp+0, -0x1.ef28aacd72230p-3}, {0x1.2d50a012d50a0p+0, -0x1.e18b00e13123cp-3}, \
    {0x1.2bef98e5a3711p+0, -0x1.d3fd543a4ad5dp-3}, {0x1.2a91c92f3c105p+0, -0x1.c67f7f770a67bp-3}, \
    {0x1.293725bb804a5p+0, -0x1.b9115db83a3dfp-3}, {0x1.27dfa38a1ce4dp+0, -0x1.abb2ca9ec746ep-3}, \
    {0x1.268b37cd60127p+0, -0x1.9e63a24971f4ap-3}, {0x1.2539d7e9177b2p+0, -0x1.9123c1528c6cdp-3}, \
    {0x1.23eb79717605bp+0, -0x1.83f304cdc5aa4p-3}, {0x1.22a0122a0122ap+0, -0x1.76d14a4601225p-3}, \
    {0x1.21579804855e6p+0, -0x1.69be6fbb3aa6fp-3}, {0x1.2012012012012p+0, -0x1.5cba53a0762edp-3}, \
    {0x1.1ecf43c7fb84cp+0, -0x1.4fc4d4d9bb311p-3}, {0x1.1d8f5672e4abdp+0, -0x1.42ddd2ba1b4aep-3}, \
    {0x1.1c522fc1ce059p+0, -0x1.36052d01c3dd8p-3}, {0x1.1b17c67f2bae3p+0, -0x1.293ac3dc1a66cp-3}, \
    {0x1.19e0119e0119ep+0, -0x1.1c7e77dde33dbp-3}, {0x1.18ab083902bdbp+0, -0x1.0fd02a03727edp-3}, \
    {0x1.1778a191bd684p+0, -0x1.032fbbaee6d64p-3}, {0x1.1648d50fc3201p+0, -0x1.ed3a1d4cdbeb9p-4}, \
    {0x1.151b9a3fdd5c9p+0, -0x1.d4300a2524d46p-4}, {0x1.13f0e8d344724p+0, -0x1.bb4102f925391p-4}, \
    {0x1.12c8b89edc0acp+0, -0x1.a26ccd9981858p-4}, {0x1.11a3019a74826p+0, -0x1.89b33091d6fdep-4}, \
    {0x1.107fbbe011080p+0, -0x1.7113f3259e07fp-4}, {0x1.0f5edfab325a2p+0, -0x1.588edd4d1ceb2p-4}, \
    {0x1.0e40655826011p+0, -0x1.4023b7b26aca0p-4}, {0x1.0d24456359e3ap+0, -0x1.27d24bae824dfp-4}, \
    {0x1.0c0a7868b4171p+0, -0x1.0f9a634663ae7p-4}, {0x1.0af2f722eecb5p+0, -0x1.eef792508b68ap-5}, \
    {0x1.09ddba6af8360p+0, -0x1.beec9151aac2bp-5}, {0x1.08cabb37565e2p+0, -0x1.8f135b8107911p-5}, \
    {0x1.07b9f29b8eae2p+0, -0x1.5f6b8a11c3c73p-5}, {0x1.06ab59c7912fbp+0, -0x1.2ff4b77413db9p-5}, \
    {0x1.059eea0727586p+0, -0x1.00ae7f502c1b2p-5}, {0x1.04949cc1664c5p+0, -0x1.a330fd028f734p-6}, \
    {0x1.038c6b78247fcp+0, -0x1.4564a62192839p-6}, {0x1.02864fc7729e9p+0, -0x1.cfee70c5ce606p-7}, \
    {0x1.0182436517a37p+0, -0x1.15cfe8eaec7f5p-7}, {0x1.0000000000000p+0, 0x0.0p+0}, \
    {0x1.0000000000000p+0, 0x0.0p+0}, {0x1.fa11caa01fa12p-1, 0x1.1363117a97b03p-6}, \
    {0x1.f6310aca0dbb5p-1, 0x1.c9363ba850f9cp-6}, {0x1.f25f644230ab5p-1, 0x1.3ed3094685a27p-5}, \
    {0x1.ee9c7f8458e02p-1, 0x1.985bfc3495193p-5}, {0x1.eae807aba01ebp-1, 0x1.f13898332539dp-5}, \
    {0x1.e741aa59750e4p-1, 0x1.24b5b7e135a41p-4}, {0x1.e3a9179dc1a73p-1, 0x1.507b836033bbap-4}, \
    {0x1.e01e01e01e01ep-1, 0x1.7beee96b8a281p-4}, {0x1.dca01dca01dcap-1, 0x1.a7111df348494p-4}, \
    {0x1.d92f2231e7f8ap-1, 0x1.d1e34e35b82d7p-4}, {0x1.d5cac807572b2p-1, 0x1.fc66a0f0b00a5p-4}, \
    {0x1.d272ca3fc5b1ap-1, 0x1.134e1b4890631p-3}, {0x1.cf26e5c44bfc6p-1, 0x1.284294b07a640p-3}, \
    {0x1.cbe6d9601cbe7p-1, 0x1.3d1146d9a8a63p-3}, {0x1.c8b265afb8a42p-1, 0x1.51bab907a5c8ap-3}, \
    {0x1.c5894d10d4986p-1, 0x1.663f6fac91315p-3}, {0x1.c26b5392ea01cp-1, 0x1.7a9fec7d05de0p-3}, \
    {0x1.bf583ee868d8bp-1, 0x1.8edcae8352b6bp-3}, {0x1.bc4fd65883e7bp-1, 0x1.a2f632320b86cp-3}, \
    {0x1.b951e2b18ff23p-1, 0x1.b6ecf175f95ecp-3}, {0x1.b65e2e3beee05p-1, 0x1.cac163c770dcap-3}, \
    {0x1.b37484ad806cep-1, 0x1.de73fe3b1480ep-3}, {0x1.b094b31d922a4p-1, 0x1.f205339208f27p-3}, \
    {0x1.adbe87f94905ep-1, 0x1.02baba24d0664p-2}, {0x1.aaf1d2f87ebfdp-1, 0x1.0c62975542a8dp-2}, \
    {0x1.a82e65130e159p-1, 0x1.15fa676bb08fep-2}, {0x1.a574107688a4ap-1, 0x1.1f825f6d88e13p-2}, \
    {0x1.a2c2a87c51ca0p-1, 0x1.28fab35b32684p-2}, {0x1.a01a01a01a01ap-1, 0x1.32639636b2836p-2}, \
    {0x1.9d79f176b682dp-1, 0x1.3bbd3a0a1dcfbp-2}, {0x1.9ae24ea5510dap-1, 0x1.4507cfedd4fc5p-2}, \
    {0x1.9852f0d8ec0ffp-1, 0x1.4e43880e8fb6bp-2}, {0x1.95cbb0be377aep-1, 0x1.577091b3378c9p-2}, \
    {0x1.934c67f9b2ce6p-1, 0x1.608f1b42948aep-2}, {0x1.90d4f120190d5p-1, 0x1.699f5248cd4b8p-2}, \
    {0x1.8e6527af1373fp-1, 0x1.72a1637cbc183p-2}, {0x1.8bfce8062ff3ap-1, 0x1.7b957ac51aac4p-2}, \
    {0x1.899c0f601899cp-1, 0x1.847bc33d8618ep-2}, {0x1.87427bcc092b9p-1, 0x1.8d54673b5c371p-2}, \
    {0x1.84f00c2780614p-1, 0x1.961f90527409bp-2}, {0x1.82a4a0182a4a0p-1, 0x1.9edd6759b25e0p-2}, \
    {0x1.8060180601806p-1, 0x1.a78e146f7bef4p-2}, {0x1.7e225515a4f1dp-1, 0x1.b031befe06435p-2}, \
    {0x1.7beb3922e017cp-1, 0x1.b8c88dbf88679p-2}, {0x1.79baa6bb6398bp-1, 0x1.c152a6c24cae7p-2}, \
    {0x1.77908119ac60dp-1, 0x1.c9d02f6ca47b5p-2}, {0x1.756cac201756dp-1, 0x1.d2414c80bf27cp-2},
#if defined(__HIP_DEVICE_COMPILE__)
__constant__ const double kLog2Tab[128][2] = {SBX_LOG2_TAB_VALUES};
#else
constexpr double kLog2Tab[128][2] = {SBX_LOG2_TAB_VALUES};
#endif
SBX_HD double d_log2_tab(double x) {
    const uint64_t ix = d2u(x);
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((int64_t)tmp >> 52);
    const double z = u2d(ix - (tmp & 0xfff0000000000000ull));
    const double r = __builtin_fma(z, kLog2Tab[i][0], -1.0);
    double p = -0x1.71547652b82fep-3;                          // A7 = -1/(8 ln2)
    p = __builtin_fma(p, r, 0x1.a61762a7aded9p-3);
    p = __builtin_fma(p, r, -0x1.ec709dc3a03fdp-3);
    p = __builtin_fma(p, r, 0x1.2776c50ef9bfep-2);
    p = __builtin_fma(p, r, -0x1.71547652b82fep-2);
    p = __builtin_fma(p, r, 0x1.ec709dc3a03fdp-2);
    p = __builtin_fma(p, r, -0x1.71547652b82fep-1);
    p = __builtin_fma(p, r, 0x1.71547652b82fep+0);             // A0 = 1/ln2
    return __builtin_fma(r, p, (double)k + kLog2Tab[i][1]);
}
SBX_HD double d_exp2_tab(double t) {
    double kd = __builtin_fma(t, 32.0, D_MAGIC);
    const int32_t ki = (int32_t)(uint32_t)(d2u(kd) & 0xffffffffull);
    kd = kd - D_MAGIC;
    const double u = __builtin_fma(kd, -0.03125, t) * D_LN2;
    double p = 0x1.6c16c16c16c17p-10;
    p = __builtin_fma(p, u, 0x1.1111111111111p-7);
    p = __builtin_fma(p, u, 0x1.5555555555555p-5);
    p = __builtin_fma(p, u, 0x1.5555555555555p-3);
    p = __builtin_fma(p, u, 0.5);
    p = __builtin_fma(p, u, 1.0);
    p = __builtin_fma(p, u, 1.0);
    const double y = p * kExp2Tab[ki & 31];
    return u2d(d2u(y) + ((uint64_t)(int64_t)(ki >> 5) << 52));
}
// the statement of pow (oracle/sbx_math_ref.h m_pow): what the host runs, and the device's test hook "pow_spec"
SBX_HD float pow_spec_(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (x != x || y != y) return u2f(0x7fc00000u);
    if (x < 0.0f) return u2f(0x7fc00000u);
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : u2f(0x7f800000u);
    if (x == u2f(0x7f800000u)) return (y > 0.0f) ? x : 0.0f;
    double t = (double)y * d_log2_tab((double)x);
    if (t < -160.0) t = -160.0;
    if (t > 136.0) t = 136.0;
    return (float)d_exp2_tab(t);
}
// pow on the device: the SAME operations on the same operands as pow_spec_ in about half the instructions — the polynomial
// coefficients as scalar operands of three-address v_fma_f64 (as literals the compiler moves each of the 19 constants into a register
// pair: 29 v_mov_b32 + 13 two-address v_fmac_f64), exponent / table index / significand from the high word with 32-bit operations, the
// final scaling as a 32-bit add to the high word.  Equal to pow_spec_ by
// construction; tests/test_gpu_round3.py::test_pow_equals_its_statement runs all 2^32 x for every exponent the kernels use.
SBX_HD float pow_(float x, float y) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return pow_spec_(x, y);
#else
    const double xd = (double)x;
    const uint32_t hi = (uint32_t)(d2u(xd) >> 32);
    const uint32_t tmp = hi - 0x3fe60000u;
    const int i = (int)((tmp >> 13) & 127u);
    const int k = (int)tmp >> 20;
    const double z = u2d((d2u(xd) & 0xffffffffull) | ((uint64_t)(hi - (tmp & 0xfff00000u)) << 32));
    const double r = __builtin_fma(z, kLog2Tab[i][0], -1.0);
    double p;
    asm("v_fma_f64 %0, %1, %2, %3\n\t"
        "v_fma_f64 %0, %0, %2, %4\n\t"
        "v_fma_f64 %0, %0, %2, %5\n\t"
        "v_fma_f64 %0, %0, %2, %6\n\t"
        "v_fma_f64 %0, %0, %2, %7\n\t"
        "v_fma_f64 %0, %0, %2, %8\n\t"
        "v_fma_f64 %0, %0, %2, %9"
        : "=&v"(p) : "v"(-0x1.71547652b82fep-3), "v"(r), "s"(0x1.a61762a7aded9p-3), "s"(-0x1.ec709dc3a03fdp-3), "s"(0x1.2776c50ef9bfep-2),
                     "s"(-0x1.71547652b82fep-2), "s"(0x1.ec709dc3a03fdp-2), "s"(-0x1.71547652b82fep-1), "s"(0x1.71547652b82fep+0));
    const double lg = __builtin_fma(r, p, (double)k + kLog2Tab[i][1]);
    double t = (double)y * lg;
    t = (t < -160.0) ? -160.0 : t;                                        // (a NaN t — x = 1, y = inf — stays NaN, as in pow_spec_)
    t = (t > 136.0) ? 136.0 : t;
    double kd = __builtin_fma(t, 32.0, D_MAGIC);
    const int32_t ki = (int32_t)(uint32_t)(d2u(kd) & 0xffffffffull);
    kd = kd - D_MAGIC;
    const double u = __builtin_fma(kd, -0.03125, t) * D_LN2;
    double q;
    asm("v_fma_f64 %0, %1, %2, %3\n\t"
        "v_fma_f64 %0, %0, %2, %4\n\t"
        "v_fma_f64 %0, %0, %2, %5"
        : "=&v"(q) : "v"(0x1.6c16c16c16c17p-10), "v"(u), "s"(0x1.1111111111111p-7), "s"(0x1.5555555555555p-5), "s"(0x1.5555555555555p-3));
    q = __builtin_fma(q, u, 0.5);
    q = __builtin_fma(q, u, 1.0);
    q = __builtin_fma(q, u, 1.0);
    const double yv = q * kExp2Tab[ki & 31];
    const uint64_t yb = d2u(yv);
    const float v = (float)u2d((yb & 0xffffffffull) | ((uint64_t)((uint32_t)(yb >> 32) + ((uint32_t)(ki >> 5) << 20)) << 32));
    // the special cases, in pow_spec_'s order of precedence
    float res = v;
    res = (x == u2f(0x7f800000u)) ? ((y > 0.0f) ? x : 0.0f) : res;
    res = (x == 0.0f) ? ((y > 0.0f) ? 0.0f : u2f(0x7f800000u)) : res;
    res = (x < 0.0f || x != x || y != y) ? u2f(0x7fc00000u) : res;
    return (y == 0.0f) ? 1.0f : res;
#endif
}
// pow_(x, 1 / 2.2f) — linear_to_srgb's exponent (src/util.h:72-77), three times per pixel in every kernel — in ~45 instructions
// instead of pow_'s ~100 (its 19 binary64 constants arrive as literals: the compiler moves each into a register pair first).  The SAME
// log2 (table, reduction and polynomial of d_log2_tab, the exponent / index / significand taken from the high word with 32-bit
// operations, the coefficients as scalar operands of three-address v_fma_f64), t = y log2 x without the clamps (|t| < 70), and
// d_exp2_tab's own 2^t, scaled after the rounding to binary32 (x^(1/2.2) of a binary32 x lies in [2^-68, 2^59]: normal).  The same
// operations on the same operands, hence EQUAL to pow_(x, 1 / 2.2f); run on ALL 2^32 binary32 arguments all the same
// (2^t through exp_reg4k_'s table and degree 3 instead: ONE argument of the 2^32 differs, 0x1.6e3b6ap+93 — not used) (tests/test_gpu_round3.py::test_srgb_pow_equals_pow_everywhere),
// so it needs no domain: to_srgb (sbx_frame.h) uses it on the device.
SBX_HD float srgb_pow_(float x) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return pow_(x, 1.f / 2.2f);
#else
    const double xd = (double)x;
    const uint32_t hi = (uint32_t)(d2u(xd) >> 32);                      // high word: sign, exponent, 20 significand bits
    const uint32_t tmp = hi - 0x3fe60000u;
    const int i = (int)((tmp >> 13) & 127u);
    const int k = (int)tmp >> 20;
    const double z = u2d((d2u(xd) & 0xffffffffull) | ((uint64_t)(hi - (tmp & 0xfff00000u)) << 32));
    const double r = __builtin_fma(z, kLog2Tab[i][0], -1.0);
    double p;
    asm("v_fma_f64 %0, %1, %2, %3\n\t"
        "v_fma_f64 %0, %0, %2, %4\n\t"
        "v_fma_f64 %0, %0, %2, %5\n\t"
        "v_fma_f64 %0, %0, %2, %6\n\t"
        "v_fma_f64 %0, %0, %2, %7\n\t"
        "v_fma_f64 %0, %0, %2, %8\n\t"
        "v_fma_f64 %0, %0, %2, %9"
        : "=&v"(p) : "v"(-0x1.71547652b82fep-3), "v"(r), "s"(0x1.a61762a7aded9p-3), "s"(-0x1.ec709dc3a03fdp-3), "s"(0x1.2776c50ef9bfep-2),
                     "s"(-0x1.71547652b82fep-2), "s"(0x1.ec709dc3a03fdp-2), "s"(-0x1.71547652b82fep-1), "s"(0x1.71547652b82fep+0));
    const double lg = __builtin_fma(r, p, (double)k + kLog2Tab[i][1]);
    const double t = (double)(1.f / 2.2f) * lg;
    double kd = __builtin_fma(t, 32.0, D_MAGIC);                          // d_exp2_tab's own steps
    const int32_t ki = (int32_t)(uint32_t)(d2u(kd) & 0xffffffffull);
    kd = kd - D_MAGIC;
    const double u = __builtin_fma(kd, -0.03125, t) * D_LN2;
    double q;
    asm("v_fma_f64 %0, %1, %2, %3\n\t"
        "v_fma_f64 %0, %0, %2, %4\n\t"
        "v_fma_f64 %0, %0, %2, %5"
        : "=&v"(q) : "v"(0x1.6c16c16c16c17p-10), "v"(u), "s"(0x1.1111111111111p-7), "s"(0x1.5555555555555p-5), "s"(0x1.5555555555555p-3));
    q = __builtin_fma(q, u, 0.5);
    q = __builtin_fma(q, u, 1.0);
    q = __builtin_fma(q, u, 1.0);
    const float y = __builtin_ldexpf((float)(q * kExp2Tab[ki & 31]), ki >> 5);
    // pow_'s special cases for y > 0: NaN or negative -> the quiet NaN; +-0 -> +0; +inf -> +inf
    float res = (x == 0.0f) ? 0.0f : y;
    res = (x == u2f(0x7f800000u)) ? x : res;
    return (x < 0.0f || x != x) ? u2f(0x7fc00000u) : res;
#endif
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
