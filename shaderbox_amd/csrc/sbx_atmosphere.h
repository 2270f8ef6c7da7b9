// shaderbox_amd/csrc/sbx_atmosphere.h — get_incident_light / get_sun_light of /root/reference/src/app_atmosphere.h:50-160 as
// device functions: APP_ATMOSPHERE's kernel (kern_atmosphere.hip) and the config-5 composite SBX_APP_PLANET_ATMOSPHERE
// (kern_planet.hip: APP_PLANET's background() replaced by this sky) share them.  The ATM_* switches and their measurements are
// kern_atmosphere.hip's; FIN = false is the plain statement of the spec (guarded exp, sqrt_n_, division by the exact reciprocal).
#pragma once
#include "sbx_device.h"
#include "sbx_exp4k_table.h"

namespace sbx {


constexpr float ATM_EARTH_R = 6360e3f, ATM_ATMOS_R = 6420e3f, ATM_HR = 7994.0f, ATM_HM = 1200.0f;   // :34-38
constexpr double ATM_HR_RD = 1.0 / (double)ATM_HR, ATM_HM_RD = 1.0 / (double)ATM_HM;   // exact division by constants (sbx_math.h div_by)

// isect_sphere with the atmosphere sphere (origin 0)                        :15-26
__device__ __forceinline__ bool isect_atmosphere(v3 ro, v3 rd, float& t1) {
    const v3 rc = V3(0, 0, 0) - ro;
    const float radius2 = ATM_ATMOS_R * ATM_ATMOS_R;
    const float tca = dot(rc, rd);
    const float d2 = dot(rc, rc) - tca * tca;
    const float thc = sqrt_n_(radius2 - d2);   // both ~4e13: the difference is a multiple of 4e6, zero or negative
    t1 = tca + thc;
    return d2 < radius2;
}

// exp_ of this kernel reads the 2^(j/32) table from LDS.  For the density terms exp(-height / H) — 288 of the 336 exp of an
// in-dome pixel — the binary32 range guard is left out when the uniforms are finite (FIN, decided on the host): a sample lies
// inside the atmosphere sphere, so -height / H is >= -60e3 / 1200 = -50.1; a light sample below the ground returns before its exp
// (:65), but a VIEW ray that dips below the horizon marches through the planet (no ground test, :119-122) with heights down to
// -6.36e6, i.e. arguments up to +5300, where exp_ (guard at 89) and the guard-less forms alike overflow to +inf.  The guard-less
// forms are shown equal to exp_ on every argument in [-80, 2^18]; a NaN passes through the guard unchanged.  exp(-tau): grazing sun
// rays reach optical depths beyond 104 (a ring of 9 % of the pixels turns NaN / inf without the guard), so the guard-less form is
// used only where a wave-wide test shows tau <= 80 for every lane (ATM_TAU4K: 4.02 -> 3.94 ms).
#ifndef ATM_EXP_REG
#define ATM_EXP_REG 1      // the density terms through exp_reg_ (sbx_math.h) when the uniforms are finite
#endif
#ifndef ATM_EXP64
#define ATM_EXP64 1        // ... in its 64-entry / degree-5 form (exp_reg64_: one binary64 fma less, exhaustively equal on |x| <= 80)
#endif
#ifndef ATM_EXP4K
#define ATM_EXP4K 1        // ... in its 4096-entry / degree-3 form (exp_reg4k_: two more fma and a register-pair move less, exhaustively equal
#endif                     // on |x| <= 80).  The 32 KB table is read where it lies, in global memory through the vector L1: 7680x4320
                           // 4.29 -> 4.03 ms.  (A copy in LDS per workgroup of 16 / 8 / 4 waves: 4.28 / 4.05 / 4.23 ms — 65 000 copies of
                           // 32 KB, and a workgroup's LDS is held until its last wave ends; per-CU persistent workgroups that copy once
                           // and walk through the tiles: the tile loop makes the compiler hoist the kernel's constants into registers,
                           // 95 VGPRs / 5 waves or 84 B of scratch at 64, 4.66 ms.  profiles/r03_log.md)
#ifndef ATM_TAU4K
#define ATM_TAU4K 1         // exp(-tau) through exp_reg4k_ where a wave's optical depths allow it
#endif
#ifndef ATM_DIV3
#define ATM_DIV3 1         // FIN kernels: -height / H through div3_ (sbx_math.h; the divisors and their reciprocals held in VGPRs) instead of
#endif                     // div_by's binary64 multiply: |height| is 0 or in [.5, 6.4e6] (a multiple of ulp(6.4e6)), the quotient only feeds exp
#ifndef ATM_SQRT_RS
#define ATM_SQRT_RS 1      // FIN kernels: length(s) of a march position through sqrt_rs_ (sbx_math.h: five instructions, exact for finite
#endif                     // x >= 2^-102): |s|^2 is ~4e13 along rays that stay above the ground; a view ray below the horizon passes through
                           // the planet, but a position's components are multiples of their own ulp, so |s|^2 is either >= 2^-40 or exactly 0,
                           // and exactly 0 needs all three components to cancel at once: rd.x = rd.z = 0 only for theta = 0, the ray straight
                           // UP (acos never returns pi exactly), and a light sample under the ground ends its march before it could reach the
                           // centre.  (isect_atmosphere keeps sqrt_n_: its argument can be exactly 0.)
#define ATM_LEN(x) ((FIN && ATM_SQRT_RS) ? sqrt_rs_(x) : sqrt_n_(x))
// PREC = 1: the TOLERANCE tier (include/sbx.h sbx_set_precision, SBX_PRECISION_1E4; opt-in, never the default, never in bench.py's
// `value`).  north_star's bar is 1e-4 per channel, not bit-equality, and this kernel — 336 exp per in-dome pixel, each 15 instructions
// of binary64 table arithmetic — has no threshold that turns a rounding difference into a different pixel: every exp becomes
// v_exp_f32 of x * log2(e) (two instructions, ~2 ulp) and -height / H one multiply by RN(1 / H).  Everything else (the dome mapping,
// the square roots, the phase functions, the order of every sum, the overflow to +inf of a ray inside the planet) is unchanged.
// Measured against the oracle on every pixel of the 7680x4320 frame and over the sun sweep: tests/test_gpu_round5.py.
__device__ __forceinline__ float atm_exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
#ifndef ATM_TX
#define ATM_TX 1           // waves per workgroup (1: 4.03 ms, 4: 4.06)
#endif
#if ATM_EXP_REG && ATM_EXP4K
#define ATM_EXP_H0(x) (FIN ? exp_reg4k_((x), kExp2Tab4096) : exp_tab_<true>((x), etab))
#elif ATM_EXP_REG && ATM_EXP64
#define ATM_EXP_H0(x) (FIN ? exp_reg64_<false>((x), etab64) : exp_tab_<true>((x), etab))
#elif ATM_EXP_REG
#define ATM_EXP_H0(x) (FIN ? exp_reg_<false>((x), etab) : exp_tab_<true>((x), etab))
#else
#define ATM_EXP_H0(x) exp_tab_<!FIN>((x), etab)
#endif
#define ATM_EXP_H(x) (PREC ? atm_exp_fast(x) : ATM_EXP_H0(x))
#define ATM_EXP(x) (PREC ? atm_exp_fast(x) : exp_tab_<true>((x), etab))

// (march_pos + 0.5 * march_step below is written fma(.5, march_step, march_pos): the half is exact, so it is one rounding either way)
struct AtmDiv { float hr, rhr, hm, rhm; };            // H_R, RN(1 / H_R), H_M, RN(1 / H_M)
#define ATM_DIV_HR(h) (PREC ? (h) * K.rhr : (FIN && ATM_DIV3) ? div3_((h), K.hr, K.rhr) : div_by((h), ATM_HR_RD))
#define ATM_DIV_HM(h) (PREC ? (h) * K.rhm : (FIN && ATM_DIV3) ? div3_((h), K.hm, K.rhm) : div_by((h), ATM_HM_RD))
template <bool FIN, int PREC = 0>
__device__ __forceinline__ bool sun_light(v3 ro, v3 rd, float& odR, float& odM, const double (&etab)[32], const double* etab64, const AtmDiv& K) {   // :50-76
    float t1;
    isect_atmosphere(ro, rd, t1);
    float march_pos = 0.f;
    const float march_step = t1 / 8.f;
#ifndef ATM_BATCH
#define ATM_BATCH 0        // FIN kernels: the eight heights of a sun march first, then its sixteen exp (see below).  MEASURED AND NOT
#endif                     // TAKEN: 7680x4320 3.41 -> 3.57 ms (1), 3.50 (2: first height alone, then the other seven); same bits
    if (FIN && ATM_BATCH) {
        // The reference leaves the march at the first sample under the ground (:65-67) and its caller then ignores the optical
        // depths.  So the eight heights — which depend on nothing but the ray — come first, a lane with any of them negative is
        // done (it never pays for an exp; before, a lane that failed at sample 5 had paid for ten), and the sixteen exp of the
        // other lanes are independent chains of binary64 fma that the scheduler interleaves; the sums keep the reference's order.
        float h[8];
        bool under = false;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const v3 s = ro + rd * __builtin_fmaf(0.5f, march_step, march_pos);
            h[i] = ATM_LEN(dot(s, s)) - ATM_EARTH_R;
            under = under || (h[i] < 0.f);                      // (a NaN height is not "under": the reference marches on)
            march_pos += march_step;
            if (ATM_BATCH == 2 && i == 0 && under) return false;   // the usual failure is the first sample (far side of the planet)
        }
        if (under) return false;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            odR += ATM_EXP_H(ATM_DIV_HR(-h[i])) * march_step;
            odM += ATM_EXP_H(ATM_DIV_HM(-h[i])) * march_step;
        }
        return true;
    }
    for (int i = 0; i < 8; ++i) {
        const v3 s = ro + rd * __builtin_fmaf(0.5f, march_step, march_pos);
        const float height = ATM_LEN(dot(s, s)) - ATM_EARTH_R;   // length(s), |s| ~ 6.4e6
        if (height < 0.f) return false;
        odR += ATM_EXP_H(ATM_DIV_HR(-height)) * march_step;
        odM += ATM_EXP_H(ATM_DIV_HM(-height)) * march_step;
        march_pos += march_step;
    }
    return true;
}

// get_incident_light :78-160 for the ray (ro, rd): 16 view samples, each with an 8-sample march towards the sun
template <bool FIN, int PREC = 0>
__device__ __forceinline__ v3 atm_incident_light(v3 ro, v3 rd, v3 sun_dir, const double (&etab)[32], const double* etab64) {
    v3 col = V3(0.f, 0.f, 0.f);
    float t1;
    AtmDiv K{ATM_HR, 1.0f / ATM_HR, ATM_HM, 1.0f / ATM_HM};
    if ((FIN && ATM_DIV3) || PREC) asm volatile("" : "+v"(K.hr), "+v"(K.rhr), "+v"(K.hm), "+v"(K.rhm));      // VGPR operands: full rate
    if (isect_atmosphere(ro, rd, t1)) {                             // get_incident_light :78-160
        const v3 betaR = V3(5.5e-6f, 13.0e-6f, 22.4e-6f), betaM = V3(21e-6f, 21e-6f, 21e-6f);   // :29-30
        const float march_step = t1 / 16.f;
        const float mu = dot(rd, sun_dir);
        const float phaseR = 3.f * (1.f + mu * mu) / (16.f * 3.14159265359f);            // volumetric.h:13-19
        const float g = .76f;
        const float phaseM = (1.f - g * g) / ((4.f + 3.14159265359f) * pow_(1.f + g * g - 2.f * g * mu, 1.5f));
        float odR = 0.f, odM = 0.f, march_pos = 0.f;
        v3 sumR = V3(0, 0, 0), sumM = V3(0, 0, 0);
        for (int i = 0; i < 16; ++i) {
            const v3 s = ro + rd * __builtin_fmaf(0.5f, march_step, march_pos);
            const float height = ATM_LEN(dot(s, s)) - ATM_EARTH_R;   // length(s)
            const float hr = ATM_EXP_H(ATM_DIV_HR(-height)) * march_step;
            const float hm = ATM_EXP_H(ATM_DIV_HM(-height)) * march_step;
            odR += hr;
            odM += hm;
            // DEAD RAYS.  A view ray below the horizon dives into the planet (there is no ground test in this march, :119-122);
            // 106 km down exp(-height / hM) overflows and optical_depthM is +inf from then on (709 km: optical_depthR).  Every
            // later sample's tau = betaR (odR + lR) + 1.1 betaM (odM + lM) is then +inf in all three components — the terms are
            // non-negative, so no inf - inf — its attenuation exp(-inf) is exactly 0, and what it would add, hr * 0 and hm * 0, is
            // exactly +0: a sample is only lit when its FIRST light sample is above the ground (:65-67), which the geometry
            // allows down to ~4 km under it (a light step is at most 1/16 of the way out of the atmosphere), so the hr, hm of a
            // lit sample are finite (<= e^3.4 * step).  So a lane whose optical depth has overflowed is finished: it skips the
            // sun marches of its far-side samples (which the reference runs for nothing), and the march ends when the whole
            // wave is finished.  The band 1.09 < z2 <= 1.4 of the dome — 10 % of a 16:9 frame — paid a sky pixel's full price for
            // a black pixel before: 7680x4320 3.90 -> 3.66 ms, same bits (tools/ab_time.py; waves that mix finished and live rays still pay).
#ifndef ATM_DEAD_EXIT
#define ATM_DEAD_EXIT 1
#endif
            const float inf = u2f(0x7f800000u);
            const bool dead = ATM_DEAD_EXIT && (odR == inf || odM == inf);
            if (ATM_DEAD_EXIT && __builtin_amdgcn_ballot_w64(!dead) == 0ull) break;      // wave-uniform
            float lR = 0.f, lM = 0.f;
            if (!dead && sun_light<FIN, PREC>(s, sun_dir, lR, lM, etab, etab64, K)) {
                const v3 tau = betaR * (odR + lR) + betaM * 1.1f * (odM + lM);
                // exp(-tau): the guard-less form where every lane that got here has all three tau <= 80 (tau >= 0: sums of
                // non-negative terms; a NaN fails the test), exp_'s guarded form for the wave otherwise (grazing sun rays)
                v3 att;
                if (!PREC && FIN && ATM_EXP_REG && ATM_EXP4K && ATM_TAU4K &&
                    __builtin_amdgcn_ballot_w64(!(fmax_(fmax_(tau.x, tau.y), tau.z) <= 80.f)) == 0ull)
                    att = V3(exp_reg4k_(-tau.x, kExp2Tab4096), exp_reg4k_(-tau.y, kExp2Tab4096), exp_reg4k_(-tau.z, kExp2Tab4096));
                else
                    att = V3(ATM_EXP(-tau.x), ATM_EXP(-tau.y), ATM_EXP(-tau.z));
                sumR = sumR + hr * att;
                sumM = sumM + hm * att;
            }
            march_pos += march_step;
        }
        col = 20.0f * (sumR * phaseR * betaR + sumM * phaseM * betaM);   // sun_power :41
    }
    return col;
}

}  // namespace sbx
