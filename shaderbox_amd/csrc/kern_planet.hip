// shaderbox_amd/csrc/kern_planet.hip — APP_PLANET: fBm terrain sphere + volumetric cloud shell.
//
// Follows /root/reference/src/app_planet.h (CLOUDS and LIGHT defined, :63,249): render :303-367,
// sdf_terrain_map :175-186, sdf_terrain_map_detail :188-199, sdf_terrain_normal :201-212,
// clouds_map :102-119, clouds_march :121-141, clouds_shadow_march :143-160, integrate_volume
// :79-100, setup_lights :217-236, illuminate :238-298, background :23-41.  The three rotation
// matrices, transpose(rot) and the light vector are frame constants (FramePlanet).
// NaN policy: smoothstep(1-.3s, 1-.2s, N) with s = 0 is 0/0 for N == 1 (:270-273); NaN is data and
// flows to the framebuffer exactly as IEEE arithmetic dictates.
// Single-wave workgroups (as k_clouds): waves never talk to each other (one hash cache and one park area per wave), and a
// 4-wave workgroup keeps its four slots until its slowest wave is done — across the planet's limb and the cloud shell the waves of
// one workgroup differ a lot.  7680x4320: 7.09 -> 6.75 ms (2 waves per workgroup: 7.18).
#ifndef SBX_WG_WAVES
#define SBX_WG_WAVES 1
#endif
#ifndef SBX_HC_SLOTS
#define SBX_HC_SLOTS 32         // 4.7 KB of LDS per wave: 5 waves per SIMD fit (64 slots: 9.3 KB, 4 waves)
#endif
#include "sbx_device.h"
#include "sbx_noise.h"
#include "sbx_hashcache.h"
#include "sbx_exp4k_table.h"
#include "sbx_atmosphere.h"

// All noise_iq evaluations go through the per-wave lattice-hash cache (sbx_hashcache.h): octave k of any fBm
// uses table k & 3.  The cross-lane steps of the cache need wave-uniform control flow, so lanes never leave
// a loop early; they carry predicates (`tm` terrain march, `cm` cloud march, `hitl` ground hit) and the loops
// test wave_any().  A predicated-off lane still evaluates the arithmetic (harmlessly) but commits nothing.

namespace sbx {

constexpr float PL_MAX_HEIGHT = .4f;                    // :20
constexpr float PL_MAX_RAY_DIST = PL_MAX_HEIGHT * 4.f;  // :21

struct Vol { v3 origin, pos; float height, transmittance, radiance, alpha; };   // volumetric.h:47-68 (radiance r=g=b)
__device__ __forceinline__ Vol make_vol(v3 o) { return Vol{o, o, 0.f, 1.f, 0.f, 0.f}; }

// smoothstep with literal edges: the division by (e1 - e0) is an exact multiply by its binary64 reciprocal — or, in the SKIP kernels,
// div3_ of sbx_math.h: three full-rate binary32 instructions with the literal divisor and its reciprocal, equal to the IEEE quotient
// for every pair of significands.  Its domain (finite dividend, zero or within 2^+-100; the sign of a zero quotient unused) holds on
// the tame frames those kernels run for: the dividends are differences of numbers of order 1 (zero or >= 2^-26), exp of a height in
// [-2.5, 2], a smoothstep value (zero or >= 2^-60) or |p| - 1 (zero — always +0 — or >= 2^-24); a NaN gives NaN either way.
#ifndef PL_DIV3
#define PL_DIV3 1
#endif
#define PL_DIVK(a, dlit) ((SKIP && PL_DIV3) ? div3_((a), (dlit), 1.0f / (dlit)) : div_by((a), 1.0 / (double)(dlit)))
#ifndef PL_MED3
#define PL_MED3 1           // the clamp of those smoothsteps as one v_med3_f32: their arguments (fBm sums of hashes, heights of finite positions)
#endif                      // are never NaN on the tame frames of the SKIP kernels (finite u_time, u_res checked by the C API, fixed camera)
#define SMOOTHSTEP_K(e0, e1, x) ((SKIP && PL_DIV3 && PL_MED3) ? smoothstep_d3_med3((e0), (e1) - (e0), 1.0f / ((e1) - (e0)), (x)) \
                                 : (SKIP && PL_DIV3) ? smoothstep_d3((e0), (e1) - (e0), 1.0f / ((e1) - (e0)), (x)) \
                                                   : smoothstep_rd((e0), 1.0 / (double)((e1) - (e0)), (x)))
template <bool SKIP>
__device__ __forceinline__ float band(float t) {                     // band(.2, .35, .65, t)  util.h:103-112
    return SMOOTHSTEP_K(.2f, .35f, t) * (1.f - SMOOTHSTEP_K(.35f, .65f, t));
}

// DECL_FBM_FUNC(name, OCT, basis(noise_iq(p)))  fbm.h:6 — the OCT noise values come from one cooperative batch
// MODE 0: noise(p)   1: anoise = |2 noise - 1| (:65)   2: rnoise = 1 - |2 noise - 1| (:167)
template <int MODE>
__device__ __forceinline__ float pl_basis(float n) {
    if (MODE == 0) return n;
    // (n * 2 is exact, so n * 2 - 1 is one rounding: fma(n, 2, -1) has the reference's bits)
    if (MODE == 1) return abs_(__builtin_fmaf(n, 2.f, -1.f));
    return 1.f - abs_(__builtin_fmaf(n, 2.f, -1.f));
}
template <int OCT, int MODE>
__device__ __forceinline__ float pl_fbm_from(const float (&nz)[OCT], float init_gain, float gain) {
    float H = init_gain, t = 0.f;
#pragma unroll
    for (int i = 0; i < OCT; ++i) {
        t += pl_basis<MODE>(nz[i]) * H;
        H *= gain;
    }
    return t;
}

// fBm over the cached noise: octaves START..OCT-1 are fetched in cooperative batches of <= 4 (register budget) and
// added to (t, H, q) in octave order exactly as fbm.h:6 (t += basis * H; p *= lacunarity; H *= gain)
#ifndef PL_BATCH
#define PL_BATCH 3          // octaves fetched per cooperative batch: 4 at a time (round 1) peaked at 32 hash registers and spilled; 3 (round 5): 91 VGPRs, no scratch, -0.5 % against 2
#endif
#ifndef PL_ROLL_DETAIL
#define PL_ROLL_DETAIL 1
#endif
#ifndef PL_BATCH_DETAIL
#define PL_BATCH_DETAIL 2   // the 7-octave detail maps of the hit shading (6 per hit pixel)
#endif
#ifndef PL_SPEC
#define PL_SPEC 1           // the hash pass of a lone pending cell also hashes the rest of the 2 x 2 x 2 block ahead of the march
#endif                      // (sbx_hashcache.h hc_insert SPEC; the SKIP kernels only)
#ifndef PL_TB2
#define PL_TB2 3            // table of octave k of terrain_map's SECOND fBm: (PL_TB2 + k) & 3 — with 0 its octave k shares table k with
#endif                      // the first fBm's octave k (another lattice), and the two evict each other's cells on every slot clash
template <int OCT, int MODE, int START, bool XI = false, int TB = 0>
__device__ __forceinline__ void coop_fbm_range(WaveCache& S, v3& q, float lacunarity, float& H, float gain, float& t, bool on, int lane) {
    constexpr int B = (OCT == 7) ? PL_BATCH_DETAIL : PL_BATCH;
#if PL_ROLL_DETAIL
    if (OCT == 7) {
        // the 7-octave detail maps as a REAL loop over their batches (same operations in the same order): unrolled, the six maps of
        // the hit shading are ~40 KB of straight-line code that the scheduler interleaves across octaves — the spills of the kernel
        constexpr int NB = (OCT - START) / B;
#pragma unroll 1
        for (int b = 0; b < NB; ++b) {
            const int base = START + b * B;
            v3 p[B]; int tab[B]; float nz[B];
#pragma unroll
            for (int i = 0; i < B; ++i) { p[i] = q; tab[i] = (TB + base + i) & 3; q = q * lacunarity; }
            coop_noise_n<B, XI, XI && PL_SPEC>(S, p, tab, on, lane, nz);
#pragma unroll
            for (int i = 0; i < B; ++i) { t += pl_basis<MODE>(nz[i]) * H; H *= gain; }
        }
        constexpr int R = (OCT - START) % B;
        if (R > 0) {
            v3 p[R > 0 ? R : 1]; int tab[R > 0 ? R : 1]; float nz[R > 0 ? R : 1];
#pragma unroll
            for (int i = 0; i < R; ++i) { p[i] = q; tab[i] = (TB + START + NB * B + i) & 3; q = q * lacunarity; }
            coop_noise_n<(R > 0 ? R : 1), XI, XI && PL_SPEC>(S, p, tab, on, lane, nz);
#pragma unroll
            for (int i = 0; i < R; ++i) { t += pl_basis<MODE>(nz[i]) * H; H *= gain; }
        }
        return;
    }
#endif
#pragma unroll
    for (int base = START; base < OCT; base += B) {
        if (base + B <= OCT) {
            v3 p[B]; int tab[B]; float nz[B];
#pragma unroll
            for (int i = 0; i < B; ++i) { p[i] = q; tab[i] = (TB + base + i) & 3; q = q * lacunarity; }
            coop_noise_n<B, XI, XI && PL_SPEC>(S, p, tab, on, lane, nz);
#pragma unroll
            for (int i = 0; i < B; ++i) { t += pl_basis<MODE>(nz[i]) * H; H *= gain; }
        } else {
            constexpr int R = (OCT - START) % B;
            v3 p[R > 0 ? R : 1]; int tab[R > 0 ? R : 1]; float nz[R > 0 ? R : 1];
#pragma unroll
            for (int i = 0; i < R; ++i) { p[i] = q; tab[i] = (TB + base + i) & 3; q = q * lacunarity; }
            coop_noise_n<(R > 0 ? R : 1), XI, XI && PL_SPEC>(S, p, tab, on, lane, nz);
#pragma unroll
            for (int i = 0; i < R; ++i) { t += pl_basis<MODE>(nz[i]) * H; H *= gain; }
        }
    }
}
template <int OCT, int MODE, bool XI = false>
__device__ __forceinline__ float coop_fbm(WaveCache& S, v3 q, float lacunarity, float init_gain, float gain, bool on, int lane) {
    float H = init_gain, t = 0.f;
    coop_fbm_range<OCT, MODE, 0, XI>(S, q, lacunarity, H, gain, t, on, lane);
    return t;
}

// The two exp of a cloud sample in the SKIP kernels: exp_reg4k_ of sbx_math.h (4096-entry table read through the vector L1, degree 3,
// no range guard: 15 instructions against exp_'s 21), equal to exp_ for |x| <= 80.  The SKIP kernels run only for tame frames
// (sbx_capi.hip tame_time: finite, bounded camera), where a sample lies within the atmosphere shell: the arguments are
// -30.034 * dens * t_step with dens in [0, .94] and t_step <= .08, i.e. [-2.26, 0], and height = (|p| - 1) / .4 in [-2.5, 1.1]; a NaN
// stays a NaN in both forms.  7680x4320: see profiles/r03_log.md.
#ifndef PL_EXP4K
#define PL_EXP4K 1
#endif
// length() of a march position in the SKIP kernels: sqrt_n_ of sbx_math.h — v_sqrt_f32 and the two-sided fix-up WITHOUT the input
// scaling and class tests of the compiler's IEEE expansion (~35 instead of ~60 issue cycles), bit-identical to it for every
// argument that is 0, not finite, negative, or >= 2^-96 (all 2^32 run: test_sqrt_n_is_ieee_sqrt).  Here the argument is |p|^2 of a
// point p = rot * (eye + s * rd) with the reference's fixed eye (0, 0, -2.5) (:47-58): lanes that miss the atmosphere keep
// o == 0 exactly (hit_o = 0, t = 0), every other lane has rd.x, rd.y either exactly 0 (the centre ray, which stops on the terrain
// at |o| >= 1) or >= 1e-6 in magnitude (pixel centres), and o.x = (t0 + t) * rd.x has no cancellation: |o|^2 >= 1e-12 >> 2^-96 = 1.3e-29.
// The SKIP kernels run only for tame frames (finite, bounded u_time); a NaN resolution gives NaN, which sqrt_n_ passes through.
#ifndef PL_SQRT_RS
#define PL_SQRT_RS 1
#endif
#ifndef PL_SQRT_N
#define PL_SQRT_N 1
#endif
template <bool SKIP>
__device__ __forceinline__ float pl_length(v3 v) {
    // (PL_SQRT_RS: sqrt_rs_, five instructions, exact for finite x >= 2^-102 — the lanes whose |o|^2 is exactly 0 are the ones that
    //  missed the atmosphere: they never commit a result, and what they compute here, a NaN, decides nothing)
    return (SKIP && PL_SQRT_RS) ? sqrt_rs_(dot(v, v)) : (SKIP && PL_SQRT_N) ? sqrt_n_(dot(v, v)) : length(v);
}
#define PL_EXP(x) ((SKIP && PL_EXP4K) ? exp_reg4k_((x), kExp2Tab4096) : exp_(x))
// clouds_map :102-119 + integrate_volume :79-100; `on` = lanes that commit
// SKIP = false (sbx_set_variant 1) evaluates everything: the reference form, kept for the parity sweeps
// The density part of clouds_map: false = nothing would change for any committing lane of the wave (see below), else
// dens and T_i = exp(-30.034 dens t_step) of every lane are set.
template <bool SKIP>
__device__ __forceinline__ bool clouds_density(WaveCache& S, v3 pos, float height, float t_step, bool on, int lane, float& dens_out,
                                               float& T_i_out) {
    // The cloud shell is the height band (.2, .65): outside it band() is exactly +0, so dens = fbm * 0 = +0,
    // T_i = exp(-0) = 1, and the three updates below are `*= 1`, `+= 0`, `+= 0 * (1 - alpha)`: nothing changes.
    // When that holds for every committing lane of the wave the noise is not evaluated at all (most steps of the
    // 75-step march and 2-3 of the 5 shadow steps).  A NaN height compares unequal and takes the full path.
    const float bd = band<SKIP>(height);
    if (SKIP && !wave_any(on && bd != 0.f)) return false;
    // fbm of |2 noise - 1| in [0, 1], gains .5 .25 .125 .0625: evaluated in stages {0,1}, {2}, {3}; once the part so far
    // plus the largest possible rest is below the coverage edge for every committing lane, smoothstep(cov, ..) is
    // exactly 0, the density +0, and nothing below would change anything.
    const float cov = .29475675f, fuzzy = .0335f;
    v3 q = pos * 3.2343f + V3(.35f, 13.35f, 2.67f);
    float H = .5f, dens = 0.f;
    // (SKIP kernels run only for tame frames — sbx_capi.hip tame_time — where every lattice coordinate is a small integer: XI)
    coop_fbm_range<2, 1, 0, SKIP>(S, q, 2.0276f, H, .5f, dens, on, lane);
    if (SKIP && !wave_any(on && !(dens + .1876f < cov))) return false;
    coop_fbm_range<3, 1, 2, SKIP>(S, q, 2.0276f, H, .5f, dens, on, lane);
    if (SKIP && !wave_any(on && !(dens + .06255f < cov))) return false;
    coop_fbm_range<4, 1, 3, SKIP>(S, q, 2.0276f, H, .5f, dens, on, lane);
    dens *= SMOOTHSTEP_K(cov, cov + fuzzy, dens);
    dens *= bd;
    // dens is exactly +0 below the coverage edge as well (smoothstep = 0): same identities, skip the two exp
    if (SKIP && !wave_any(on && dens != 0.f)) return false;
    dens_out = dens;
    T_i_out = PL_EXP(-30.034f * dens * t_step);
    return true;
}
// clouds_map :102-119 + integrate_volume :79-100; `on` = lanes that commit
// SKIP = false (sbx_set_variant 1) evaluates everything: the reference form, kept for the parity sweeps
template <bool SKIP>
__device__ __forceinline__ void clouds_map(WaveCache& S, Vol& c, float t_step, bool on, int lane) {
    float dens = 0.f, T_i = 1.f;
    if (!clouds_density<SKIP>(S, c.pos, c.height, t_step, on, lane, dens, T_i)) return;
    if (on) {
        c.transmittance *= T_i;
        c.radiance += dens * PL_DIVK(PL_EXP(c.height), .055f) * c.transmittance * t_step;
        c.alpha += (1.f - T_i) * (1.f - c.alpha);
    }
}

// sdf_terrain_map / sdf_terrain_map_detail :175-199
template <int OCT, bool SKIP>
__device__ __forceinline__ v2 terrain_map(WaveCache& S, v3 pos, bool on, int lane) {
    const float h0 = coop_fbm<OCT, 0, SKIP>(S, pos * 2.0987f, 2.0244f, .454f, .454f, on, lane);
    const float n0 = SMOOTHSTEP_K(.35f, 1.f, h0);
    // Second fBm (ridged basis, values in [0, 1]): n1 = smoothstep(.6, 1, h1) is exactly +0 whenever h1 < .6.
    // After the first octave, h1 <= t + (.454^2 + ... + .454^OCT); when that bound is below .6 for every
    // committing lane of the wave the remaining octaves cannot change n1 = +0 and are not evaluated.
    v3 q = pos * 1.50987f + V3(1.9489f, 2.435f, .5483f);
    float H = .454f, h1 = 0.f;
    coop_fbm_range<1, 2, 0, SKIP, PL_TB2>(S, q, 2.0244f, H, .454f, h1, on, lane);
    constexpr float TAIL = (OCT == 3) ? .2998f : .3745f;       // sum of .454^k, k = 2..OCT, rounded up (OCT = 3 or 7)
    static_assert(OCT == 3 || OCT == 7, "tail bound tabulated for 3 and 7 octaves");
    float n1 = 0.f;
    if (!SKIP || wave_any(on && !(h1 + TAIL * 1.0001f < .6f))) {
        coop_fbm_range<OCT, 2, 1, SKIP, PL_TB2>(S, q, 2.0244f, H, .454f, h1, on, lane);
        n1 = SMOOTHSTEP_K(.6f, 1.f, h1);
    }
    const float n = n0 + n1;
    return V2(pl_length<SKIP>(pos) - 1.f - n * PL_MAX_HEIGHT, PL_DIVK(n, PL_MAX_HEIGHT));
}


// ---- the central differences of sdf_terrain_normal (:201-212) as PAIRS --------------------------------------------------------
// The six detail maps of a hit pixel are three pairs pos + d, pos - d with d along one axis (e = .001): the two points of a pair
// have two coordinates in common — bit for bit, and so all the way down both fBms (the same multiplications of the same numbers) —
// and lie in the same lattice cell except where the 0.002 between them straddles a cell face.  A pair is therefore evaluated
// together: per octave the floor / fraction / smoothstep weight of the two common axes ONCE, ONE lookup where the cells coincide,
// and of the trilinear blend (x, then y, then z: noise_iq.h:20-23) everything below the differing axis once — nothing shared for an
// x pair, the four x-mixes for a y pair, the x- and y-mixes for a z pair.  Lanes whose pair straddles a face get the second point's
// own lookup and blend behind a wave-wide test.  Every value is computed by the reference's operations on the reference's operands;
// only the duplicates are gone.  (pos.c + 0 and pos.c - 0 are pos.c unless it is a zero: a wave with a zero coordinate takes the
// unpaired path.)
// (All SIX points together — they are perturbations of one centre: one lookup instead of three, the centre's x- and y-mixes shared
//  by the y and z points, 90 blend operations instead of 109 — was built and measured: same bits, 5.57 against 5.53 ms; six
//  accumulators and six perturbed coordinates through both fBms cost more than the shared work saves.  profiles/r05_log.md)
#ifndef PL_PAIRS
#define PL_PAIRS 1
#endif
template <bool XI, bool SPEC>
__device__ __forceinline__ H8 pl_lookup(WaveCache& S, int tab, unsigned nbits, int slot, bool active, int lane) {
    H8 h;
    if (wave_any(active && S.tag[tab][slot] != nbits)) {
        h = hc_slow<XI, SPEC>(S, tab, nbits, slot, active, lane);
    } else {
        h.lo = *reinterpret_cast<const float4*>(&S.h[tab][slot][0]);
        h.hi = *reinterpret_cast<const float4*>(&S.h[tab][slot][4]);
    }
    return h;
}
// one octave: qa = the pair's first point, qb_ax = the second point's coordinate on axis AX (its other two are qa's)
template <int AX, bool XI>
__device__ __forceinline__ void pl_noise_pair(WaveCache& S, v3 qa, float qb_ax, int tab, bool on, int lane, float& na, float& nb) {
    const float px = floor_(qa.x), py = floor_(qa.y), pz = floor_(qa.z);
    const float ax = qa.x - px, ay = qa.y - py, az = qa.z - pz;
    const float fx = ax * ax * tm2_(ax), fy = ay * ay * tm2_(ay), fz = az * az * tm2_(az);
    const float pb = floor_(qb_ax), ab = qb_ax - pb, fb = ab * ab * tm2_(ab);
    const float bx = AX == 0 ? pb : px, by = AX == 1 ? pb : py, bz = AX == 2 ? pb : pz;
    const float n_a = XI ? __builtin_fmaf(113.0f, pz, __builtin_fmaf(py, 157.0f, px)) : px + py * 157.0f + 113.0f * pz;
    const float n_b = XI ? __builtin_fmaf(113.0f, bz, __builtin_fmaf(by, 157.0f, bx)) : bx + by * 157.0f + 113.0f * bz;
    const unsigned ka = f2u(n_a), kb = f2u(n_b);
    const int sa = (XI && SBX_HC_MAGIC_SLOT) ? (int)(f2u(n_a + 12582912.0f) & (unsigned)(HC_SLOTS - 1)) : ((int)n_a & (HC_SLOTS - 1));
    const H8 h = pl_lookup<XI, false>(S, tab, ka, sa, on, lane);
    const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz, gb = 1.0f - fb;
    // the first point, and of the second whatever does not depend on the differing axis (hc_blend's operations, its order)
    if (AX == 0) {
        na = hc_blend(h.lo, h.hi, fx, fy, fz);
        nb = hc_blend(h.lo, h.hi, fb, fy, fz);
    } else {
        const float a = h.lo.x * gx + h.lo.y * fx, b = h.lo.z * gx + h.lo.w * fx;
        const float c = h.hi.x * gx + h.hi.y * fx, d = h.hi.z * gx + h.hi.w * fx;
        if (AX == 1) {
            na = (a * gy + b * fy) * gz + (c * gy + d * fy) * fz;
            nb = (a * gb + b * fb) * gz + (c * gb + d * fb) * fz;
        } else {
            const float m = a * gy + b * fy, n = c * gy + d * fy;
            na = m * gz + n * fz;
            nb = m * gb + n * fb;
        }
    }
    // lanes whose two points lie in different cells: the second point's own cell
    const bool other = ka != kb;
    if (wave_any(on && other)) {
        const int sb = (XI && SBX_HC_MAGIC_SLOT) ? (int)(f2u(n_b + 12582912.0f) & (unsigned)(HC_SLOTS - 1)) : ((int)n_b & (HC_SLOTS - 1));
        const H8 g = pl_lookup<XI, false>(S, tab, kb, sb, on && other, lane);
        const float v = hc_blend(g.lo, g.hi, AX == 0 ? fb : fx, AX == 1 ? fb : fy, AX == 2 ? fb : fz);
        if (other) nb = v;
    }
}
// octaves START .. OCT-1 of an fBm for both points of a pair: t += basis(noise) * H; p *= lacunarity; H *= gain   (fbm.h:6)
template <int OCT, int MODE, int START, int AX, bool XI, int TB>
__device__ __forceinline__ void pl_fbm_pair(WaveCache& S, v3& qa, float& qb_ax, float lacunarity, float& H, float gain, float& ta, float& tb,
                                            bool on, int lane) {
#pragma unroll 1
    for (int k = START; k < OCT; ++k) {
        float na, nb;
        pl_noise_pair<AX, XI>(S, qa, qb_ax, (TB + k) & 3, on, lane, na, nb);
        ta += pl_basis<MODE>(na) * H;
        tb += pl_basis<MODE>(nb) * H;
        qa = qa * lacunarity;
        qb_ax = qb_ax * lacunarity;
        H *= gain;
    }
}
// sdf_terrain_map_detail(pos + d).x - sdf_terrain_map_detail(pos - d).x with d = e along axis AX   (:188-199, :201-212)
template <int AX, bool SKIP>
__device__ __forceinline__ float terrain_detail_difference(WaveCache& S, v3 pos, float e, bool on, int lane) {
    // the two points: pos.c + e and pos.c - e on axis AX; pos.c + 0 = pos.c - 0 = pos.c on the others (no zero coordinates here)
    const float ca = (AX == 0 ? pos.x : AX == 1 ? pos.y : pos.z) + e, cb = (AX == 0 ? pos.x : AX == 1 ? pos.y : pos.z) - e;
    const v3 pa = V3(AX == 0 ? ca : pos.x, AX == 1 ? ca : pos.y, AX == 2 ? ca : pos.z);
    const v3 pb = V3(AX == 0 ? cb : pos.x, AX == 1 ? cb : pos.y, AX == 2 ? cb : pos.z);
    // first fBm: fbm(pos * 2.0987, 2.0244, .454, .454), 7 octaves of noise
    v3 q = pa * 2.0987f;
    float qb = cb * 2.0987f;
    float H = .454f, h0a = 0.f, h0b = 0.f;
    pl_fbm_pair<7, 0, 0, AX, SKIP, 0>(S, q, qb, 2.0244f, H, .454f, h0a, h0b, on, lane);
    const float n0a = SMOOTHSTEP_K(.35f, 1.f, h0a), n0b = SMOOTHSTEP_K(.35f, 1.f, h0b);
    // second fBm (ridged): pos * 1.50987 + (1.9489, 2.435, .5483); the tail bound of terrain_map for both points
    const v3 off = V3(1.9489f, 2.435f, .5483f);
    q = pa * 1.50987f + off;
    qb = cb * 1.50987f + (AX == 0 ? off.x : AX == 1 ? off.y : off.z);
    H = .454f;
    float h1a = 0.f, h1b = 0.f;
    pl_fbm_pair<1, 2, 0, AX, SKIP, PL_TB2>(S, q, qb, 2.0244f, H, .454f, h1a, h1b, on, lane);
    float n1a = 0.f, n1b = 0.f;
    if (!SKIP || wave_any(on && !(h1a + .3745f * 1.0001f < .6f && h1b + .3745f * 1.0001f < .6f))) {
        pl_fbm_pair<7, 2, 1, AX, SKIP, PL_TB2>(S, q, qb, 2.0244f, H, .454f, h1a, h1b, on, lane);
        n1a = SMOOTHSTEP_K(.6f, 1.f, h1a);
        n1b = SMOOTHSTEP_K(.6f, 1.f, h1b);
    }
    const float va = pl_length<SKIP>(pa) - 1.f - (n0a + n1a) * PL_MAX_HEIGHT;
    const float vb = pl_length<SKIP>(pb) - 1.f - (n0b + n1b) * PL_MAX_HEIGHT;
    return va - vb;
}

__device__ __forceinline__ v3 setup_lights(v3 L, v3 normal) {                          // :217-236
    v3 diffuse = V3(0, 0, 0);
    diffuse = diffuse + fmax_(0.f, dot(L, normal)) * V3(7, 5, 3);
    const float hemi = clamp_(.25f + .5f * normal.y, .0f, 1.f);
    diffuse = diffuse + hemi * V3(.4f, .6f, .8f) * .2f;
    const float amb = clamp_(.12f + .8f * fmax_(0.f, dot(-L, normal)), 0.f, 1.f);
    diffuse = diffuse + amb * V3(.4f, .5f, .6f);
    return diffuse;
}

__device__ __forceinline__ v3 planet_background(v3 dir) {                              // :23-41
    const v3 sun_color = V3(1.f, .9f, .55f);
    const float sun_amount = clamp_(dot(dir, V3(0, 0, 1)), 0.f, 1.f);
    v3 sky = mix3(V3(.0f, .05f, .2f), V3(.15f, .3f, .4f), 1.0f - dir.y);
    sky = sky + sun_color * clamp_(pow_(sun_amount, 30.0f) * 5.0f, 0.f, 1.f);
    sky = sky + sun_color * clamp_(pow_(sun_amount, 10.0f) * .6f, 0.f, 1.f);
    return abs3(sky);
}

// Occupancy (7680x4320, tools/ab_time.py): 4 waves/SIMD with batches of 4 octaves: 128 VGPRs, 34 spills, 8.02 ms (round 1);
// batches of 2: 126 VGPRs, NO spills, 8.00 ms; held to 3 waves: 9.65 ms; 5 waves (96 VGPRs, 32-slot tables so that LDS allows
// them): 30 spills, some inside the cloud march, and still 7.3 ms — the fifth wave pays more than the spills cost (parking the
// terrain results in LDS across the cloud march removed a third of them and changed nothing).
#ifndef PL_MIN_WAVES
#define PL_MIN_WAVES 5
#endif
#ifndef PL_TW
#define PL_TW 8              // wave tile PL_TW x 64/PL_TW pixels
#endif
#ifndef PL_PARK
#define PL_PARK 1          // the hit shading's state waits in LDS while the 6 detail terrain maps run
#endif
// (6 waves per SIMD, 80 VGPRs, no parking — its LDS does not fit beside 32-slot tables then: 7.03 ms, but 176 B of scratch per lane,
//  now inside the marches, and 4.3 GB of HBM traffic per frame: 4 % of time bought with 4x the traffic; not taken.  7 waves with
//  16-slot tables: 7.21 ms.)
// (tools/ab_time.py, 7680x4320, same bits: no parking 7.36 ms, 112 B of scratch per lane, 1.72 GB of HBM traffic per frame
//  against the 0.53 GB framebuffer; hit-shading parking 7.29 ms, 60 B, ~1.0 GB; parking the cloud march's ray and integrator
//  as well: no spills there to remove, 7.66 ms; recomputing pixel and ray in the epilogue instead of keeping them: 7.54 ms)
#ifndef PL_PARK_MARCH
#define PL_PARK_MARCH 0    // the two results of the terrain march that only the shading reads (height term, t of the last step) in LDS
#endif                     // slots 10-11 during both marches instead of in registers
#define PL_PARK_N (PL_PARK_MARCH ? 12 : 10)
// ATM: the config-5 composite SBX_APP_PLANET_ATMOSPHERE (include/sbx.h; SURVEY.md §8a note: "planet background() replaced by
// get_incident_light"): wherever APP_PLANET shows its background() (app_planet.h:316-318, 364-366) the pixel shows APP_ATMOSPHERE's
// sky instead — get_incident_light (app_atmosphere.h:78-160) for a ray from 1 m above the ground (:204-207) along the VIEW
// direction, with APP_ATMOSPHERE's sun (setup_scene :177-181, FramePlanet.atm_sun).  No reference-held answers: parity unpinned.
// The sky runs k_atmosphere's exhaustively-equal forms (atm_incident_light<true>: exp_reg4k_, div3_, sqrt_rs_) in the SKIP kernels
// (tame frames: finite uniforms) where their domains hold for THIS camera too (round 5): the ray starts at (0, Re + 1, 0) whatever
// its direction, so every sample lies inside the atmosphere sphere and the exp arguments stay in [-50.1, 5300] (sbx_atmosphere.h);
// and |s|^2 of a march position cannot be 0 or tiny because the view direction normalize(pc.x, pc.y, 1) has rd.z > 0 — a wave
// checks rd.z > 1e-15 for its pixels (s.z = rd.z t with t >= 4e5 on a ray that can come near the centre, so |s|^2 >= 1e-19 against
// sqrt_rs_'s 2^-102) and takes the plain statement of the spec otherwise (fragCoords 1e15 frames below the frame, through the point
// list).  9.73 -> 8.75 ms came from the planet's own marches; this: see profiles/r05_log.md.
// (Reading the marches' two rotations from LDS at every step, so that their 15 multiply-adds per step have no SGPR source — half rate
//  on gfx950 — was measured: 5.763 against 5.759 ms, nothing; profiles/r05_log.md.)
#ifndef PL_ATM_FIN
#define PL_ATM_FIN 1
#endif
template <bool SKIP, bool ATM = false>
__global__ void __launch_bounds__(WG_THREADS, PL_MIN_WAVES) k_planet(FramePlanet F, RowMap M, float* __restrict__ out) {
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();      // (the dispatch order's cost table, RowMap.cost)
    __shared__ double etab[ATM ? 32 : 1];
    if (ATM) {
        if (threadIdx.x < 32) etab[threadIdx.x & (ATM ? 31 : 0)] = kExp2Tab[threadIdx.x];
        if (WG_THREADS > 64) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    }
    __shared__ WaveCache cache[WG_THREADS / 64];
#if PL_PARK
    __shared__ float park[WG_THREADS / 64][PL_PARK_N * 64];
#endif
    const int lane = threadIdx.x & 63;
    WaveCache& S = cache[threadIdx.x >> 6];
    hc_init(S, lane);
    const Pixel px = pixel_of_thread<PL_TW>(M);
    const v2 pc = point_cam(F.cam, px.fx, px.fy);
    const v3 ro = F.cam.eye, rd = primary_dir(F.cam, pc);

    // intersect_sphere(eye, atmosphere = {0, 1 + max_height}) on no_hit      intersect.h:7-33
    bool hit_atm = false;
    v3 hit_o = V3(0, 0, 0);
    {
        const float radius = 1.f + PL_MAX_HEIGHT;
        const v3 rc = V3(0, 0, 0) - ro;
        const float radius2 = radius * radius;
        const float tca = dot(rc, rd);
        if (!(tca < 0.f)) {
            const float d2 = dot(rc, rc) - tca * tca;
            if (!(d2 > radius2)) {
                const float thc = sqrt_(radius2 - d2);
                float t0 = tca - thc;
                const float t1 = tca + thc;
                if (t0 < 0.f) t0 = t1;
                if (!(t0 > (float)(1e8f + 1e1f))) { hit_atm = true; hit_o = ro + rd * t0; }
            }
        }
    }
    hit_atm = hit_atm && px.valid;
    v3 col = V3(0, 0, 0);
    bool cloud_sky = false;                                      // atmosphere hit, no ground hit: background under the clouds
    float sky_r = 0.f, sky_a = 0.f;
    if (wave_any(hit_atm)) {
        // terrain march :328-342
        float t = 0.f;
        v2 df = V2(1, PL_MAX_HEIGHT);
        v3 pos = V3(0, 0, 0);
#ifndef PL_TLAST
#define PL_TLAST 1         // the march keeps the t of its last committed step instead of the step's point (one register for three, across
#endif                     // both marches); the point is computed again — same operations, same operands — where the hit is shaded
        float t_pos = 0.f;
        float max_cld = PL_MAX_RAY_DIST;
        bool tm = hit_atm;
        if (SKIP && PL_SPEC) hc_set_direction(S, mul(F.rot, rd), hit_atm, lane);       // both terrain fBms scale the rotated point by positive factors
        for (int i = 0; i < 120; ++i) {
            if (tm && t > PL_MAX_RAY_DIST) tm = false;           // `if (t > max_ray_dist) break;`
            if (!wave_any(tm)) break;
            const v3 o = hit_o + t * rd;
            const v3 p = mul(F.rot, o - V3(0, 0, 0));
            const v2 d = terrain_map<3, SKIP>(S, p, tm, lane);
            if (tm) {
#if PL_PARK && PL_PARK_MARCH && PL_TLAST
                {
                    volatile float* const pm = &park[threadIdx.x >> 6][lane];
                    pm[10 * 64] = d.y; pm[11 * 64] = t;
                    df.x = d.x;
                }
#else
                if (PL_TLAST) t_pos = t; else pos = p;
                df = d;
#endif
                if (df.x < .005f) { max_cld = t; tm = false; }
                else t += df.x * .4567f;
            }
        }
        // clouds_march :121-141 on construct_volume(hit.origin)
        Vol cloud = make_vol(hit_o);
        {
            const float t_step = PL_MAX_RAY_DIST / 75.f;
            float tc = 0.f;
            bool cm = hit_atm;
            if (SKIP && PL_SPEC) hc_set_direction(S, mul(F.rot_cloud, rd), hit_atm, lane);
            for (int i = 0; i < 75; ++i) {
                if (cm && (tc > max_cld || cloud.alpha >= 1.f)) cm = false;   // `return` of clouds_march
                if (!wave_any(cm)) break;
                const v3 o = cloud.origin + tc * rd;
                // The band (.2, .65) of height = (|p| - 1) / .4 is the shell 1.08 < |p| < 1.26.  |o|^2 outside
                // (1.166, 1.5885) puts the rotated point outside it with a margin (3e-4 relative) far above the rounding
                // of rotation, sqrt and division, so clouds_map() would find band() == +0 and change nothing: when that
                // holds for every committing lane the step is only `tc += t_step` (most steps of the march).
                {
                    const float d2 = dot(o, o);
                    if (SKIP && !wave_any(cm && !(d2 > 1.5885f || d2 < 1.166f))) { tc += t_step; continue; }
                }
                const v3 cp = mul(F.rot_cloud, o - V3(0, 0, 0));
                const float ch = PL_DIVK(pl_length<SKIP>(cp) - 1.f, PL_MAX_HEIGHT);
                if (cm) { cloud.pos = cp; cloud.height = ch; }
                tc += t_step;
                clouds_map<SKIP>(S, cloud, t_step, cm, lane);
            }
        }
        const bool hitl = hit_atm && (df.x < .005f);              // :349
        if (SKIP && PL_SPEC) hc_no_direction(S, lane);              // the hit shading samples around a point, not along a ray
        v3 c_hit = V3(0, 0, 0);
        if (wave_any(hitl)) {
#if PL_PARK && PL_PARK_MARCH && PL_TLAST
            {
                volatile float* const pm = &park[threadIdx.x >> 6][lane];
                df.y = pm[10 * 64]; t_pos = pm[11 * 64];           // (lanes without an atmosphere hit never wrote them and are not shaded)
            }
#endif
            if (PL_TLAST) pos = mul(F.rot, (hit_o + t_pos * rd) - V3(0, 0, 0));   // the point of the last committed step, as the march computed it
            // illuminate :238-298
            float h = df.y;
            const float e = 0.001f;
            float g[3];                                            // sdf_terrain_normal :201-212
#if PL_PARK
            // 6 x (7 + 7)-octave terrain_map: the second register peak.  What it does not need — the hit point, its height
            // term, the ray, the cloud integrator (already in slots 1-2) — waits in LDS, and the three differences land there.
            float* const pk = &park[threadIdx.x >> 6][lane];
            pk[3 * 64] = pos.x; pk[4 * 64] = pos.y; pk[5 * 64] = pos.z; pk[6 * 64] = h;
            pk[1 * 64] = cloud.radiance; pk[2 * 64] = cloud.alpha;
            asm volatile("" ::: "memory");
            // (a zero coordinate of the hit point: pos.c + 0 and pos.c - 0 then differ in the sign of their zero — the unpaired path)
            const bool pairs = SKIP && PL_PAIRS && !wave_any(hitl && (pk[3 * 64] == 0.f || pk[4 * 64] == 0.f || pk[5 * 64] == 0.f));
            if (pairs) {
                pk[7 * 64] = terrain_detail_difference<0, SKIP>(S, V3(pk[3 * 64], pk[4 * 64], pk[5 * 64]), e, hitl, lane);
                asm volatile("" ::: "memory");
                pk[8 * 64] = terrain_detail_difference<1, SKIP>(S, V3(pk[3 * 64], pk[4 * 64], pk[5 * 64]), e, hitl, lane);
                asm volatile("" ::: "memory");
                pk[9 * 64] = terrain_detail_difference<2, SKIP>(S, V3(pk[3 * 64], pk[4 * 64], pk[5 * 64]), e, hitl, lane);
                asm volatile("" ::: "memory");
            } else {
#pragma unroll 1
            for (int ax = 0; ax < 3; ++ax) {                       // one code copy for the three central differences
                const v3 d = V3(ax == 0 ? e : 0.f, ax == 1 ? e : 0.f, ax == 2 ? e : 0.f);
                const float va = terrain_map<7, SKIP>(S, V3(pk[3 * 64], pk[4 * 64], pk[5 * 64]) + d, hitl, lane).x;
                asm volatile("" ::: "memory");
                const float vb = terrain_map<7, SKIP>(S, V3(pk[3 * 64], pk[4 * 64], pk[5 * 64]) - d, hitl, lane).x;
                pk[(7 + ax) * 64] = va - vb;
                asm volatile("" ::: "memory");
            }
            }
            pos = V3(pk[3 * 64], pk[4 * 64], pk[5 * 64]);
            h = pk[6 * 64];
            cloud.radiance = pk[1 * 64]; cloud.alpha = pk[2 * 64];
            g[0] = pk[7 * 64]; g[1] = pk[8 * 64]; g[2] = pk[9 * 64];
#else
#pragma unroll 1
            for (int ax = 0; ax < 3; ++ax) {                       // one code copy for the three central differences
                const v3 d = V3(ax == 0 ? e : 0.f, ax == 1 ? e : 0.f, ax == 2 ? e : 0.f);
                const float v = terrain_map<7, SKIP>(S, pos + d, hitl, lane).x - terrain_map<7, SKIP>(S, pos - d, hitl, lane).x;
                if (ax == 0) g[0] = v; else if (ax == 1) g[1] = v; else g[2] = v;
            }
#endif
            const v3 w_normal = normalize(pos);
            const v3 normal = normalize(V3(g[0], g[1], g[2]));
            const float N = dot(normal, w_normal);
            const v3 c_water = V3(.015f, .110f, .455f), c_grass = V3(.086f, .132f, .018f),
                     c_beach = V3(.153f, .172f, .121f), c_rock = V3(.080f, .050f, .030f),
                     c_snow = V3(.600f, .600f, .600f);
            const float l_water = .05f, l_shore = .17f, l_grass = .211f, l_rock = .351f;
            const float s = smoothstep_(.4f, 1.f, h);
            const v3 rock = mix3(c_rock, c_snow, smoothstep_(1.f - .3f * s, 1.f - .2f * s, N));
            const v3 grass = mix3(c_grass, rock, smoothstep_(l_grass, l_rock, h));
            v3 shoreline = mix3(c_beach, grass, smoothstep_(l_shore, l_grass, h));
            const v3 water = mix3(c_water / 2.f, c_water, smoothstep_(0.f, l_water, h));
            shoreline = shoreline * setup_lights(F.L, normal);
            const v3 ocean = setup_lights(F.L, w_normal) * water;
            const v3 c_terr = mix3(ocean, shoreline, smoothstep_(l_water, l_shore, h));

            const float c_cld = cloud.radiance, alpha = cloud.alpha;
            // cloud shadow on the ground :355-360, clouds_shadow_march :143-160
            const v3 lp = mul(F.rot_t, pos);
            Vol sh = make_vol(lp);
            const v3 local_up = normalize(lp);
            {
                const float t_step = PL_MAX_HEIGHT / 5.f;
                float ts = 0.f;
                for (int i = 0; i < 5; ++i) {
                    const v3 o = sh.origin + ts * local_up;
                    sh.pos = mul(F.rot_cloud, o - V3(0, 0, 0));
                    sh.height = PL_DIVK(pl_length<SKIP>(sh.pos) - 1.f, PL_MAX_HEIGHT);
                    ts += t_step;
                    clouds_map<SKIP>(S, sh, t_step, hitl, lane);
                }
            }
            const float shadow = mix_(.7f, 1.f, step_(sh.alpha, 0.33f));
            c_hit = abs3(mix3(c_terr * shadow, V3s(c_cld), alpha));
        }
        cloud_sky = !hitl;
        sky_r = cloud.radiance; sky_a = cloud.alpha;
        col = c_hit;
    }
    const Pixel pxe = px;
    const v3 rde = rd;
    if (!pxe.valid) return;
    if (ATM) {
        if (cloud_sky || !hit_atm) {
            v3 sky;
            if (SKIP && PL_ATM_FIN && !wave_any(!(rde.z > 1e-15f)))       // (wave-uniform; a NaN direction takes the plain form)
                sky = atm_incident_light<true>(V3(0, ATM_EARTH_R + 1.f, 0), rde, F.atm_sun, reinterpret_cast<const double (&)[32]>(etab), nullptr);
            else
                sky = atm_incident_light<false>(V3(0, ATM_EARTH_R + 1.f, 0), rde, F.atm_sun, reinterpret_cast<const double (&)[32]>(etab), nullptr);
            col = hit_atm ? abs3(mix3(sky, V3s(sky_r), sky_a)) : sky;
        }
    } else {
        if (cloud_sky) col = abs3(mix3(planet_background(rde), V3s(sky_r), sky_a));   // :364-366
        if (!hit_atm) col = planet_background(rde);                  // :316-318
    }
    tile_cost_store(M, tl_t0);
    store_rgba(M, out, pxe.idx, to_srgb(col));
}

dim3 planet_grid(const RowMap& M) { return grid_for<PL_TW>(M); }

void launch_planet(const FramePlanet& F, const RowMap& M, float* out, hipStream_t s, int variant) {
    if (F.atm_sky) {
        if (variant == 1) hipLaunchKernelGGL((k_planet<false, true>), grid_for<PL_TW>(M), dim3(WG_THREADS), 0, s, F, M, out);
        else hipLaunchKernelGGL((k_planet<true, true>), grid_for<PL_TW>(M), dim3(WG_THREADS), 0, s, F, M, out);
        return;
    }
    if (variant == 1) hipLaunchKernelGGL(k_planet<false>, grid_for<PL_TW>(M), dim3(WG_THREADS), 0, s, F, M, out);
    else hipLaunchKernelGGL(k_planet<true>, grid_for<PL_TW>(M), dim3(WG_THREADS), 0, s, F, M, out);
}

hipError_t bind_fault_planet(unsigned* word) { return hc_bind_fault_word(word); }

}  // namespace sbx
