// shaderbox_amd/csrc/kern_clouds.hip — APP_CLOUDS: volumetric fBm cloud integrator.
//
// Follows /root/reference/src/app_clouds.h (SKY_SPHERE / USE_NOISE_TEX undefined, :8-9):
// render :204-218, render_sky_color :36-46, render_clouds :153-202, integrate_volume :125-148,
// illuminate_volume :91-123, density_func :62-86 over fbm = 4 octaves of noise_iq (:59),
// henyey_greenstein_phase_func src/volumetric.h:27-33 with hg_g = .2 (:5).
// 16-slot hash tables: with the exp table (256 B) and the parked march state (3 KB) a wave needs 5.7 KB of LDS, so 6 waves
// per SIMD (24 per CU) fit in the 160 KB.  64, 32 and 16 slots time the same at equal occupancy (profiles/r02_clouds_ab.txt):
// a wave rarely holds more than a few cells per octave.
#ifndef SBX_HC_SLOTS
#define SBX_HC_SLOTS 16
#endif
#include "sbx_device.h"
#include "sbx_noise.h"
#include "sbx_hashcache.h"
#include "sbx_exp4k_table.h"
#include <cmath>
#include <cstdlib>

#ifndef CL_PARK
#define CL_PARK 1          // park the march state in LDS during a lit step's light march
#endif
#define CL_PARK_N 12
#ifndef CL_LIPSKIP
#define CL_LIPSKIP 1       // skip main samples proved clear by the Lipschitz bound (coop_density_row)
#endif
#ifndef CL_LIPSKIP2
#define CL_LIPSKIP2 1      // the Lipschitz skip also where the SECOND stage fails
#endif
#ifndef CL_EPILOGUE_RELOAD
#define CL_EPILOGUE_RELOAD 1
#endif
#ifndef CL_EXP_ASM
#define CL_EXP_ASM 1
#endif
#ifndef CL_SEED
#define CL_SEED 1
#endif
// (Every CL_* switch below is an A/B switch whose OTHER setting must still compile and render the same bits: `tools/ab_build.py
//  --all-variants` builds each one, `tools/sweep_clouds_variants.py --all` runs the random-frame sweep against the per-lane kernel
//  on them — profiles/r04_clouds_variants.txt: 22 variants, 0 differing frames.  Switches that changed pixels on purpose are gone.)
#ifndef CL_NO_REG
#define CL_NO_REG 0       // 1: never use the REG kernels (A/B timing)
#endif
#ifndef CL_EXP_LDS
#define CL_EXP_LDS 1       // exp_ reads its table from LDS (0: from __constant__ memory through the vector L1)
#endif
#if CL_EXP_LDS
#define CL_EXP(x) exp_tab_((x), etab)
#else
#define CL_EXP(x) exp_(x)
#endif
#ifndef CL_NOTAB_GEN
#define CL_NOTAB_GEN 1       // 1: the table-less kernels (more march steps than the y table's rows) at CL_MIN_WAVES_GEN too
                             //    (1100 steps without a table at 4K: 48.1 ms at 6 waves, 42.7 at 5)
#endif
#ifndef CL_MIN_WAVES_GEN
#define CL_MIN_WAVES_GEN 5   // the instantiations with the general light march (a sun off the z axis): 3840x2160 6.20 ms at 6 waves (92 B of
                             // scratch per lane), 5.31 at 5 (28 B), 5.53 at 4 (104 VGPRs, none)
#endif
#ifndef CL_MAX3
#define CL_MAX3 1
#endif

#ifndef CL_MIN_WAVES_YZ
#define CL_MIN_WAVES_YZ 5    // the instantiations with the y-z light march
#endif
#ifndef CL_YZ_MARCH
#define CL_YZ_MARCH 1        // 0: suns in the y-z plane take the general march (A/B timing)
#endif
#ifndef CL_MIN_WAVES
#define CL_MIN_WAVES 6     // waves per SIMD the register allocation is held to (__launch_bounds__): 80 VGPRs (78 used, no spills;
                           // 7 waves = 72 VGPRs spill 6 and lose 4 %)
#endif

namespace sbx {

// The REG kernels' exp (|sigma * dt| <= 80 shown on the host per launch) is one of the guard-less forms of sbx_math.h that are equal
// to exp_ on their whole domain: exp_small_ in the SM instantiations (every argument in [-0.205, -0], launch_clouds),
// exp_reg4k_ in the others.  The CL_EXP* switches below keep the older forms (exp_reg64_, exp_reg_, exp_tab_<false>) for A/B timing.
#ifndef CL_EXP64
#define CL_EXP64 1         // REG kernels: the 64-entry / degree-5 form of exp_reg_ (one binary64 fma less; exhaustively equal on |x| <= 80)
#endif
#ifndef CL_EXP_SMALL
#define CL_EXP_SMALL 1     // the SM instantiations (REG kernels with the y table, z-only or general light march): exp_small_ (no reduction, no
#endif                     // table: 10 half-rate instructions instead of 17) on frames whose exp arguments all lie in [-0.205, -0] (decided per
                           // launch, launch_clouds; the default frame's reach -0.1758): 3840x2160 2.637 -> 2.527 ms, same bits
#ifndef CL_EXP_SMALL_ASM
#define CL_EXP_SMALL_ASM 1
#endif
#ifndef CL_YZ_SM
#define CL_YZ_SM 1        // exp_small_ in the y-z light march's kernel as well
#endif
#ifndef CL_DIV3
#define CL_DIV3 1          // SM kernels: the coverage smoothstep's division through div3_ (sbx_math.h)
#endif
#ifndef CL_EXP4K
#define CL_EXP4K 1         // the REG kernels outside exp_small_'s domain: exp_reg4k_ (4096-entry table through the vector L1, degree 3)
#endif                     // instead of exp_reg64_ (64 entries in LDS, degree 5)
#define CL_USES_4K (CL_EXP_ASM && CL_EXP64 && CL_EXP_SMALL && CL_EXP4K)
#if CL_EXP_ASM && CL_EXP64 && CL_EXP_SMALL
#if CL_EXP4K
#define CL_EXP_REG(x) (SM ? exp_small_<CL_EXP_SMALL_ASM != 0>(x) : exp_reg4k_((x), kExp2Tab4096))     // SM: a template parameter in scope
#else
#define CL_EXP_REG(x) (SM ? exp_small_<CL_EXP_SMALL_ASM != 0>(x) : exp_reg64_((x), etab))
#endif
#elif CL_EXP_ASM && CL_EXP64
#define CL_EXP_REG(x) exp_reg64_((x), etab)
#elif CL_EXP_ASM
#define CL_EXP_REG(x) exp_reg_((x), etab)
#else
#define CL_EXP_REG(x) exp_tab_<false>((x), etab)
#endif

__device__ __forceinline__ float clouds_density(const FrameClouds& F, v3 pos_in) {
    v3 pos = pos_in * F.nf;                                    // cld_noise_factor, :18,20,66 (.001 unless SKY_SPHERE)
    float shape = fbm<4>(pos * 2.03f, 2.64f, .5f, .5f, [](v3 p) { return noise_iq(p); });   // :72
    return shape * smoothstep_(F.cov, F.cov_hi, shape);        // :83-84 (per-lane cross-check kernel keeps the IEEE division)
}

__device__ __forceinline__ float hg_phase(float mu, float g) {  // volumetric.h:27-33, note (4 + PI)
    return (1.f - g * g) / ((4.f + 3.14159265359f) * pow_(1.f + g * g - 2.f * g * mu, 1.5f));
}

// illuminate_volume :91-123.  `phase` = henyey_greenstein(clamp(dot(L,V),0,1)) depends only on
// the pixel's ray, so the caller evaluates it once per pixel (same inputs, same bits).
__device__ __forceinline__ float clouds_illuminate(const FrameClouds& F, v3 origin, float phase) {
    const v3 step = F.sun_dir * F.dt;
    v3 pos = origin + step;
    float transmittance = 1.f;
    for (int i = 0; i < F.lsteps; ++i) {
        float density = clouds_density(F, pos);
        transmittance *= exp_(-density * F.sigma * F.dt);
        pos = pos + step;
    }
    return transmittance * F.sun_power * phase;
}

// ---------------------------------------------------------------------------------------------
// variant 1 ("per-lane"): every lane evaluates all 8 lattice hashes of every noise cell itself.
// Kept as the in-library cross-check of the cooperative kernel below (sbx_set_variant(ctx, 1)).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WG_THREADS) k_clouds_perlane(FrameClouds F, RowMap M, float* __restrict__ out) {
    const Pixel px = pixel_of_thread<32>(M);
    if (!px.valid) return;
    const v2 pc = point_cam(F.cam, px.fx, px.fy);
    const v3 dir = primary_dir(F.cam, pc);

    // render_sky_color :36-46
    float sun_amount = fmax_(dot(dir, F.sun_dir), 0.f);
    v3 sky = mix3(V3(.0f, .1f, .4f), V3(.3f, .6f, .8f), 1.0f - dir.y);
    sky = sky + F.sun_color * fmin_(pow_(sun_amount, 1500.0f) * 5.0f, 1.0f);
    sky = sky + F.sun_color * fmin_(pow_(sun_amount, 10.0f) * .6f, 1.0f);
    sky = abs3(sky);

    const float cutoff = dot(dir, V3(0, 1, 0));
    v3 col = sky;
    if (!(cutoff < 0.05f)) {                                   // :212
        // render_clouds :153-202
        v3 projection = dir / dir.y;
        v3 origin = F.cam.eye + projection * 150.f;
        origin = origin + F.wind_off;
        if (F.sky) {                                           // #ifdef SKY_SPHERE :154-162, intersect_sphere_from_inside intersect.h:35-53
            const v3 ac = V3(0, F.atm_y, 0);
            const v3 rc = ac - F.cam.eye;
            const float radius2 = F.atm_r * F.atm_r;
            const float tca = dot(rc, dir);
            const float d2 = dot(rc, rc) - tca * tca;
            const float thc = sqrt_(radius2 - d2);
            const float t0 = tca - thc;
            const v3 impact = F.cam.eye + dir * t0;
            projection = dir;
            origin = mul(F.sky_rot, impact - ac);
        }
        const float phase = hg_phase(clamp_(dot(F.sun_dir, dir), 0.f, 1.f), .2f);
        float transmittance = 1.f, radiance = 0.f, alpha = 0.f, t = 0.f;
        for (int i = 0; i < F.steps; ++i) {
            const v3 pos = origin + t * projection;
            t += F.dt;
            const float density = clouds_density(F, pos);
            if (!(density < .005f)) {                          // integrate_volume :132
                const float T_i = exp_(-density * F.sigma * F.dt);
                transmittance *= T_i;
                radiance += (density * F.sigma) * clouds_illuminate(F, pos, phase) * transmittance * F.dt;
                alpha += (1.f - T_i) * (1.f - alpha);
            }
            if (alpha > .999f) break;
        }
        const float a = alpha * smoothstep_(.0f, .2f, cutoff);
        col = abs3(mix3(sky, V3s(radiance), a));               // :215-217 (radiance is r=g=b)
    }
    store_rgba(M, out, px.idx, to_srgb(col));
}

// ---------------------------------------------------------------------------------------------
// variant 0 (default): per-wave lattice-hash cache in LDS.
//
// hash(n) depends only on the integer lattice index n = p.x + 157 p.y + 113 p.z (noise_iq.h:19), and
// the 64 rays of a wave's pixel tile sample almost the same place: measured on the 3840x2160 frame (8x8 tiles) a
// wave touches on average 1.07 / 1.18 / 1.49 / 2.25 distinct lattice cells in octaves 0..3 per sample,
// and consecutive samples along the march (and the six light samples of a lit step) mostly stay in the
// cells of the previous sample.  So the 8 corner hashes of a cell are computed ONCE per wave and kept
// in a small direct-mapped table in LDS (per octave 64 slots of {tag = bits of n, 8 hashes}, 9 KB per
// wave), keyed by slot = int(n) & 63:
//   hit  (the common case): one ds_read_b32 for the tag, a compare, two ds_read_b128 for the 8 hashes
//         (lanes of a tile mostly read the same slot: LDS broadcast), then the lane's own blend;
//   miss: the missing cells are inserted by the wave together — leader election over the missing lanes
//         with ballot/readlane (one cell per slot per round), then ONE pass in which lane (r, c)
//         evaluates corner c of pending cell r (8 cells x 8 corners = 64 lanes) — and the lookup repeats.
//         A lane always reads its hashes before a later insertion can evict them.
// Each hash is the same function of the same binary32 argument as in the per-lane variant, so the
// result is bit-identical by construction; only redundant evaluations are removed (on the 4K frame the
// per-lane variant evaluates 32 sin per lane per density_func; this one ~0.1).  Control flow around the
// cross-lane steps is wave-uniform: lanes never exit early, they carry `alive`/`lit` predicates.
// (The first cooperative version re-elected the distinct cells of every sample: 11.6 ms per 4K frame,
// of which 2.3 ms election and 2.0 ms hash passes)
// ---------------------------------------------------------------------------------------------
// ---- y terms of the main march --------------------------------------------------------------------
// render_clouds marches along projection = dir / dir.y (:165), whose y component is dir.y / dir.y = 1 exactly,
// from origin.y = (eye.y + 150) + wind.y — the same for every pixel.  So the sample height of march step i,
// pos.y = origin.y + t_i, is a frame constant, and with it, per octave, floor/fract/smoothstep of y and the
// term 157 * floor(y) of the lattice index.  They are computed once per frame by k_clouds_ytab (one thread
// per step, the same operations a lane would do) into a small table that the march reads with scalar loads:
// the values arrive in SGPRs and ~9 VALU instructions per octave per sample disappear from every lane.
// The z light march of a lit step (x, y unchanged) uses the same row.
struct YRow { float4 fy, gy, py157; };     // per march step: 4 octaves each (fy, gy scaled by the octave's gain, see k_clouds_ytab)
#ifndef CL_PRESCALE
#define CL_PRESCALE 1
#endif
__device__ __forceinline__ constexpr float hk_(int k) { return CL_PRESCALE ? (k == 0 ? .5f : (k == 1 ? .25f : (k == 2 ? .125f : .0625f))) : 1.0f; }


__global__ void __launch_bounds__(64) k_clouds_ytab(FrameClouds F, YRow* __restrict__ tab) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= F.steps) return;
    float t = 0.f;
    for (int j = 0; j < i; ++j) t += F.dt;                    // t after i steps of `t += dt` (:186)
    const float origin_y = (F.cam.eye.y + 1.0f * 150.f) + F.wind_off.y;   // :166-167 with projection.y = 1
    const float y = origin_y + t * 1.0f;                      // :185
    float q = (y * .001f) * 2.03f;                            // :66,72
    float fy[4], gy[4], p157[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float py = floor_(q);
        const float ay = q - py;
        fy[k] = ay * ay * tm2_(ay);
        gy[k] = 1.0f - fy[k];
        p157[k] = py * 157.0f;
        q = q * 2.64f;
    }
    // The y weights are stored PRE-SCALED by the octave's gain H = 2^-(k+1) (CL_PRESCALE): the y-mixes a gy + b fy then come out
    // scaled, and with them the octave's term (ab gz + cd fz) H of fbm's `t += noise * H` — bit for bit, because a power-of-two
    // scale commutes with every rounding on the way (see light_march_z for the subnormal corner) — so neither the main sample nor
    // the z light march multiplies by H.
    tab[i].fy = make_float4(fy[0] * hk_(0), fy[1] * hk_(1), fy[2] * hk_(2), fy[3] * hk_(3));
    tab[i].gy = make_float4(gy[0] * hk_(0), gy[1] * hk_(1), gy[2] * hk_(2), gy[3] * hk_(3));
    tab[i].py157 = make_float4(p157[0], p157[1], p157[2], p157[3]);
}

// NFL: the noise factor is the literal .001 (the y-table kernels, which never run SKY_SPHERE frames); else F.nf
template <bool NFL, bool B40 = false>          // B40: lattice indices shown below 2^40 on the host (sbx_noise.h hash1_b)
__device__ __forceinline__ float coop_density(const FrameClouds& F, v3 pos_in, bool active, WaveCache& S, int lane) {
    v3 p = (pos_in * (NFL ? .001f : F.nf)) * 2.03f;      // :66,72
    float fx[4], fy[4], fz[4];
    unsigned nbits[4];
    int slot[4];
    bool ne[4];
    bool miss = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                        // lattice part of noise_iq.h:14-19, all octaves first
        const float px = floor_(p.x), py = floor_(p.y), pz = floor_(p.z);
        const float ax = p.x - px, ay = p.y - py, az = p.z - pz;
        fx[k] = ax * ax * tm2_(ax);
        fy[k] = ay * ay * tm2_(ay);
        fz[k] = az * az * tm2_(az);
        const float n = px + py * 157.0f + 113.0f * pz;
        nbits[k] = f2u(n);
        slot[k] = (int)n & (HC_SLOTS - 1);
        ne[k] = (S.tag[k][slot[k]] != nbits[k]);         // the four tag reads issue back to back
        miss |= ne[k];
        p = p * 2.64f;                                   // fbm.h:6  p *= lacunarity
    }
    float t = 0.f, H = .5f;
    if (!wave_any(active && miss)) {
        // every active lane finds all four cells cached: straight-line reads + blends
        float4 lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[k] = *reinterpret_cast<const float4*>(&S.h[k][slot[k]][0]);
            hi[k] = *reinterpret_cast<const float4*>(&S.h[k][slot[k]][4]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t = __builtin_fmaf(hc_blend(lo[k], hi[k], fx[k], fy[k], fz[k]), H, t);    // noise * 2^-(k+1) is exact: `t += noise * H` in one rounding
            H *= .5f;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            H8 h;
            if (wave_any(active && ne[k])) {
                h = hc_slow<B40>(S, k, nbits[k], slot[k], active, lane);
            } else {
                h.lo = *reinterpret_cast<const float4*>(&S.h[k][slot[k]][0]);
                h.hi = *reinterpret_cast<const float4*>(&S.h[k][slot[k]][4]);
            }
            t = __builtin_fmaf(hc_blend(h.lo, h.hi, fx[k], fy[k], fz[k]), H, t);
            H *= .5f;
        }
    }
    return t * smoothstep_rd(F.cov, F.cov_rd, t);        // :83-84
}

// x/y half of the trilinear blend (the four x-mixes and two y-mixes of noise_iq.h:20-23)
__device__ __forceinline__ void hc_blend_xy(float4 lo, float4 hi, float fx, float fy, float gy, float& ab, float& cd) {
    const float gx = 1.0f - fx;
    const float a = lo.x * gx + lo.y * fx;
    const float b = lo.z * gx + lo.w * fx;
    const float c = hi.x * gx + hi.y * fx;
    const float d = hi.z * gx + hi.w * fx;
    ab = a * gy + b * fy;
    cd = c * gy + d * fy;
}

// Octaves [K0, K1) of a main sample's fBm: lattice terms, tag checks, ONE wave-uniform all-hit test, reads, blends.
template <int K0, int K1, bool B40>
__device__ __forceinline__ void row_octaves(const float (&rfy)[4], const float (&rgy)[4], const float (&rpy)[4], float& qx, float& qz,
                                            float& t, float& H, bool active, unsigned long long active_mask, WaveCache& S,
                                            int lane, float (&fx)[4], float (&nxy)[4], float (&mab)[4], float (&mcd)[4],
                                            float (&mpz)[4]) {
    float fz[4];
    unsigned nbits[4];
    int slot[4];
    bool ne[4];
    unsigned long long miss_mask = 0;
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        const float px = floor_(qx), pz = floor_(qz);
        const float ax = qx - px, az = qz - pz;
        fx[k] = ax * ax * tm2_(ax);
        fz[k] = az * az * tm2_(az);
        nxy[k] = px + rpy[k];
        mpz[k] = pz;
        const float n = nxy[k] + 113.0f * pz;            // p.x + p.y*157 + 113*p.z, noise_iq.h:19
        nbits[k] = f2u(n);
        slot[k] = (int)n & (HC_SLOTS - 1);
        ne[k] = (S.tag[k][slot[k]] != nbits[k]);
        miss_mask |= wave_mask(ne[k]);
        qx = qx * 2.64f; qz = qz * 2.64f;
    }
    if (!wave_any_mask(miss_mask & active_mask)) {
        float4 lo[4], hi[4];
#pragma unroll
        for (int k = K0; k < K1; ++k) {
            lo[k] = *reinterpret_cast<const float4*>(&S.h[k][slot[k]][0]);
            hi[k] = *reinterpret_cast<const float4*>(&S.h[k][slot[k]][4]);
        }
#pragma unroll
        for (int k = K0; k < K1; ++k) {
            hc_blend_xy(lo[k], hi[k], fx[k], rfy[k], rgy[k], mab[k], mcd[k]);      // rfy, rgy carry the octave's gain (k_clouds_ytab)
            t += CL_PRESCALE ? (mab[k] * (1.0f - fz[k]) + mcd[k] * fz[k]) : (mab[k] * (1.0f - fz[k]) + mcd[k] * fz[k]) * H;
            H *= .5f;
        }
    } else {
#pragma unroll
        for (int k = K0; k < K1; ++k) {
            H8 h;
            if (wave_any(active && ne[k])) {
                h = hc_slow<B40>(S, k, nbits[k], slot[k], active, lane);
            } else {
                h.lo = *reinterpret_cast<const float4*>(&S.h[k][slot[k]][0]);
                h.hi = *reinterpret_cast<const float4*>(&S.h[k][slot[k]][4]);
            }
            hc_blend_xy(h.lo, h.hi, fx[k], rfy[k], rgy[k], mab[k], mcd[k]);
            t += CL_PRESCALE ? (mab[k] * (1.0f - fz[k]) + mcd[k] * fz[k]) : (mab[k] * (1.0f - fz[k]) + mcd[k] * fz[k]) * H;
            H *= .5f;
        }
    }
}

// density_func of a MAIN march sample with the y terms of its step taken from the frame table (SGPRs).
// The value is only used when it is >= .005 (integrate_volume :132), and shape * smoothstep(cov, cov + .0135, shape)
// is exactly +0 for shape <= cov.  The noise of an octave lies in [0, 1], so after octaves 0-1 the full sum is at
// most t + .125 + .0625 and after octave 2 at most t + .0625 (plus rounding, covered by the margins): when that
// bound is below cov for EVERY alive lane of the wave the sample cannot be lit and the remaining octaves are not
// evaluated (54 % / 73 % of the main samples of the 4K frame; the skipped octaves are the ones that miss most).
// A skipped sample returns 0, which integrate_volume treats exactly like the true value.
//
// Skipping clear air (LIP, REG kernels only).  When the first stage already fails — s + .1876 < cov with s = .5 N(q0) + .25 N(q1)
// the sum of octaves 0-1 — the NEXT samples of the march are often clear as well, and that can be PROVED without evaluating
// them.  The value noise N is a trilinear blend, with smoothstep weights S(a) = a^2 (3 - 2a), |S'| <= 1.5, of lattice hashes
// in [0, 1], hence 1.5-Lipschitz in every coordinate and continuous across cells.  One march step moves the sample by
// dt * (proj.x, 1, proj.z), i.e. by dt * .00203 * (|proj.x| + 1 + |proj.z|) =: D in the L1 norm of q0 and by 2.64 D in q1, so
//     |s(i + j) - s(i)| <= j * 1.5 * (.5 + .25 * 2.64) * D = j * 1.74 D   (+ a few 1e-6 of binary32 rounding).
// With gap = (cov - .1876 - 1e-3) - s(i) and c = 1.01 * 1.74 D (rounded up, per lane, fixed for the march), the next
// floor(gap / c) samples of that lane satisfy the stage-1 test; the minimum over the alive lanes (taken as the largest of
// 1, 2, 4, 8, 16 that every alive lane allows) is the number of main samples the wave skips entirely: each would have
// returned density 0, which integrate_volume ignores (src/app_clouds.h:132), so only `t += dt` remains of them.
// DOMAIN of the proof (clouds_lip_domain, checked on the host per launch; F.lip_ok = 0 turns the skip off and nothing else).
// The bound is about the real-valued noise; the kernel evaluates it at ROUNDED positions.  With every march position
// |pos.{x,y,z}| <= 2^17 (ulp <= 2^-6): pos, pos * .001 and * 2.03 each round by half an ulp, i.e. q0 is off by <= 5e-5 and
// q1 = 2.64 q0 by <= 1.6e-4, the lattice index n <= 271 * 2^17 * .00203 * 2.64^3 = 1.3e6 < 2^24 is an exact integer (cells
// share their corner hashes, the blend is continuous across faces), and s moves by <= 1.5 * 2 * (.5 * 5e-5 + .25 * 1.6e-4)
// = 2e-4 per end point (s3: 4e-4): the 1e-3 in the gaps covers both ends, the 1 % in c the rounding of t and D.  Positions
// are eye + proj * 150 + wind_dir * u_time * 1000 + t * proj with |proj.x|, |proj.z| <= 20 (dir.y >= .05) — unbounded in
// u_time and wind_dir, both caller-set: at |wind_off| = 2e7 (ulp 2) the bound fails numerically, hence the check.
template <bool LIP, bool B40, bool D3>          // D3: the coverage smoothstep through div3_ (vcd, vcr = F.cov_d, F.cov_r in VGPRs)
__device__ __forceinline__ float coop_density_row(const FrameClouds& F, v3 pos_in, const YRow& row, bool active,
                                                  unsigned long long active_mask, WaveCache& S, int lane,
                                                  float (&fx)[4], float (&nxy)[4], float (&mab)[4], float (&mcd)[4],
                                                  float (&mpz)[4], const float* lip_slot, int& skip, float vcd, float vcr) {
    float qx = (pos_in.x * .001f) * 2.03f, qz = (pos_in.z * .001f) * 2.03f;
    const float rfy[4] = {row.fy.x, row.fy.y, row.fy.z, row.fy.w};
    const float rgy[4] = {row.gy.x, row.gy.y, row.gy.z, row.gy.w};
    const float rpy[4] = {row.py157.x, row.py157.y, row.py157.z, row.py157.w};
    float t = 0.f, H = .5f;
    float lip_inv = 0.f;
    if (LIP) lip_inv = *lip_slot;                        // 1 / c of this lane, kept in LDS (read early, used after the first stage)
    row_octaves<0, 2, B40>(rfy, rgy, rpy, qx, qz, t, H, active, active_mask, S, lane, fx, nxy, mab, mcd, mpz);
    // (t + .1876 < cov written as t < cov - .1876 with the right side a frame constant: one instruction; the 1e-4 of slack over
    //  .1875 covers the half ulp by which the two forms can differ, and a skipped sample is exactly 0 either way)
    if (!wave_any_mask(active_mask & wave_mask(!(t < F.thr1)))) {                        // NaN compares false: goes on
        if (LIP) {
            float c = F.cov;
            asm volatile("" : "+v"(c));                  // (keeps cov - .1886 from being hoisted into a register held for the whole march)
            const float r = ((c - .1886f) - t) * lip_inv;                                // samples this lane can prove clear
            if (!wave_any_mask(active_mask & wave_mask(!(r >= 1.f)))) {
                skip = 1;
                if (!wave_any_mask(active_mask & wave_mask(!(r >= 4.f)))) {
                    skip = 4;
                    if (!wave_any_mask(active_mask & wave_mask(!(r >= 16.f)))) skip = 16;
                    else if (!wave_any_mask(active_mask & wave_mask(!(r >= 8.f)))) skip = 8;
                } else if (!wave_any_mask(active_mask & wave_mask(!(r >= 2.f)))) skip = 2;
            }
        }
        return 0.f;
    }
#ifdef SBX_CL_STATS
    if (lane == 0) S.stat[4] += 1.f;
#endif
    row_octaves<2, 3, B40>(rfy, rgy, rpy, qx, qz, t, H, active, active_mask, S, lane, fx, nxy, mab, mcd, mpz);
    if (!wave_any_mask(active_mask & wave_mask(!(t < F.thr2)))) {
#if CL_LIPSKIP2
        if (LIP) {
            // the same proof one octave further: s3 = s + .125 N(q2) moves by at most j * 1.5 * (.5 + .25 * 2.64 + .125 * 2.64^2) D
            // = j * 3.0468 D over j steps, and a sample with s3 + .06255 < cov is clear.  1 / c3 = (1 / c) * 1.74 / 3.0468
            // (.571, rounded down); gap = (cov - .06255 - 1e-3) - s3.
            float c = F.cov;
            asm volatile("" : "+v"(c));
            const float r = (((c - .06355f) - t) * *lip_slot) * .571f;
            if (!wave_any_mask(active_mask & wave_mask(!(r >= 1.f)))) {
                skip = 1;
                if (!wave_any_mask(active_mask & wave_mask(!(r >= 2.f)))) {
                    skip = 2;
                    if (!wave_any_mask(active_mask & wave_mask(!(r >= 4.f)))) skip = 4;
                }
            }
        }
#endif
        return 0.f;
    }
#ifdef SBX_CL_STATS
    if (lane == 0) S.stat[5] += 1.f;
#endif
    row_octaves<3, 4, B40>(rfy, rgy, rpy, qx, qz, t, H, active, active_mask, S, lane, fx, nxy, mab, mcd, mpz);
    return D3 ? t * smoothstep_d3(F.cov, vcd, vcr, t) : t * smoothstep_rd(F.cov, F.cov_rd, t);        // :83-84
}

// illuminate_volume's march (:106-113) when the light step L*dt has no x and no y component — the
// reference's default sun_dir (0,0,-1), src/uniform_buffer.h:42.  Then pos.x + 0 and pos.y + 0 never change
// along the light march, so for every octave floor/fract/smoothstep of x and y, the partial lattice index
// p.x + 157 p.y, and — as long as the sample stays in the same lattice cell — the x- and y-mixes of the
// blend are the SAME binary32 values for all light samples of the step.  They are computed once; per
// sample only the z terms, the cell lookup and the final z-mix remain.  Every value is produced by the
// same operations on the same inputs as in the general path, hence identical bits.
// REG ("regular frame", decided on the host per launch: launch_clouds): the coverage edge and its reciprocal are finite and
// |sigma * dt| <= 80.  Then (a) exp_'s binary32 range guard is the identity for every argument the march can produce
// (-d * sigma * dt with d in [0, 1) or NaN) and is left out, (b) the smoothstep clamp is one v_med3_f32
// (sbx_math.h x_smoothstep_rd_med3), and (c) "did the sample leave its lattice cell" is decided without a floor:
// az = qz - curz is the fract the reference computes iff floor(qz) == curz, and
//     0 <= RN(qz - curz) < 1   ==>   curz <= qz < curz + 1   ==>   floor(qz) == curz
// (a float difference is never rounded to zero or across zero; a difference >= 1 is never rounded below 1), so ONE
// unsigned compare of az's bits against those of 1.0f replaces v_floor + v_cmp (both half-rate instructions on gfx950,
// profiles/r02_ubench_issue.txt).  The test is conservative (a fract that rounds up to 1.0 is treated as a move);
// a move recomputes floor / fract / lookups exactly as the general form does.
template <bool YTAB, bool REG, bool SM>
__device__ __forceinline__ float light_march_z(const FrameClouds& F, v3 lp, v3 lstep, bool lit, unsigned long long lit_mask,
                                               WaveCache& S, int lane, const YRow& row, const float (&mfx)[4],
                                               const float (&mnxy)[4], const float (&mab)[4], const float (&mcd)[4],
                                               const float (&mpz)[4], const double* etab, float vsigma, float vdt,
                                               float vcov, float vcd, float vcr) {       // vcd, vcr: F.cov_d, F.cov_r in VGPRs (SM)
    float fx[4], fy[4], gy[4], nxy[4], ab[4], cd[4], curz[4];
    if (YTAB) {
        // lp.x = pos.x + 0 and lp.y = pos.y + 0: the x terms are the main sample's own (same operations on
        // the same value; a -0 turned +0 by the addition changes no result bit), the y terms
        // are the step's row of the frame table.
        const float rfy[4] = {row.fy.x, row.fy.y, row.fy.z, row.fy.w};
        const float rgy[4] = {row.gy.x, row.gy.y, row.gy.z, row.gy.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            fx[k] = mfx[k]; fy[k] = rfy[k]; gy[k] = rgy[k]; nxy[k] = mnxy[k];
            // The march starts in the lattice cells of the MAIN sample of the step (same x, same y, z one light step away):
            // its x/y blends ab, cd — hc_blend_xy of the same hashes with the same fx, fy — and its floor(z) are the current cell
            // of every octave, so the first light sample needs a lookup only where it has left that cell, like every later one
            // (a lit main sample has always evaluated all four octaves).
            curz[k] = CL_SEED ? mpz[k] : u2f(0x7fc00001u);
            ab[k] = mab[k]; cd[k] = mcd[k];                        // already scaled by the octave's gain: the row's fy, gy are
        }
    } else {
        const float nf0 = F.nf;                              // (YTAB = false: .001, or SKY_SPHERE's factor)
        float qx = (lp.x * nf0) * 2.03f, qy = (lp.y * nf0) * 2.03f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float px = floor_(qx);
            const float ax = qx - px;
            fx[k] = ax * ax * tm2_(ax);
            const float py = floor_(qy);
            const float ay = qy - py;
            fy[k] = ay * ay * tm2_(ay);
            gy[k] = 1.0f - fy[k];
            nxy[k] = px + py * 157.0f;
            curz[k] = u2f(0x7fc00001u);
            ab[k] = cd[k] = 0.f;
            qx = qx * 2.64f; qy = qy * 2.64f;
        }
    }
    float ltrans = 1.f;
    for (int j = 0; j < F.lsteps; ++j) {
        // z terms of the sample.  The cell of octave k is the one of the previous sample iff floor(z) is (nxy is
        // fixed), and then ab/cd still hold its x/y blends whatever happened to the cache since: no lookup at all.
        float az[4], pzv[4], qzv[4];
        unsigned long long moved_mask = 0, mk[4];
        float qz = (lp.z * (YTAB ? .001f : F.nf)) * 2.03f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            qzv[k] = qz;
            if (REG) {
                az[k] = qz - curz[k];                         // general form's first sample: curz is NaN -> az NaN -> "moved"
#if !CL_MAX3
                mk[k] = wave_mask(f2u(az[k]) >= 0x3f800000u);              // not (+0 <= az < 1)
#else
                mk[k] = 0;
#endif
            } else {
                const float pz = floor_(qz);
                az[k] = qz - pz;
                pzv[k] = pz;
                mk[k] = wave_mask(pz != curz[k]);             // first sample: curz is NaN, always true
            }
            moved_mask |= mk[k];
            qz = qz * 2.64f;
        }
#if CL_MAX3
        if (REG) {
            // "some octave left its cell" from ONE compare: the largest of the four bit patterns (two v_max instead of three more
            // v_cmp, all half-rate, and no mask arithmetic); which octaves, only when that happened
            unsigned mx, m4;
            asm("v_max3_u32 %0, %1, %2, %3" : "=v"(mx) : "v"(f2u(az[0])), "v"(f2u(az[1])), "v"(f2u(az[2])));
            asm("v_max_u32 %0, %1, %2" : "=v"(m4) : "v"(mx), "v"(f2u(az[3])));
            moved_mask = wave_mask(m4 >= 0x3f800000u);
        }
#endif
        if (wave_any_mask(moved_mask & lit_mask)) {
#if CL_MAX3
            if (REG) {
#pragma unroll
                for (int k = 0; k < 4; ++k) mk[k] = wave_mask(f2u(az[k]) >= 0x3f800000u);
            }
#endif
#ifdef SBX_CL_STATS
            if (lane == 0) S.stat[3] += 1.f;
#endif
            // some lit lane has left its cell in some octave: look up again — and redo the x/y blends of — exactly the octaves
            // in which that happened (mostly the finest one); the others keep their cell and their blends
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!wave_any_mask(mk[k] & lit_mask)) continue;
                if (REG) {                                    // the general form's floor / fract, for every lane
                    pzv[k] = floor_(qzv[k]);
                    az[k] = qzv[k] - pzv[k];
                }
                const float n = nxy[k] + 113.0f * pzv[k];
                const unsigned nbits = f2u(n);
                const int slot = (int)n & (HC_SLOTS - 1);
                const bool ne = (S.tag[k][slot] != nbits);
                curz[k] = pzv[k];
                H8 h;
                if (wave_any(lit && ne)) {
                    h = hc_slow<SM>(S, k, nbits, slot, lit, lane);
                } else {
                    h.lo = *reinterpret_cast<const float4*>(&S.h[k][slot][0]);
                    h.hi = *reinterpret_cast<const float4*>(&S.h[k][slot][4]);
                }
                hc_blend_xy(h.lo, h.hi, fx[k], fy[k], gy[k], ab[k], cd[k]);
                if (!YTAB) { ab[k] *= hk_(k); cd[k] *= hk_(k); }          // (the table's fy, gy carry the gain already)
            }
        }
        float t = 0.f, H = .5f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float fz = az[k] * az[k] * tm2_(az[k]);
            const float gz = 1.0f - fz;
            // fbm's `t += noise * H` with H = 2^-(k+1) folded into the kept x/y blends: (ab gz + cd fz) H == (ab H) gz + (cd H) fz
            // bit for bit — a power-of-two scale commutes with every rounding on the way — unless an intermediate is subnormal,
            // i.e. below 2^-122; a term that small is either absorbed by the other octaves' terms or, if they are all that
            // small, t < 2^-98 and d = t * smoothstep(..) underflows to exactly +-0 either way.  One multiply less per octave
            // and sample (CL_PRESCALE = 0: the scale per sample).
            const float term = CL_PRESCALE ? (ab[k] * gz + cd[k] * fz) : (ab[k] * gz + cd[k] * fz) * H;
            // fbm's t = 0; t += ...: a blend of hashes in [0, 1) with weights in [0, 1] is >= +0 (or NaN), and 0 + x == x then
            t = (REG && k == 0) ? term : t + term;
            H *= .5f;
        }
        if (REG) {
            // SM: the coverage smoothstep's division through div3_ (three full-rate instructions; launch_clouds checks its domain)
            const float d = (SM && CL_DIV3) ? x_smoothstep_d3_med3(vcov, vcd, vcr, t) : x_smoothstep_rd_med3(vcov, F.cov_rd, t);
            ltrans *= CL_EXP_REG(-d * vsigma * vdt);
        } else {
            const float d = t * smoothstep_rd(F.cov, F.cov_rd, t);
            ltrans *= CL_EXP(-d * F.sigma * F.dt);
        }
        lp = lp + lstep;
    }
    return ltrans;
}

// illuminate_volume's march (:106-113) when the light step L * dt has no x component (REG + YTAB kernels) — a sun anywhere in the
// y-z plane: the default (0, 0, -1) raised or lowered, the "sun elevation" edit of hlsltoy's panel.  lp.x = pos.x + 0 never changes,
// so per octave the x fract / smoothstep weight are the main sample's (mfx, as in light_march_z) and, while a light sample stays
// in its lattice cell, the four x-mixes of the blend (noise_iq.h:20-21)
//     a = h000 gx + h100 fx,  b = h010 gx + h110 fx,  c = h001 gx + h101 fx,  d = h011 gx + h111 fx
// are the SAME binary32 values for every sample: they are kept in registers (16), and a sample costs the y and z terms, the
// two y-mixes and the z-mix: ~140 instructions against ~250 for the general march and 85 for the z-only one.  "Still in the cell"
// is decided as in light_march_z, for y and z: a = q - cur is the reference's fract iff 0 <= RN(q - cur) < 1, one unsigned compare
// of the larger bit pattern against 1.0f.  The kept x-mixes are register values: nothing the cache does can invalidate them.
// A sample that leaves a cell looks that octave up in the general form (floor / index / tag check / cooperative insert).
template <bool REG, bool SM>
__device__ __forceinline__ float light_march_yz(const FrameClouds& F, v3 lp, v3 lstep, bool lit, unsigned long long lit_mask,
                                                WaveCache& S, int lane, const float (&mfx)[4], const double* etab,
                                                float vsigma, float vdt, float vcov, float vcd, float vcr) {
    float xa[4], xb[4], xc[4], xd[4], cy[4], cz[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { cy[k] = cz[k] = u2f(0x7fc00001u); xa[k] = xb[k] = xc[k] = xd[k] = 0.f; }
    const float qx0 = (lp.x * .001f) * 2.03f;             // the same for every sample of the march (lp.x + 0)
    float ltrans = 1.f;
    for (int j = 0; j < F.lsteps; ++j) {
        float ay[4], az[4], qyv[4], qzv[4];
        unsigned m[4];
        float qy = (lp.y * .001f) * 2.03f, qz = (lp.z * .001f) * 2.03f;                  // :66,72
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            qyv[k] = qy; qzv[k] = qz;
            ay[k] = qy - cy[k]; az[k] = qz - cz[k];        // first sample: cur is NaN -> "moved"
            asm("v_max_u32 %0, %1, %2" : "=v"(m[k]) : "v"(f2u(ay[k])), "v"(f2u(az[k])));
            qy = qy * 2.64f; qz = qz * 2.64f;                                            // fbm.h:6
        }
        unsigned m3, mall;
        asm("v_max3_u32 %0, %1, %2, %3" : "=v"(m3) : "v"(m[0]), "v"(m[1]), "v"(m[2]));
        asm("v_max_u32 %0, %1, %2" : "=v"(mall) : "v"(m3), "v"(m[3]));
        if (wave_any_mask(wave_mask(mall >= 0x3f800000u) & lit_mask)) {                  // some lit lane left a cell in y or z
#ifdef SBX_CL_STATS
            if (lane == 0) S.stat[3] += 1.f;
#endif
            float qx = qx0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float qxk = qx;
                qx = qx * 2.64f;
                if (!wave_any_mask(wave_mask(m[k] >= 0x3f800000u) & lit_mask)) continue;
                // the general form for every lane (noise_iq.h:14-19)
                const float px = floor_(qxk), py = floor_(qyv[k]), pz = floor_(qzv[k]);
                ay[k] = qyv[k] - py; az[k] = qzv[k] - pz;
                cy[k] = py; cz[k] = pz;
                const float n = px + py * 157.0f + 113.0f * pz;
                const unsigned nbits = f2u(n);
                const int slot = (int)n & (HC_SLOTS - 1);
                const bool ne = (S.tag[k][slot] != nbits);
                H8 h;
                if (wave_any(lit && ne)) {
                    h = hc_slow<SM>(S, k, nbits, slot, lit, lane);
                } else {
                    h.lo = *reinterpret_cast<const float4*>(&S.h[k][slot][0]);
                    h.hi = *reinterpret_cast<const float4*>(&S.h[k][slot][4]);
                }
                const float fx = mfx[k], gx = 1.0f - fx;
                xa[k] = h.lo.x * gx + h.lo.y * fx;
                xb[k] = h.lo.z * gx + h.lo.w * fx;
                xc[k] = h.hi.x * gx + h.hi.y * fx;
                xd[k] = h.hi.z * gx + h.hi.w * fx;
            }
        }
        float t = 0.f, H = .5f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float fy = ay[k] * ay[k] * tm2_(ay[k]);
            const float fz = az[k] * az[k] * tm2_(az[k]);
            const float gy = 1.0f - fy, gz = 1.0f - fz;
            const float ab = xa[k] * gy + xb[k] * fy;
            const float cd = xc[k] * gy + xd[k] * fy;
            const float nz = ab * gz + cd * fz;
            t = (REG && k == 0) ? nz * H : __builtin_fmaf(nz, H, t);   // 0 + x == x for x >= +0 or NaN (see light_march_z); noise * 2^-(k+1)
                                                                       // is exact, so `t += noise * H` is one rounding: an fma
            H *= .5f;
        }
        if (REG) {
            // SM: the coverage smoothstep's division through div3_ (three full-rate instructions; launch_clouds checks its domain)
            const float d = (SM && CL_DIV3) ? x_smoothstep_d3_med3(vcov, vcd, vcr, t) : x_smoothstep_rd_med3(vcov, F.cov_rd, t);
            ltrans *= CL_EXP_REG(-d * vsigma * vdt);
        } else {
            const float d = t * smoothstep_rd(F.cov, F.cov_rd, t);
            ltrans *= CL_EXP(-d * F.sigma * F.dt);
        }
        lp = lp + lstep;
    }
    return ltrans;
}

// The REG kernels' exp (cl_exp) as a standalone function for the exhaustive equivalence test against exp_ (sbx_math_eval
// "exp_reg"; tests/test_gpu_round2.py): the same LDS table, the same instruction sequence as inside k_clouds.
template <bool ASM>
__global__ void __launch_bounds__(256) k_cl_exp_eval(const float* __restrict__ a, float* __restrict__ out, size_t n) {
    __shared__ double etab[32];
    if (threadIdx.x < 32) etab[threadIdx.x] = kExp2Tab[threadIdx.x];
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = ASM ? exp_reg_<true>(a[i], etab) : exp_reg_<false>(a[i], etab);     // three-address asm form / the compiler's chain
}
template <bool ASM>
__global__ void __launch_bounds__(256) k_cl_exp64_eval(const float* __restrict__ a, float* __restrict__ out, size_t n) {
    __shared__ double etab[64];
    if (threadIdx.x < 64) etab[threadIdx.x] = kExp2Tab64[threadIdx.x];
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = exp_reg64_<ASM>(a[i], etab);
}
template <bool ASM>
__global__ void __launch_bounds__(256) k_cl_exp_small_eval(const float* __restrict__ a, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = exp_small_<ASM>(a[i]);
}
void launch_cl_exp_eval(const float* a, float* out, size_t n, hipStream_t s, int form) {     // 0 exp_reg asm, 1 plain, 2 / 3 the 64-entry forms,
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);                                // 4 / 5 exp_small_
    if (form == 4) hipLaunchKernelGGL(k_cl_exp_small_eval<true>, grid, block, 0, s, a, out, n);
    else if (form == 5) hipLaunchKernelGGL(k_cl_exp_small_eval<false>, grid, block, 0, s, a, out, n);
    else if (form == 1) hipLaunchKernelGGL(k_cl_exp_eval<false>, grid, block, 0, s, a, out, n);
    else if (form == 2) hipLaunchKernelGGL(k_cl_exp64_eval<true>, grid, block, 0, s, a, out, n);
    else if (form == 3) hipLaunchKernelGGL(k_cl_exp64_eval<false>, grid, block, 0, s, a, out, n);
    else hipLaunchKernelGGL(k_cl_exp_eval<true>, grid, block, 0, s, a, out, n);
}

// sky colour of a view direction (render_sky_color :36-46)
__device__ __forceinline__ v3 clouds_sky(const FrameClouds& F, v3 dir) {
    float sun_amount = fmax_(dot(dir, F.sun_dir), 0.f);
    v3 sky = mix3(V3(.0f, .1f, .4f), V3(.3f, .6f, .8f), 1.0f - dir.y);
    sky = sky + F.sun_color * fmin_(pow_(sun_amount, 1500.0f) * 5.0f, 1.0f);
    sky = sky + F.sun_color * fmin_(pow_(sun_amount, 10.0f) * .6f, 1.0f);
    return abs3(sky);
}

// Waves per workgroup. Waves never talk to each other (one LDS cache per wave), so the workgroup is only a
// scheduling unit: measured at 4K, 1 wave/WG 5.26 ms, 2 5.57 ms, 4 5.48 ms per launch (profiles/r01_tile_shapes.txt);
// single-wave workgroups also drain best at the end of a launch (8-rank strips 0.63 vs 0.71 ms/frame pipelined).
#ifndef CL_TX
#define CL_TX 1
#endif
#ifndef CL_TOP_FIRST
#define CL_TOP_FIRST false  // measured: top rows first is SLOWER (4.16 vs 4.01 ms): the heaviest tiles are the ones just above the horizon
#endif
#ifndef CL_TW
#define CL_TW 32          // wave tile CL_TW x 64/CL_TW pixels (profiles/r01_tile_shapes.txt)
#endif
// ZL: the light step has no x and no y component (decided on the host), so the light march is light_march_z; the general
// march is then not even compiled into the kernel (it was the register-pressure peak of the hot loop).
// The epilogue reads the camera / sky constants AGAIN from the kernarg segment (a fresh pointer the compiler cannot connect
// with the loads of the prologue) instead of keeping ~40 SGPRs live across the march: the march was at the 102-SGPR limit,
// with SGPRs spilled to VGPR lanes.  ClArgs mirrors the kernel's argument list: the kernarg segment lays the arguments out in
// order at their natural alignment, exactly like this struct.  (Passing ONE struct argument instead made every use in the
// march a scalar load: 3.5 -> 4.0 ms.)
struct ClArgs { FrameClouds F; RowMap M; float* out; const YRow* ytab; };
// LM, the light march (decided on the host from L * dt): 1 = no x and no y component (light_march_z), 2 = no x component
// (light_march_yz; YTAB kernels only), 0 = general (coop_density per light sample)
template <bool YTAB, bool REG, int LM, bool SM = false>   // SM: exp_small_ (launch_clouds: the frame's exp arguments lie in its domain)
__global__ void __launch_bounds__(64 * CL_TX, (LM == 1 && (YTAB || !CL_NOTAB_GEN)) ? CL_MIN_WAVES : (LM == 2 ? CL_MIN_WAVES_YZ : CL_MIN_WAVES_GEN)) k_clouds(FrameClouds F, RowMap M, float* __restrict__ out_arg,
                                                          const YRow* __restrict__ ytab) {
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();        // (for the dispatch order's cost table, RowMap.cost)
    __shared__ WaveCache cache[CL_TX];
#if CL_PARK
    // the march state a lit step does not need while its light march runs, parked in LDS for that time (a manual spill to
    // the fast memory: what the register allocator would otherwise send to scratch when the kernel is held to 96 VGPRs)
    __shared__ float park[CL_TX][CL_PARK_N][64];
#endif
    // exp's table: per-lane reads come from LDS, not from the vector L1.  REG kernels: the 64 entries of exp_reg64_ (512 B: 5.8 KB per
    // wave with the hash tables and the parked state — the 128-entry form's 6.3 KB is one allocation granule more and costs the
    // sixth wave: 2.62 -> 2.71 ms); the others: exp_'s 32.
    constexpr bool NOTAB = SM || (REG && CL_USES_4K);        // exp_small_ reads no table, exp_reg4k_ its own
    constexpr int ETAB_N = NOTAB ? 1 : (REG && CL_EXP64 && CL_EXP_ASM) ? 64 : 32;
    __shared__ double etab[ETAB_N];
    const int lane = threadIdx.x & 63;
    WaveCache& S = cache[threadIdx.x >> 6];
    if (!NOTAB) for (int i = threadIdx.x; i < ETAB_N; i += 64 * CL_TX) etab[i] = (ETAB_N == 64) ? kExp2Tab64[i] : kExp2Tab[i];
    if (CL_TX > 1) __syncthreads();
    for (int i = lane; i < 4 * HC_SLOTS; i += 64) (&S.tag[0][0])[i] = 0x7fc00001u;   // empty
#ifdef SBX_CL_STATS
    if (lane < 8) S.stat[lane] = 0.f;
#endif
    __builtin_amdgcn_wave_barrier();

#if CL_PARK
    // The integrator state of the pixel (transmittance, radiance, alpha) and its phase value change only in LIT steps (a fifth
    // of the main steps): they live in LDS slots 5-8 of the wave's park area for the whole march, not in registers.
    float* const pk = &park[threadIdx.x >> 6][0][lane];
    pk[5 * 64] = 1.f; pk[6 * 64] = 0.f; pk[7 * 64] = 0.f;
#else
    float transmittance = 1.f, radiance = 0.f, alpha = 0.f;
#endif
    bool marches;
#ifdef SBX_CL_STATS
    float st_steps = 0.f, st_lit = 0.f, st_alive = 0.f, st_litl = 0.f, st_skipped = 0.f;
#endif
    {
        // Only what the march needs stays live across it (origin, projection, phase): the view direction
        // and the sky colour are recomputed in the epilogue from the pixel coordinates — same operations,
        // same bits — which keeps the register budget of the march at 4 waves per SIMD without spills.
        const Pixel px = pixel_of_thread<CL_TW, CL_TX, CL_TOP_FIRST>(M);
        const v2 pc = point_cam(F.cam, px.fx, px.fy);
        const v3 dir = primary_dir(F.cam, pc);
        const float cutoff = dot(dir, V3(0, 1, 0));
        marches = px.valid && !(cutoff < 0.05f);                  // :212
        bool alive = marches;
        if (wave_any(alive)) {                                    // wave-uniform
            v3 projection = dir / dir.y;                          // render_clouds :153-202
            v3 origin = (F.cam.eye + projection * 150.f) + F.wind_off;
            if (!YTAB && F.sky) {                                 // #ifdef SKY_SPHERE :154-162 (wave-uniform: a kernel argument)
                // intersect_sphere_from_inside (intersect.h:35-53): t0 = tca - thc whatever its sign, NaN flows on
                const v3 ac = V3(0, F.atm_y, 0);
                const v3 rc = ac - F.cam.eye;
                const float radius2 = F.atm_r * F.atm_r;
                const float tca = dot(rc, dir);
                const float d2 = dot(rc, rc) - tca * tca;
                const float thc = sqrt_(radius2 - d2);
                const float t0 = tca - thc;
                const v3 impact = F.cam.eye + dir * t0;
                projection = dir;
                origin = mul(F.sky_rot, impact - ac);
            }
#if CL_PARK
            pk[8 * 64] = hg_phase(clamp_(dot(F.sun_dir, dir), 0.f, 1.f), .2f);
#else
            float phase = hg_phase(clamp_(dot(F.sun_dir, dir), 0.f, 1.f), .2f);
#endif
            const v3 lstep = F.sun_dir * F.dt;
            // frame constants the light march multiplies / subtracts with, held in VGPRs: an fp32 VALU instruction with an
            // SGPR source issues at half rate on gfx950 (profiles/r02_ubench_issue.txt)
            constexpr bool LIP = YTAB && REG && (CL_LIPSKIP != 0);
            // 1 / c of coop_density_row's skip bound; c = 1.01 * 1.74 * D, D = dt * .001 * 2.03 * (|proj.x| + 1 + |proj.z|)
            const float lip_inv = 1.0f / ((1.01f * 1.74f) * ((F.dt * (.001f * 2.03f)) * ((abs_(projection.x) + 1.0f) + abs_(projection.z))));
#if CL_PARK
            if (LIP) pk[9 * 64] = F.lip_ok ? lip_inv : 0.f;   // lives in LDS: it is needed only where a first stage fails
                                                              // (0: r = gap * 0 is 0 or NaN, never >= 1: no sample is skipped)
            const float* lip_slot = &pk[9 * 64];
#else
            const float lip_inv_ok = F.lip_ok ? lip_inv : 0.f;
            const float* lip_slot = &lip_inv_ok;
#endif
            int skip = 0;
            float vsigma = F.sigma, vdt = F.dt, vcov = F.cov, vcd = F.cov_d, vcr = F.cov_r;
            asm volatile("" : "+v"(vsigma), "+v"(vdt), "+v"(vcov));
            // (not in the general light march's kernels: two more live registers there mean 12 B of scratch, 4.82 -> 4.86 ms)
            constexpr bool D3 = SM && CL_DIV3 && LM != 0;
            if (D3) asm volatile("" : "+v"(vcd), "+v"(vcr));
            float t = 0.f;
            unsigned long long alive_mask = wave_mask(alive);
            for (int i = 0; i < F.steps; ++i) {
                if (!wave_any_mask(alive_mask)) break;
#ifdef SBX_CL_STATS
                st_steps += 1.f; st_alive += (float)__builtin_popcountll(alive_mask);
#endif
                if (LIP && skip > 0) {                            // proved clear for every alive lane (coop_density_row): density 0
                    --skip;
                    t += F.dt;
#ifdef SBX_CL_STATS
                    st_skipped += 1.f;
#endif
                    continue;
                }
                const v3 pos = origin + t * projection;
                t += F.dt;
                YRow row;
                if (YTAB) row = ytab[i];                          // uniform index: scalar loads (reading row i + 1 ahead
                                                                  // over the back edge costs 12 more live SGPRs: +6 % time)
                float mfx[4] = {0.f, 0.f, 0.f, 0.f}, mnxy[4] = {0.f, 0.f, 0.f, 0.f};
                float mab[4] = {0.f, 0.f, 0.f, 0.f}, mcd[4] = {0.f, 0.f, 0.f, 0.f};
                float mpz[4] = {u2f(0x7fc00001u), u2f(0x7fc00001u), u2f(0x7fc00001u), u2f(0x7fc00001u)};
                float density = YTAB ? coop_density_row<LIP, SM, D3>(F, pos, row, alive, alive_mask, S, lane, mfx, mnxy, mab, mcd, mpz, lip_slot, skip, vcd, vcr)
                                           : coop_density<false, SM>(F, pos, alive, S, lane);
                const bool lit = alive && !(density < .005f);     // integrate_volume :132
                const unsigned long long lit_mask = alive_mask & wave_mask(!(density < .005f));
                if (wave_any_mask(lit_mask)) {
#ifdef SBX_CL_STATS
                    st_lit += 1.f; st_litl += (float)__builtin_popcountll(lit_mask);
#endif
                    float T_i = REG ? CL_EXP_REG(-density * vsigma * vdt) : CL_EXP(-density * F.sigma * F.dt);
                    v3 lp = pos + lstep;                           // illuminate_volume :91-123
                    float ltrans = 1.f;
                    constexpr bool ZL = LM == 1 || (LM == 2 && YTAB);   // the marches that run with the march state parked
                    if (ZL) {                                      // lstep.x == 0 (&& lstep.y == 0) (launch_clouds)
#if CL_PARK
                        // what the march does not need while a light march runs: parked for that time (slots 0-4, 9-11)
                        pk[0 * 64] = origin.x; pk[1 * 64] = origin.z; pk[2 * 64] = projection.x; pk[3 * 64] = projection.z;
                        pk[4 * 64] = t;
                        pk[10 * 64] = density; pk[11 * 64] = T_i;
                        asm volatile("" ::: "memory");     // the reloads below cannot be forwarded from these stores: the values
                                                           // are dead across the light march
#endif
                        if (LM == 1) ltrans = light_march_z<YTAB, REG, SM>(F, lp, lstep, lit, lit_mask, S, lane, row, mfx, mnxy, mab, mcd, mpz, etab, vsigma, vdt, vcov, vcd, vcr);
                        else ltrans = light_march_yz<REG, SM>(F, lp, lstep, lit, lit_mask, S, lane, mfx, etab, vsigma, vdt, vcov, vcd, vcr);
#if CL_PARK
                        asm volatile("" ::: "memory");
                        origin.x = pk[0 * 64]; origin.z = pk[1 * 64]; projection.x = pk[2 * 64]; projection.z = pk[3 * 64];
                        t = pk[4 * 64];
                        density = pk[10 * 64]; T_i = pk[11 * 64];
#endif
                    } else {
                        for (int j = 0; j < F.lsteps; ++j) {
                            const float d = coop_density<YTAB, SM>(F, lp, lit, S, lane);
                            ltrans *= REG ? CL_EXP_REG(-d * vsigma * vdt) : CL_EXP(-d * F.sigma * F.dt);
                            lp = lp + lstep;
                        }
                    }
#if CL_PARK
                    const float illum = ltrans * F.sun_power * pk[8 * 64];
                    bool full = false;
                    if (lit) {
                        const float transmittance = pk[5 * 64] * T_i;
                        pk[5 * 64] = transmittance;
                        pk[6 * 64] = pk[6 * 64] + (density * F.sigma) * illum * transmittance * F.dt;
                        float alpha = pk[7 * 64];
                        alpha += (1.f - T_i) * (1.f - alpha);
                        pk[7 * 64] = alpha;
                        full = alpha > .999f;                     // :197 — alpha changes in lit steps only, so the exit test lives here
                    }
                    if (full) alive = false;
                    alive_mask &= ~wave_mask(full);
                }
#else
                    const float illum = ltrans * F.sun_power * phase;
                    if (lit) {
                        transmittance *= T_i;
                        radiance += (density * F.sigma) * illum * transmittance * F.dt;
                        alpha += (1.f - T_i) * (1.f - alpha);
                    }
                }
                if (alpha > .999f) alive = false;                 // :197
                alive_mask &= ~wave_mask(alpha > .999f);
#endif
            }
        }
    }
#if CL_EPILOGUE_RELOAD
    typedef const __attribute__((address_space(4))) ClArgs* KArgPtr;
    KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));                                   // a pointer of unknown origin: its loads are new loads
    const ClArgs* kg = (const ClArgs*)ka;                          // (the address space is still known: scalar loads)
    const FrameClouds& FE = kg->F;
    const RowMap& ME = kg->M;
    float* const out = kg->out;
#else
    const FrameClouds& FE = F;
    const RowMap& ME = M;
    float* const out = out_arg;
#endif
#if CL_EPILOGUE_RELOAD
    // the thread's coordinates as values of unknown origin too: otherwise the pixel arithmetic of the prologue is kept
    // alive (spilled to scratch: 20 B per lane written and read back through HBM) for the whole march
    int tid = (int)threadIdx.x, bx = (int)blockIdx.x, by = (int)blockIdx.y;
    asm volatile("" : "+v"(tid), "+s"(bx), "+s"(by));
    const Pixel px = pixel_of<CL_TW, CL_TX, CL_TOP_FIRST>(ME, tid, bx, by, (int)gridDim.y);
#else
    const Pixel px = pixel_of_thread<CL_TW, CL_TX, CL_TOP_FIRST>(ME);
#endif
    if (!px.valid) return;
    const v2 pc = point_cam(FE.cam, px.fx, px.fy);
    const v3 dir = primary_dir(FE.cam, pc);
    const v3 sky = clouds_sky(FE, dir);
    v3 col = sky;
    if (marches) {
#if CL_PARK
        const float radiance = pk[6 * 64], alpha = pk[7 * 64];
#endif
        const float a = alpha * smoothstep_(.0f, .2f, dot(dir, V3(0, 1, 0)));
        col = abs3(mix3(sky, V3s(radiance), a));               // :215-217
    }
    tile_cost_store(ME, tl_t0);
#ifdef SBX_CL_TIMES
    {   // timeline census (tools/clouds_timeline.py): lane 0 of every wave writes its start, its duration (100 MHz ticks) and where it ran
        const unsigned long long tl_t1 = __builtin_amdgcn_s_memrealtime();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        reinterpret_cast<float4*>(out)[px.idx] = make_float4(__uint_as_float((unsigned)(tl_t0 & 0xffffffffu)), __uint_as_float((unsigned)(tl_t1 - tl_t0)),
                                                             __uint_as_float(marches ? 1u : 0u), __uint_as_float(xcc & 0xfu));
        return;
    }
#endif
#ifdef SBX_CL_STATS
    __builtin_amdgcn_wave_barrier();
    if (lane == 1) { st_steps = S.stat[0]; st_lit = S.stat[1]; st_alive = S.stat[2]; st_litl = S.stat[3]; }
    if (lane == 2) { st_steps = S.stat[4]; st_lit = S.stat[5]; st_alive = st_skipped; }
    reinterpret_cast<float4*>(out)[px.idx] = make_float4(st_steps, st_lit, st_alive, st_litl);
    return;
#endif
    store_rgba(ME, out, px.idx, to_srgb(col));
}

// "regular frame" (see light_march_z): every quantity the REG shortcuts rely on is checked here, on the host, per launch
static bool clouds_regular(const FrameClouds& F) {
#if CL_NO_REG
    return false;
#endif
    const double sd = std::fabs((double)F.sigma) * std::fabs((double)F.dt);
    return std::isfinite(F.cov) && std::isfinite(F.cov_rd) && F.cov_rd > 0.0 && std::isfinite(F.sigma) && std::isfinite(F.dt) &&
           sd <= 80.0;
}

// Domain of the Lipschitz sample skip (coop_density_row): every coordinate of every MAIN march position stays within 2^17.
// pos = (eye + proj * 150 + wind_off) + t * proj, |proj.x|, |proj.z| <= 1 / .05 (+ rounding: 21), proj.y = 1,
// |t| <= steps * |dt| (+ its accumulated rounding, covered by the factor 22).  NaN / inf anywhere fails the test.
static bool clouds_lip_domain(const FrameClouds& F) {
    const double reach = 21.0 * 150.0 + 22.0 * std::fabs((double)F.dt) * (double)F.steps;
    const double e = std::fmax(std::fabs((double)F.cam.eye.x), std::fmax(std::fabs((double)F.cam.eye.y), std::fabs((double)F.cam.eye.z)));
    const double w = std::fmax(std::fabs((double)F.wind_off.x), std::fmax(std::fabs((double)F.wind_off.y), std::fabs((double)F.wind_off.z)));
    const double far = e + w + reach;
    return std::isfinite(F.wind_off.x) && std::isfinite(F.wind_off.y) && std::isfinite(F.wind_off.z) && far <= 131072.0;   // NaN compares false
}

// sin_b40_'s domain (the SM kernels' hash passes): every lattice index n = px + 157 py + 113 pz (+ up to 271) the frame can hash is
// below 2^39 in magnitude.  A sample's coordinates are bounded by |eye| + |wind_off| + the main march's reach (|dir / dir.y| <= 20
// above the horizon cut; SKY_SPHERE: the atmosphere sphere and a unit direction) + the light march's (lsteps + 1) |L dt|; the finest
// octave scales them by nf * 2.03 * 2.64^3.
// div3_'s domain for the coverage smoothstep (x - cov) / (cov_hi - cov) of the SM kernels: x is an fBm sum in [0, .9375] or NaN, so
// with 2^-20 <= |cov| <= 2^20 a non-zero x - cov is at least 2^-45 in magnitude (and at most 2^21); the divisor within 2^+-60.
static bool clouds_div3_domain(const FrameClouds& F) {
    const double c = std::fabs((double)F.cov), d = std::fabs((double)F.cov_d);
    return std::isfinite(F.cov) && std::isfinite(F.cov_d) && std::isfinite(F.cov_r) && c >= 0x1p-20 && c <= 0x1p20 && d >= 0x1p-60 &&
           d <= 0x1p60 && F.cov_d == F.cov_hi - F.cov && F.cov_r == 1.0f / F.cov_d;
}
static bool clouds_index_domain(const FrameClouds& F) {
    auto m3 = [](v3 v) { return std::fmax(std::fabs((double)v.x), std::fmax(std::fabs((double)v.y), std::fabs((double)v.z))); };
    const double reach = 21.0 * 150.0 + 22.0 * std::fabs((double)F.dt) * (double)F.steps;
    const double light = ((double)F.lsteps + 1.0) * m3(F.sun_dir) * std::fabs((double)F.dt);
    const double sky = F.sky ? std::fabs((double)F.atm_r) + std::fabs((double)F.atm_y) : 0.0;
    const double P = m3(F.cam.eye) + m3(F.wind_off) + reach + light + sky;
    const double n = 271.0 * (P * std::fabs((double)F.nf) * 2.03 * 18.4 + 2.0);
    return std::isfinite(n) && n <= 549755813888.0;                  // 2^39; NaN compares false
}

dim3 clouds_grid(const RowMap& M) { return grid_for<CL_TW, CL_TX>(M); }

void launch_clouds(const FrameClouds& F_in, const RowMap& M, float* out, hipStream_t s, int variant, void* ytab, int ytab_rows,
                   bool build_table) {
    FrameClouds F = F_in;
    F.lip_ok = clouds_lip_domain(F) ? 1 : 0;
    // exp_small_'s domain: x = -density * sigma * dt with 0 <= density <= .9375 (1 + 1e-6) (a blend of hashes in [0, 1] with weights
    // in [0, 1], gains .5 + .25 + .125 + .0625, times a smoothstep in [0, 1]) lies in [-.205, -0] when sigma, dt >= 0 and:
    F.exp_small = (F.sigma >= 0.f && F.dt >= 0.f && .94 * (double)F.sigma * (double)F.dt <= .2049) ? 1 : 0;   // NaN: 0
    F.thr1 = F.cov - .1876f;                                     // coop_density_row's stage cut-offs
    F.thr2 = F.cov - .06255f;
    const bool reg = clouds_regular(F);
    const dim3 grid = grid_for<CL_TW, CL_TX>(M), block(64 * CL_TX);
#ifdef SBX_CL_OCC_SWEEP
    // experiment only (tools/ab_time.py): extra dynamic LDS per workgroup lowers the waves resident per SIMD
    const char* pad_env = std::getenv("SBX_DEBUG_LDS_PAD");
    const unsigned pad = pad_env ? (unsigned)std::atoi(pad_env) : 0u;
#else
    const unsigned pad = 0;
#endif
    const v3 lstep = F.sun_dir * F.dt;                          // the kernel's own expression
    const bool zl = lstep.x == 0.f && lstep.y == 0.f;
    const bool yz = !zl && lstep.x == 0.f && CL_YZ_MARCH;        // NaN compares false: the general march
    // (SKY_SPHERE frames: the march does not run along dir / dir.y, so no y table — the table-less kernels, F.nf, F.sky)
    if (variant == 1) {
        hipLaunchKernelGGL(k_clouds_perlane, grid_for<32>(M), dim3(WG_THREADS), 0, s, F, M, out);
    } else if (!F.sky && ytab && F.steps <= ytab_rows && F.steps > 0) {
        YRow* tab = reinterpret_cast<YRow*>(ytab);
        if (build_table) hipLaunchKernelGGL(k_clouds_ytab, dim3((F.steps + 63) / 64), dim3(64), 0, s, F, tab);
        const YRow* ct = tab;
        // SM kernels: exp_small_'s domain AND lattice indices below 2^40 for sin_b40_
        const bool sm = F.exp_small && clouds_index_domain(F) && (!CL_DIV3 || clouds_div3_domain(F)) && CL_EXP_SMALL && CL_EXP_ASM && CL_EXP64;
        if (reg && zl && sm) hipLaunchKernelGGL((k_clouds<true, true, 1, true>), grid, block, pad, s, F, M, out, ct);
        else if (reg && zl) hipLaunchKernelGGL((k_clouds<true, true, 1>), grid, block, pad, s, F, M, out, ct);
        else if (reg && yz && sm && CL_YZ_SM) hipLaunchKernelGGL((k_clouds<true, true, 2, true>), grid, block, 0, s, F, M, out, ct);
        else if (reg && yz) hipLaunchKernelGGL((k_clouds<true, true, 2>), grid, block, 0, s, F, M, out, ct);
        else if (reg && sm) hipLaunchKernelGGL((k_clouds<true, true, 0, true>), grid, block, 0, s, F, M, out, ct);
        else if (reg) hipLaunchKernelGGL((k_clouds<true, true, 0>), grid, block, 0, s, F, M, out, ct);
        else if (zl) hipLaunchKernelGGL((k_clouds<true, false, 1>), grid, block, 0, s, F, M, out, ct);
        else if (yz) hipLaunchKernelGGL((k_clouds<true, false, 2>), grid, block, 0, s, F, M, out, ct);
        else hipLaunchKernelGGL((k_clouds<true, false, 0>), grid, block, 0, s, F, M, out, ct);
    } else {
        const YRow* ct = nullptr;
        const bool sm = F.exp_small && clouds_index_domain(F) && (!CL_DIV3 || clouds_div3_domain(F)) && CL_EXP_SMALL && CL_EXP_ASM && CL_EXP64;
        if (reg && zl && sm) hipLaunchKernelGGL((k_clouds<false, true, 1, true>), grid, block, 0, s, F, M, out, ct);
        else if (reg && sm) hipLaunchKernelGGL((k_clouds<false, true, 0, true>), grid, block, 0, s, F, M, out, ct);
        else if (reg && zl) hipLaunchKernelGGL((k_clouds<false, true, 1>), grid, block, 0, s, F, M, out, ct);
        else if (reg) hipLaunchKernelGGL((k_clouds<false, true, 0>), grid, block, 0, s, F, M, out, ct);
        else if (zl) hipLaunchKernelGGL((k_clouds<false, false, 1>), grid, block, 0, s, F, M, out, ct);
        else hipLaunchKernelGGL((k_clouds<false, false, 0>), grid, block, 0, s, F, M, out, ct);
    }
}

hipError_t bind_fault_clouds(unsigned* word) { return hc_bind_fault_word(word); }
// test hook (sbx_debug_raise_fault): one wave takes the path a failed miss loop takes
__global__ void k_raise_fault(unsigned code) { hc_fault(code); }
void launch_raise_fault(unsigned code, hipStream_t s) { hipLaunchKernelGGL(k_raise_fault, dim3(1), dim3(64), 0, s, code); }

}  // namespace sbx
