// shaderbox_amd/csrc/kern_vinyl.hip — APP_VINYL: turntable SDF with anisotropic (Ward-like) vinyl shading.
//
// Follows /root/reference/src/app_vinyl.h with the C++ build's 60 march steps (:411-416; GLSL uses 180),
// SHADERTOY undefined: sdf_logo :68-85, sdf_platter :87-125, sdf_tonearm :127-255, sdf :257-265, sdf_normal
// :267-278, illuminate :293-377, sdf_shadow :379-404, render :406-457.  Everything that depends only on u_time
// (platter and wobble rotations, the tonearm's constant-angle rotations, capsule/Bezier/cylinder frames) is in
// FrameVinyl.  SURVEY.md §8f row 4; the reference has no known answers for this app (parity vs oracle only).
#include "sbx_device.h"
#include "sbx_ldsframe.h"
#include "sbx_sdf.h"
#include "sbx_noise.h"

#ifndef VI_MIN_WAVES
#define VI_MIN_WAVES 5     // 3840x2160: 3 waves (134 VGPRs) 4.03 ms, 4 waves 3.85, 5 waves (96 VGPRs, spills outside the march) 3.71
#endif

#ifndef VI_LDS_FRAME
#define VI_LDS_FRAME 1
#endif

namespace sbx {

// FV(field): a member of the frame block where it is used (sbx_ldsframe.h: the block lives in LDS, the read is volatile).
#if VI_LDS_FRAME
#define FV(x) lds_ld(F.x)
#else
#define FV(x) (F.x)
#endif

// HW (the CULL kernel: tame frames, fixed camera, hence finite points): the unions' min / max as v_min_f32 / v_max_f32 (sbx_sdf.h
// hmin_ / hmax_).  They differ from the spec's compare-and-select only when the FIRST operand is a NaN — the one member that can be a
// NaN, the Bezier link (0 / 0 at isolated points), enters its min as the SECOND operand — or for zeros of opposite sign in the
// order (+0, -0) for min, (-0, +0) for max: primitive values are never -0 (a length, a difference of equal numbers, |p| - b), a -0
// arises only from `max(x, -y)` with y == +0, and the places where such a value meets a +0 are first operands of a min (equal
// either way) or dmin2, which is only compared with 0 and added to a margin.
#ifndef VIN_HW_MINMAX
#define VIN_HW_MINMAX 1
#endif
template <bool HW>
__device__ __forceinline__ float vinyl_logo(const FrameVinyl& F, v3 pos, float thick) {       // :68-85
    const v3 b = V3(.25f, thick, 1.2f), d = V3(.7f, 0, 0);
    v3 p = mul(pos, FV(ry30));
    const float v1 = sd_box<HW>(p - d, b);
    p = mul(pos, FV(rym30));
    const float v2 = sd_box<HW>(p + d, b);
    const float x = sd_box<HW>(pos, V3(1.5f, thick, 1.35f));
    return hmax_<HW>(hmin_<HW>(v1, v2), x);                              // op_intersect(op_add(v1, v2), x)
}
template <bool HW, class W>      // W: the square roots' witness (sbx_sdf.h Wit)
__device__ __forceinline__ D2 vinyl_platter(const FrameVinyl& F, v3 p, W& w) {                       // :87-125
    const float thick = .1f;
    const D2 lead_in = {sd_y_cylinder<HW>(p, 6.f, thick - .05f, w), 2.f};
    const D2 groove = {sd_y_cylinder<HW>(p, 5.9f, thick, w), 1.f};
    const D2 dead_wax = {sd_y_cylinder<HW>(p, 3.f, thick, w), 2.f};
    const D2 label = {sd_y_cylinder<HW>(p, 2.f, thick, w), 3.f};
    const D2 logo = {vinyl_logo<HW>(F, p, thick - .0175f), 4.f};
    const float spc = sd_y_cylinder<HW>(p, .10f, .6f, w);
    const float sps = w.length(p - V3(0, .3f, 0)) - .10f;
    const D2 spindle = {hmin_<HW>(spc, sps), 5.f};
    const D2 d0 = op_add2(groove, lead_in);
    const D2 d1 = op_add2(d0, dead_wax);
    const D2 d2 = op_add2(label, logo);
    const D2 d3 = op_add2(d1, d2);
    const D2 d4 = op_add2(d3, spindle);
    const float defect1 = w.length(p + V3(6.05f, 0, 0)) - .1f;
    const float defect2 = w.length(p + V3(-6.05f, 0, 0)) - .1f;
    const float defect = hmin_<HW>(defect1, defect2);
    return D2{hmax_neg_<HW>(d4.d, defect), d4.m};                       // op_sub
}
// Exact culling (as in kern_egg.hip): `dmin` is the distance the platter already gives at pos; a group of the tonearm
// whose lower bound exceeds the running minimum cannot be returned by the union and enters it as +inf.
//  * base: three y-cylinders around base_p (max(.., -platter) >= its first argument); a y-cylinder is
//    max(length(xz) - r, |y| - h/2) >= the max-norm distance to its own bounding box, all inside x,z in base_p +- 3,
//    |y| <= 1.25;
//  * the Bezier link: bezier_far (sbx_sdf.h);
//  * headshell + cartridge: collar, finger lift and cartridge boxes all lie within 1.3 of a3 in the wobbled frame
//    (offsets and half-sizes of :163-232 added up); a box evaluated in a rotated frame is a max-norm distance
//    >= Euclidean / sqrt3, the collar's max(axis, slabs) form >= Euclidean / sqrt2 minus its size, and max(x, -cut) >= x:
//    every member is >= .577 (|p - a3| - 1.3) - so with K = 1.74 (dmin + 1e-3) + 1.35, |p - a3| > K puts them all above dmin.
template <bool CULL, class W>   // CULL false (sbx_set_variant 1): no culling, the reference form
__device__ __forceinline__ D2 vinyl_tonearm(const FrameVinyl& F, v3 pos, float dmin, W& w) {        // :127-255
    constexpr bool HW = CULL && VIN_HW_MINMAX;
    const float inf = u2f(0x7f800000u);
    const v3 base_p = V3(-7, 0, -5);
    D2 base = {inf, 5.f};
    {
        const v3 q = pos - base_p;
        const float lb = hmax_<HW>(abs_(q.x) - 3.01f, hmax_<HW>(abs_(q.y) - 1.26f, abs_(q.z) - 3.01f));
        if (!(CULL && dmin >= 0.f && lb > dmin * 1.001f + 2e-3f)) {
            const float platter = sd_y_cylinder<HW>(pos, 6.25f, 1.f, w);
            const float base_0 = sd_y_cylinder<HW>(pos - base_p, 3.f, .25f, w);
            const float base_1 = hmax_neg_<HW>(base_0, platter);
            const float base_2 = sd_y_cylinder<HW>(pos - base_p, 1.25f, 1.f, w);
            const float base_12 = hmin_<HW>(base_1, base_2);
            const D2 base_a = {base_12, 5.f};
            const D2 base_b = {sd_y_cylinder<HW>(pos - base_p, 0.5f, 2.5f, w), 5.f};
            base = op_add2(base_a, base_b);
        }
    }

    const v3 p = mul(pos, FV(wobble));
    const float R = .1f;
    const float arm1 = sd_capsule_f(p, FV(arm1.a), FV(arm1.ab), FV(arm1.rd), R, w);
    const float arm2 = sd_capsule_f(p, FV(arm2.a), FV(arm2.ab), FV(arm2.rd), R, w);
    const float arm3 = sd_capsule_f(p, FV(arm3.a), FV(arm3.ab), FV(arm3.rd), R, w);
    const float arm_link1 = hmin_<HW>(arm1, arm2);
    const float arm_link2 = hmin_<HW>(arm_link1, arm3);
    const float dmin2 = hmin_<HW>(hmin_<HW>(dmin, base.d), arm_link2);
    const BezierFrame abz = FV(armb);
    const float armb = (CULL && bezier_far(abz, p, R, dmin2)) ? inf : sd_bezier_x(abz, p, R, w);
    const D2 arm = {hmin_<HW>(arm_link2, armb), 5.f};
    {
        const v3 q = p - FV(a3);
        const float K = (dmin2 + 1e-3f) * 1.74f + 1.35f;
        if (CULL && dmin2 >= 0.f && dot(q, q) > K * K) {
            const D2 tone1 = op_add2(base, arm);
            return op_add2(tone1, D2{inf, 5.f});
        }
    }

    const v3 clr_p = p - FV(a3);
    const float clr_r = R * 1.5f;
    const float collar = sd_cylinder0<HW>(FV(collar), clr_p, clr_r, w);
    const float fl_w = .045f, fl_h = .020f;
    const float fl_len1 = clr_r * 1.f;
    const float fl_len2 = fl_len1 * 1.2f;
    const v3 fl_p = mul(clr_p - FV(fl_sub1) - FV(fl_sub2), FV(fl_rot));
    const float fl1 = sd_box<HW>(fl_p, V3(fl_w, fl_h, fl_len1));
    const float fl2 = sd_box<HW>(mul(fl_p - V3(0, 0, fl_len1), FV(fl_rot2)) - V3(0, 0, fl_len2), V3(fl_w, fl_h, fl_len2));
    const float finger_lift = hmin_<HW>(fl1, fl2);
    const D2 headshell = {hmin_<HW>(collar, finger_lift), 5.f};

    const float ctg_w = .05f, ctg_h = .05f, ctg_len1 = .3f, ctg_len2 = .5f;
    const v3 ctg_p = mul(clr_p, FV(arm_xform));
    const float ctg1 = sd_box<HW>(ctg_p, V3(ctg_len1, ctg_h, ctg_w));
    const v3 ctg2_p = mul(ctg_p - V3(ctg_len1, 0, 0), FV(ctg_rot)) - V3(ctg_len2 - 0.03f, -.01f, 0);
    const float ctg2 = sd_box<HW>(ctg2_p, V3(ctg_len2, ctg_h, ctg_w));
    const float cut = sd_box<HW>(mul(mul(ctg2_p, FV(cut_rx10)) - V3(0, .05f, .175f), FV(cut_rym5)),
                             V3(ctg_len2 * 2.f, ctg_h * 3.f, ctg_w * 3.2f));
    const float cut2 = sd_box<HW>(mul(ctg2_p - V3(.3f, .2f, 0), FV(cut2_rz10)), V3(.4f, .2f, .3f));
    const float ctg12 = hmin_<HW>(ctg1, ctg2);
    const float ctg12c = hmax_neg_<HW>(ctg12, cut);
    const D2 cartridge = {hmax_neg_<HW>(ctg12c, cut2), 5.f};

    const D2 tone1 = op_add2(base, arm);
    const D2 tone2 = op_add2(headshell, cartridge);
    return op_add2(tone1, tone2);
}
template <bool CULL, class W>
__device__ __forceinline__ D2 vinyl_sdf(const FrameVinyl& F, v3 pos, W& w) {                   // :257-265
    const D2 plat = vinyl_platter<(CULL && VIN_HW_MINMAX)>(F, mul(pos, FV(platter_rot)), w);
    const D2 arm = vinyl_tonearm<CULL>(F, pos, plat.d, w);
    return op_add2(plat, arm);
}
__device__ __forceinline__ float saw(float x) { return x - floor_(x); }                        // :280-283
__device__ __forceinline__ float pulse(float x) { return saw(x + .5f) - saw(x); }              // :285-288

__device__ __forceinline__ v3 vinyl_base_color(int mat) {                                       // setup_scene :40-54
    switch (mat) {
    case 0: return V3(1, 1, 1);
    case 1: return V3(.01f, .01f, .01f);
    case 2: return V3(.05f, .05f, .05f);
    case 3: return V3(.5f, .5f, .0f);
    case 4: return V3(0, 0, .7f);
    case 5: return V3(.7f, .7f, .7f);
    default: return V3(0, 0, 0);                                 // unset material slots are zero (App. B5)
    }
}

#ifndef VI_WITNESS
#define VI_WITNESS 1       // witnessed five-instruction square roots in sdf() (sbx_sdf.h Wit), as in k_egg: 0 = the IEEE roots only
#endif

// One pixel up to its colour — render :406-457 — with the sdf's roots of witness `w`.  F: the kernel argument (camera, sun, steps:
// SGPRs); Fs: the block sdf() reads (the LDS copy, or F itself).
template <bool CULL, class W>
__device__ __forceinline__ void vinyl_pixel(const FrameVinyl& F, const FrameVinyl& Fs, v2 pc, W& w, v3& color) {
    const v3 ro = F.cam.eye, rd = primary_dir(F.cam, pc, w);
    color = V3(1, 1, 1);                                        // background :15-18
    float t = 0.f;
    // The trace only FINDS the hit; the reference's hit block (`:436-452`: 20-step soft shadow, anisotropic or 6-tap-normal
    // shading, `break`) runs after the loop, once per wave with all of its hit lanes instead of once per distinct hit iteration
    // of the wave.  Per lane the same operations on the same values in the same order.
    bool hit = false;
    int mat = 0;
    v3 p = V3(0, 0, 0);
    for (int i = 0; i < F.steps; ++i) {                           // render :427-455; 60 (C++ build) or 180 steps :411-416
        const v3 pi = ro + rd * t;
        const D2 d = vinyl_sdf<CULL>(Fs, pi, w);
        if (t > 40.f) break;
        if (d.d < .005f) { hit = true; mat = (int)d.m; p = pi; break; }
        t += d.d;
    }
    {
        if (hit) {
            // sdf_shadow :379-404
            float sh = 1.f;
            {
                const v3 so = p + F.sun_dir * 0.05f;
                float ts = 0.f;
                for (int k = 0; k < 20; ++k) {
                                const D2 ds = vinyl_sdf<CULL>(Fs, so + F.sun_dir * ts, w);
                    if (ts > 5.f) break;
                    if (ds.d < .005f) { sh = .05f; break; }
                    ts += ds.d;
                    sh = fmin_(sh, 16.f * ds.d / ts);
                }
            }
            // illuminate :293-377
            v3 L = F.sun_dir;
            v3 V = w.normalize(ro - p);
            const v3 base = vinyl_base_color(mat);
            v3 lit;
            if (mat == 1 || mat == 2) {
                const v3 ho = mul(p, F.platter_rot);
                L = mul(L, F.platter_rot);
                V = mul(V, F.platter_rot);
                const float r = length(ho);
                const v3 B = ho / r;
                v3 N = V3(0, 1, 0);
                if (mat == 1) {
                    const float rr = r + .07575f * noise_iq(ho * 2.456f);
                    const float s = pulse(rr * 24.f);
                    if (s > 0.f) {
                        N = normalize(N + B);
                        N = N - 2.f * dot(V3(0, 1, 0), N) * V3(0, 1, 0);      // reflect(N, (0,1,0))
                    }
                }
                if (mat == 2) {
                    const float s = saw(r * 4.f);
                    N = normalize(N + B * ((s > .9f) ? 1.f : 0.f));
                }
                const v3 T = cross(B, N);
                const float ro_diff = 1.f, ro_spec = .0725f, a_x = .025f, a_y = .5f;
                const v3 H = w.normalize(V + L);
                const float dotLN = dot(L, N);
                const v3 diffuse = base * (ro_diff / 3.14159265359f) * fmax_(0.f, dotLN);
                const float spec_a = ro_spec / sqrt_(dotLN * dot(V, N));
                const float spec_b = 1.f / (4.f * 3.14159265359f * a_x * a_y);
                const float ht = dot(H, T) / a_x;
                const float hb = dot(H, B) / a_y;
                const float spec_c = -2.f * (ht * ht + hb * hb) / (1.f + dot(H, N));
                const v3 specular = V3(1, 1, 1) * spec_a * spec_b * exp_(spec_c);
                lit = diffuse + specular;
            } else {
                const float e = 0.001f;                              // sdf_normal :267-278
                const v3 n = normalize(V3(      // (the IEEE form: flat surfaces give exact zero components, which the witness records)
                    vinyl_sdf<CULL>(Fs, p + V3(e, 0, 0), w).d - vinyl_sdf<CULL>(Fs, p - V3(e, 0, 0), w).d,
                    vinyl_sdf<CULL>(Fs, p + V3(0, e, 0), w).d - vinyl_sdf<CULL>(Fs, p - V3(0, e, 0), w).d,
                    vinyl_sdf<CULL>(Fs, p + V3(0, 0, e), w).d - vinyl_sdf<CULL>(Fs, p - V3(0, 0, e), w).d));
                const v3 diffuse = base * fmax_(0.f, dot(L, n));
                const v3 H = w.normalize(V + L);
                const v3 specular = pow_(fmax_(0.f, dot(H, n)), 50.f) * V3(1, 1, 1);
                lit = diffuse + specular;
            }
            color = lit * sh;
        }
    }
}

template <bool CULL, int WIT>      // WIT: 0 IEEE roots, 1 witnessed roots, 2 the witness's test edge (sbx_set_variant 2), as k_egg
__global__ void __launch_bounds__(WG_THREADS, VI_MIN_WAVES) k_vinyl(FrameVinyl F, RowMap M, float* __restrict__ out) {
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();      // (the dispatch order's cost table, RowMap.cost)
#if VI_LDS_FRAME
    // the frame block (~220 floats of rotations and primitive frames) in LDS: sbx_ldsframe.h
    __shared__ FrameVinyl Fs;
    lds_frame_fill<FrameVinyl, WG_THREADS>(Fs);
#define VI_FS Fs
#else
#define VI_FS F
#endif
    const Pixel px = pixel_of_thread(M);
    if (!px.valid) return;
    const v2 pc = point_cam(F.cam, px.fx, px.fy);
    v3 color;
    if (WIT != 0) {
        Wit<true> w;
        if (WIT == 2) w.lo = 0x3F800000u;
        vinyl_pixel<CULL>(F, VI_FS, pc, w, color);
        if (__builtin_amdgcn_ballot_w64(w.bad) != 0ull) {
            Wit<false> w0;
            vinyl_pixel<CULL>(F, VI_FS, pc, w0, color);
        }
    } else {
        Wit<false> w0;
        vinyl_pixel<CULL>(F, VI_FS, pc, w0, color);
    }
    tile_cost_store(M, tl_t0);
    store_rgba(M, out, px.idx, to_srgb(color));
}

dim3 vinyl_grid(const RowMap& M) { return grid_for(M); }

void launch_vinyl(const FrameVinyl& F, const RowMap& M, float* out, hipStream_t s, int variant) {
    if (variant == 1) hipLaunchKernelGGL((k_vinyl<false, 0>), grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
    else if (variant == 2) hipLaunchKernelGGL((k_vinyl<true, 2>), grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
    else if (variant == 3) hipLaunchKernelGGL((k_vinyl<true, 0>), grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
    else hipLaunchKernelGGL((k_vinyl<true, VI_WITNESS>), grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
}

}  // namespace sbx
