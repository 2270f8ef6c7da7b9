// shaderbox_amd/csrc/kern_sdf_ao.hip — APP_SDF_AO: skate-ramp SDF with distance-field AO and fog.
//
// Follows /root/reference/src/app_sdf_ao.h: sdf_pipe :54-113, sdf :115-150, sdf_normal :152-163,
// sdf_ao :165-181, illuminate :211-243, render_impl :245-285 (shadow march compiled out, :269-274),
// render (height fog) :287-311; SDF primitives from src/sdf.h.  The two constant-angle rotations
// (rotate_around_x(-90), rotate_around_y(180)) and normalize(1,2,1) are frame constants.
#include "sbx_device.h"
#include "sbx_sdf.h"

#ifndef AO_MIN_WAVES
#define AO_MIN_WAVES 4
#endif

namespace sbx {

// HW (the CULL kernels, which run for tame frames only: finite u_time, u_res checked by the C API, hence finite points): the SDF's min /
// max as v_min_f32 / v_max_f32 (sbx_sdf.h hmin_ / hmax_).  Besides the primitives' own, the unions below are safe: every operand is a
// primitive's value (never -0) except `-c` in op_sub, which is the SECOND operand of a max whose first, a box distance, is never -0.
#ifndef AO_HW_MINMAX
#define AO_HW_MINMAX 1
#endif
template <bool HW, class W>      // W: the square roots' witness (sbx_sdf.h Wit)
__device__ __forceinline__ D2 ao_sdf_pipe(const FrameSdfAo& F, v3 pos, W& w) {                          // :54-113
    const v3 size = V3(1.3f, 1.f, 1.25f);                                                          // :52
    v3 p = pos - V3(0, size.y, 0);
    const float b = sd_box<HW>(p, size);
    p = p - V3(.7f, .5f, 0);
    p = mul(p, F.rx_m90);
    const float c = sd_y_cylinder<HW>(p, size.y + .55f, 2.f * size.z + .1f, w);
    const D2 pipe = {hmax_neg_<HW>(b, c), 2.f};                                                           // op_sub, mat_pipe

    p = pos - V3(0, size.y, 0);
    p = p - V3(-size.x + .525f, size.y, 0);
    p = mul(p, F.rx_m90);
    const D2 coping = {sd_y_cylinder<HW>(p, .025f, 2.f * size.z, w), 5.f};                                // mat_coping

    p = pos - V3(0, size.y * 2.f, 0);
    const float rail = sd_box<HW>(p + V3(size.x, -.25f, 0), V3(.025f, .05f, size.z));
    const v3 B = V3(.025f, .125f, .025f);
    const float H = -.125f;
    const float bar_1 = sd_box<HW>(p + V3(size.x, H, 0), B);
    const float bar_2 = sd_box<HW>(p + V3(size.x, H, size.z / 2.f), B);
    const float bar_3 = sd_box<HW>(p + V3(size.x, H, size.z), B);
    const float bar_4 = sd_box<HW>(p + V3(size.x, H, -size.z / 2.f), B);
    const float bar_5 = sd_box<HW>(p + V3(size.x, H, -size.z), B);
    const float b_a = hmin_<HW>(bar_1, bar_2);
    const float b_b = hmin_<HW>(b_a, bar_3);
    const float b_c = hmin_<HW>(bar_4, bar_5);
    const float bars = hmin_<HW>(b_b, b_c);
    const D2 railing = {hmin_<HW>(rail, bars), 4.f};                                                   // mat_deck
    const D2 deck = op_add2(railing, coping);
    return op_add2(pipe, deck);
}

// The reference's sd_box is the max-norm distance max_i(|p_i| - b_i) (sdf.h:67-73) and sd_y_cylinder is
// max(length(p.xz) - r, |p.y| - h/2) >= the max-norm distance to the cylinder's own bounding box; `max(box, -cylinder)`
// >= the box.  Every member of sdf_pipe lies inside the axis-aligned box x [-1.325, 1.3], y [0, 2.3], z [-1.275, 1.275]
// of the pipe's own coordinates (ramp block, coping, rail, bars; :54-113), and for nested intervals
// |p_i - c_member| - b_member >= |p_i - C| - H on every axis, so sdf_pipe(s) >= the max-norm distance from s to that box
// (taken .01 larger).  Where this exceeds what ground, reference pole and bottom slab already give, the pipe cannot be
// the minimum of the union and enters it as +inf — same result as evaluating it (op_add2 is a strict `<`).  The two
// ramps' boxes are .65 apart, so near any surface at least one ramp (10 primitives) is skipped.
__device__ __forceinline__ bool ao_pipe_far(v3 s, float dmin) {
    const v3 c = V3(-.0125f, 1.15f, 0.f), h = V3(1.3225f + .01f, 1.15f + .01f, 1.275f + .01f);
    const float lb = fmax_(abs_(s.x - c.x) - h.x, fmax_(abs_(s.y - c.y) - h.y, abs_(s.z - c.z) - h.z));
    return dmin >= 0.f && lb > dmin * 1.001f + 2e-3f;
}

template <bool CULL, class W>   // CULL false (sbx_set_variant 1): both ramps evaluated everywhere, the reference form
__device__ __forceinline__ D2 ao_sdf(const FrameSdfAo& F, v3 pos, W& w) {                                // :115-150
    constexpr bool HW = CULL && AO_HW_MINMAX;
    const v3 size = V3(1.3f, 1.f, 1.25f);
    const float B = .15f;
    v3 p = pos - V3(0, B, 0);
    const D2 bottom = {sd_box<HW>(p, V3(2.25f * size.x, B, size.z)), 3.f};                             // mat_bottom
    const D2 ref = {sd_box<HW>(pos, V3(.025f, 15, .025f)), 0.f};                                       // mat_debug
    const D2 ground = {dot(V3(0, 1, 0), pos) + 0.f, 1.f};                                          // sd_plane, mat_ground
    const float dmin = hmin_<HW>(hmin_<HW>(ground.d, ref.d), bottom.d);       // (ground.d ends in `+ 0.f`: never -0)
    const float inf = u2f(0x7f800000u);
    const v3 s1 = p + V3(1.25f * size.x, 0, 0);
    D2 pipe1 = {inf, 2.f};
    if (!(CULL && ao_pipe_far(s1, dmin))) pipe1 = ao_sdf_pipe<HW>(F, s1, w);
    p = p - V3(1.25f * size.x, 0, 0);
    p = mul(p, F.ry_180);
    D2 pipe2 = {inf, 2.f};
    if (!(CULL && ao_pipe_far(p, dmin))) pipe2 = ao_sdf_pipe<HW>(F, p, w);
    const D2 pipe = op_add2(pipe1, pipe2);
    const D2 g = op_add2(ground, ref);
    const D2 b = op_add2(pipe, bottom);
    return op_add2(b, g);
}

#ifndef AO_WIT_DIR
#define AO_WIT_DIR 0       // the primary direction through the witness's normalize too: 81 VGPRs, 5 instead of 6 waves
#endif
#ifndef AO_WITNESS
#define AO_WITNESS 1       // witnessed five-instruction square roots (sbx_sdf.h Wit), as in k_egg: 0 = the IEEE roots only
#endif

// One pixel up to (rgb, t) — render_impl :245-285 — with the roots of witness `w`
template <bool CULL, class W>
__device__ __forceinline__ void ao_pixel(const FrameSdfAo& F, v3 ro, v2 pc, W& w, v3& rgb, float& t, v3& rd) {
    rd = AO_WIT_DIR ? primary_dir(F.cam, pc, w) : primary_dir(F.cam, pc);
    rgb = V3(.1f, .1f, .7f);                                      // background :9-12
    t = 0.f;
    // The trace only FINDS the hit; the reference's hit block (`:258-281`: 6-tap normal, 5-tap AO, lights, material, `break`)
    // runs after the loop, once per wave with all of its hit lanes instead of once per distinct hit iteration of the wave.
    // Per lane the same operations on the same values in the same order.
    bool hit = false;
    int mat = 0;
    v3 p = V3(0, 0, 0);
    for (int i = 0; i < 70; ++i) {                                // render_impl :245-285
        const v3 pi = ro + rd * t;
        const D2 d = ao_sdf<CULL>(F, pi, w);
        if (t > 20.f) break;
        if (d.d < .005f) { hit = true; mat = (int)d.m; p = pi; break; }
        t += d.d;
    }
    {
        if (hit) {
            // sdf_normal :152-163
            const float e = 0.001f;
            const v3 n = normalize(V3(          // (the IEEE form: a flat surface's normal has exact zero components, which the witness records)
                ao_sdf<CULL>(F, p + V3(e, 0, 0), w).d - ao_sdf<CULL>(F, p - V3(e, 0, 0), w).d,
                ao_sdf<CULL>(F, p + V3(0, e, 0), w).d - ao_sdf<CULL>(F, p - V3(0, e, 0), w).d,
                ao_sdf<CULL>(F, p + V3(0, 0, e), w).d - ao_sdf<CULL>(F, p - V3(0, 0, e), w).d));
            // sdf_ao :165-181
            float occlusion = 0.f, inv2k = 1.f;
            for (float k = 1.f; k <= 5.f; k += 1.f) {
                const v3 q = p + .5f * k * n;
                const float dd = ao_sdf<CULL>(F, q, w).d;
                // pow(2, k) of the math spec is exactly 2^k for k = 1..5 (log2(2) = 1 and 2^integer are exact in its
                // binary64 sequence), so 1 / pow(2, k) is the exact power of two below
                inv2k *= .5f;
                occlusion += inv2k * (.5f * k - dd);
            }
            const float ao = 1.f - clamp_(occlusion, 0.f, 1.f);
            const float sh = 1.f;
            // illuminate :211-243
            v3 accum = V3(0, 0, 0);
            const float sun_ray = fmax_(0.f, dot(F.sun_dir, n));
            accum = accum + sh * sun_ray * V3(1.2f, 1.3f, 1.f);
            accum = accum + ao * n.y * V3(.15f, .15f, .4f);
            const float ind = fmax_(0.f, dot(F.sun_dir * V3(-1, 0, -1), n));
            accum = accum + ao * ind * V3(.4f, .28f, .2f);
            v3 mat_c = V3(0, 0, 0);                               // get_material :23-33 / setup_scene :35-43
            if (mat == 0) mat_c = V3(1, 1, 1);
            else if (mat == 1) mat_c = V3(0, .2f, 0);
            else if (mat == 2 || mat == 3 || mat == 4) mat_c = V3(.1f, .1f, .1f);
            else if (mat == 5) mat_c = V3(.4f, .4f, .4f);
            if (mat == 1) {
                const float cb = mod_(floor_(p.x * .5f) + floor_(p.z * .5f), 2.0f);   // checkboard_pattern util.h:95-101
                mat_c = mix3(mat_c - .15f * mat_c, mat_c + .15f * mat_c, cb);
            }
            rgb = accum * mat_c;
        }
    }
}

template <bool CULL, int WIT>      // WIT: 0 IEEE roots, 1 witnessed roots, 2 the witness's test edge (sbx_set_variant 2), as k_egg
__global__ void __launch_bounds__(WG_THREADS, AO_MIN_WAVES) k_sdf_ao(FrameSdfAo F, RowMap M, float* __restrict__ out) {
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();      // (the dispatch order's cost table, RowMap.cost)
    const Pixel px = pixel_of_thread(M);
    if (!px.valid) return;
    const v2 pc = point_cam(F.cam, px.fx, px.fy);
    const v3 ro = F.cam.eye;
    v3 rgb, rd;
    float t;
    if (WIT != 0) {
        Wit<true> w;
        if (WIT == 2) w.lo = 0x3F800000u;
        ao_pixel<CULL>(F, ro, pc, w, rgb, t, rd);
        if (__builtin_amdgcn_ballot_w64(w.bad) != 0ull) {
            Wit<false> w0;
            ao_pixel<CULL>(F, ro, pc, w0, rgb, t, rd);
        }
    } else {
        Wit<false> w0;
        ao_pixel<CULL>(F, ro, pc, w0, rgb, t, rd);
    }
    // fog :287-311 (t is the march length at exit)
    const float fog_factor = F.fog_density * exp_(-ro.y * F.fog_falloff)
                           * (1.f - exp_(-t * rd.y * F.fog_falloff))
                           / (rd.y * F.fog_falloff);
    const v3 col = abs3(mix3(rgb, V3(1, 1, 1), fog_factor));
    tile_cost_store(M, tl_t0);
    store_rgba(M, out, px.idx, to_srgb(col));
}

dim3 sdf_ao_grid(const RowMap& M) { return grid_for(M); }

void launch_sdf_ao(const FrameSdfAo& F, const RowMap& M, float* out, hipStream_t s, int variant) {
    if (variant == 1) hipLaunchKernelGGL((k_sdf_ao<false, 0>), grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
    else if (variant == 2) hipLaunchKernelGGL((k_sdf_ao<true, 2>), grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
    else if (variant == 3) hipLaunchKernelGGL((k_sdf_ao<true, 0>), grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
    else hipLaunchKernelGGL((k_sdf_ao<true, AO_WITNESS>), grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
}

}  // namespace sbx
