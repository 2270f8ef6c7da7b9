// shaderbox_amd/csrc/kern_egg.hip — APP_EGG: sphere-traced SDF scene ("Vectorpark egg").
//
// Follows /root/reference/src/app_egg.h: sdf :38-144, shadowmarch :161-186, render_scene
// :190-231, render (bars overlay) :233-251, with the SDF library of src/sdf.h and the IK solver of
// src/IK.h.  Everything in sdf() that does not depend on the sample point (the turntable and
// pedal rotations, both feet, both IK knees, the Bezier frames of the legs, the toe cylinders'
// axes) is a frame constant evaluated once on the host (FrameEgg); only the point-dependent part
// runs per march step.  `depth` (a _mutable global, :188) is a per-thread register that starts
// at -max_dist for every pixel = GLSL per-invocation semantics.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include "sbx_device.h"
#include "sbx_sdf.h"

#ifndef EGG_TW
#define EGG_TW 16          // wave tile EGG_TW x 64/EGG_TW pixels (profiles/r01_tile_shapes.txt)
#endif
#ifndef EGG_TX
#define EGG_TX 1           // waves per workgroup: 1 (4: the same single launch, 7 % slower with frames in flight at 1080p, 3 % at 4K)
#endif

namespace sbx {

// Is everything but the ground plane farther away than the ground?  Every other member of the union is
// >= .7 * (|p - oc| - orad) (FrameEgg, built in sbx_capi.hip), so with K = 1.43 (ground + 1e-3) + orad (1.43 > 1/.7),
// |p - oc| > K puts all of them strictly above the ground's distance: sdf() is the ground plane, exactly.
// The test runs on the WORLD point P against the centre carried to world space (FrameEgg.ocw): |P - ocw| is |p - oc| up to the 1e-6
// by which a rounded rotation matrix changes a length, far inside the 0.1 % by which 1.43 exceeds 1 / .7 — so a point that is far
// never pays for the turntable rotation (round 4: the culled call is ~28 instead of ~45 instructions; sky and ground waves are
// mostly such calls).  Any valid cull returns the same bits: it only ever claims what the full union would have returned.
__device__ __forceinline__ bool egg_far(const FrameEgg& F, v3 P, float ground_d) {
    const v3 q = P - F.ocw;
    const float K = (ground_d + 1e-3f) * 1.43f + F.orad;
    return ground_d >= 0.f && dot(q, q) > K * K;
}

// CULL = false (sbx_set_variant 1) evaluates every member everywhere: the reference form, kept for the parity sweeps
// W: the square roots' witness (sbx_sdf.h): Wit<true> takes the five-instruction roots and records arguments outside their domain
template <bool CULL, class W>
__device__ __forceinline__ D2 egg_sdf(const FrameEgg& F, v3 P, W& w) {
    const float mat_egg = 1.f, mat_bike = 2.f, mat_ground = 3.f;              // :17-20
    {
        const D2 ground = {dot(V3(0.f, 1.f, 0.f), P) + (1.2f + 0.5f), mat_ground};       // sd_plane :136-138
        if (CULL && egg_far(F, P, ground.d)) return ground;
    }
    const v3 p = mul(F.rot_y, P) - V3(0, 0.5f, 3.5f);                         // :40-41
    // Members are evaluated cheapest first with a running minimum `dmin`; a member whose lower bound exceeds it
    // cannot be the union's result and enters as +inf (op_add2 is a strict `<`, so the winner and its material
    // are unchanged).  Lower bounds (all with >= 1e-3 of slack over the rounding of the evaluation itself):
    //   wheel  : distance to the unit circle - .03             >= |pw| - 1.03
    //   foot   : max(axis, |u + 1/16| - 1/16) - .05            >= |P - M| / sqrt2 - 1/16 - .05   (M = midpoint)
    //   egg    : two smooth-mins (k = .5) of three spheres     >= |p - (0, .65, 0)| - .7 - .25
    //   leg    : bezier_far (sbx_sdf.h)
    const float inf = u2f(0x7f800000u);
    const D2 ground = {dot(V3(0.f, 1.f, 0.f), P) + (1.2f + 0.5f), mat_ground};           // sd_plane :136-138
    float dmin = ground.d;
    const bool pos_d = CULL && dmin >= 0.f;          // the bounds below assume a non-negative running minimum

    const v3 wheel_pos = V3(0, 1.2f, 0);
    const v3 pw = p + wheel_pos;
    D2 bike = {inf, mat_bike};
    {
        const float K = dmin * 1.001f + (1.03f + 2e-3f);
        if (!(pos_d && dot(pw, pw) > K * K))
            bike.d = w.length(V2(w.length(V2(pw.x, pw.y)) - 1.f, pw.z)) - .03f;              // sd_torus sdf.h:75-83
    }
    dmin = fmin_(dmin, bike.d);

    const float thick = .05f;
    D2 left_foot = {inf, mat_egg}, right_foot = {inf, mat_egg};
    {
        const float K = (dmin + (.0625f + .05f + 1e-3f)) * 1.4143f + 1e-3f;
        const v3 ql = p - F.foot_ml, qr = p - F.foot_mr;
        if (!(pos_d && dot(ql, ql) > K * K)) left_foot.d = sd_cylinder0<false>(F.foot_l, p + F.left_foot, thick, w);     // :120-123
        if (!(pos_d && dot(qr, qr) > K * K)) right_foot.d = sd_cylinder0<false>(F.foot_r, p + F.right_foot, thick, w);   // :125-128
    }
    const D2 feet = op_add2(left_foot, right_foot);
    dmin = fmin_(dmin, feet.d);

    const float egg_y = 0.65f;
    D2 egg = {inf, mat_egg};
    {
        const v3 qe = p - V3(0, egg_y, 0);
        const float K = dmin * 1.001f + (.95f + 3e-3f);
        if (!(pos_d && dot(qe, qe) > K * K)) {
            const float egg_m = w.length(p - V3(0, egg_y, 0)) - 0.475f;                  // :47-49
            const float egg_b = w.length(p - V3(0, egg_y - 0.45f, 0)) - 0.25f;
            const float egg_t = w.length(p - V3(0, egg_y + 0.45f, 0)) - 0.25f;
            const float egg_1 = op_blend(egg_m, egg_b, .5f);
            egg.d = op_blend(egg_1, egg_t, .5f);
        }
    }
    dmin = fmin_(dmin, egg.d);

    const D2 _1 = op_add2(feet, bike);
    const D2 _2 = op_add2(egg, _1);
    const float leg_l = (CULL && bezier_far(F.leg_l, p, thick, dmin)) ? inf : sd_bezier_x(F.leg_l, p, thick, w);     // :102-118
    const float leg_r = (CULL && bezier_far(F.leg_r, p, thick, dmin)) ? inf : sd_bezier_x(F.leg_r, p, thick, w);
    const D2 legs = op_add2(D2{leg_l, mat_egg}, D2{leg_r, mat_egg});
    const D2 _3 = op_add2(legs, _2);
    return op_add2(ground, _3);
}

template <bool CULL, class W>
__device__ __forceinline__ float egg_shadowmarch(const FrameEgg& F, v3 ro, v3 rd, W& w) {   // :161-186
    float t = 0.f, umbra = 1.f;
    for (int i = 0; i < 20; ++i) {
        const v3 p = ro + rd * t;
        const D2 d = egg_sdf<CULL>(F, p, w);
        if (t > 10.f) break;
        if (d.d < 0.001f) return 0.1f;
        t += d.d;
        umbra = fmin_(umbra, 15.f * d.d / t);
    }
    return umbra;
}

// HOT-FIRST DISPATCH.  Workgroups start in increasing (blockIdx.y, blockIdx.x).  The census of a 1920x1080 launch
// (tools/egg_census.py, profiles/r04_egg_census.txt) shows the chip full for the first 95 us and then 125 us of tail with fewer than
// 500 of 7168 wave slots in use: the ~350 waves on the SILHOUETTE of the egg and its legs, whose grazing rays run all 80 trace steps
// next to the surface (no member of the union can be culled there), take 100-175 us each and start around t = 50 us because the
// rows are dealt bottom to top.  With `hot` = the tiles under the projected bounding sphere of everything but the ground
// (launch_egg), the first hot.w * hot.h workgroups take those tiles and the others take the rest of the frame in row order: the
// long waves start at t = 0 and the cheap ones fill in behind them.  Which tile a workgroup renders changes nothing about a pixel.
#ifndef EGG_HOT_FIRST
#define EGG_HOT_FIRST 1
#endif
#ifndef EGG_PRIO_STEP
#define EGG_PRIO_STEP 0    // a wave still tracing after this many steps raises its issue priority (s_setprio): 0 = never.  Measured
                           // with steps 8 ... 45 and priorities 2 and 3: no difference at all (0.237-0.243 ms either way)
#endif
#ifndef EGG_PRIO
#define EGG_PRIO 2
#endif
#ifndef EGG_LDS_PAD
#define EGG_LDS_PAD 0      // bytes of dynamic LDS per (single-wave) workgroup, allocated only to CAP the waves per SIMD (see launch_egg)
#endif
struct HotRect { int x0, y0, w, h; };       // in workgroup tiles; w = 0: plain order
__device__ __forceinline__ void hot_first_tile(const HotRect& R, int gx, int& bx, int& by) {
    const int b = by * gx + bx, nr = R.w * R.h;
    if (b < nr) { by = R.y0 + b / R.w; bx = R.x0 + (b - (b / R.w) * R.w); return; }
    int c = b - nr;
    const int below = R.y0 * gx;
    if (c < below) { by = c / gx; bx = c - by * gx; return; }
    c -= below;
    const int side = gx - R.w, mid = R.h * side;
    if (c < mid) {
        const int q = c / side, k = c - q * side;
        by = R.y0 + q;
        bx = k < R.x0 ? k : k + R.w;
        return;
    }
    c -= mid;
    by = R.y0 + R.h + c / gx;
    bx = c - (c / gx) * gx;
}

#ifndef EGG_WITNESS
#define EGG_WITNESS 1      // five-instruction square roots with a recorded domain (sbx_sdf.h Wit): 0 = the IEEE roots only
#endif

// One pixel up to (colour, depth) — render_scene :190-231 — with the roots of witness `w`
template <bool CULL, class W>
__device__ __forceinline__ void egg_pixel(const FrameEgg& F, v2 pc, W& w, v3& color, float& depth, int& st_trace, int& st_shadow) {
    const v3 ro = F.cam.eye, rd = primary_dir(F.cam, pc, w);
    depth = -1e8f;                                          // :188, fresh per pixel
    color = V3(.1f, .1f, .7f);                              // background :9-12
    st_trace = 0; st_shadow = 0;
    float t = 0.f;
    // The trace only FINDS the hit; what the reference does inside the loop at the hit (`:205-228`: depth, the 20-step shadow
    // march of ground pixels, the flat colours, `break`) runs after the loop, once per wave with all of its hit lanes, instead of
    // once per distinct hit iteration of the wave with the few lanes that hit in that iteration.  Per lane the same operations
    // on the same values in the same order.  1920x1080: 0.54 -> 0.27 ms.  (Trace and shadow march as ONE loop around one copy of
    // the sdf — lanes with a ground hit start their shadow march while neighbours still trace —
    // is slower: 0.283 vs 0.273 ms, 4K 0.68 vs 0.64; the per-lane phase logic costs more than the shorter waves save.)
    bool hit = false;
    int mat = 0;
    v3 hp = V3(0, 0, 0);
    for (int i = 0; i < 80; ++i) {                          // render_scene :190-231
        if (EGG_PRIO_STEP > 0 && i == EGG_PRIO_STEP) __builtin_amdgcn_s_setprio(EGG_PRIO);   // a long wave: ahead of the short ones on its SIMD
        const v3 p = ro + rd * t;
        const D2 d = egg_sdf<CULL>(F, p, w);
        if (t > 15.f) break;
        if (d.d < 0.001f) { hit = true; mat = (int)d.m; hp = p; break; }
        t += d.d;
#ifdef SBX_EGG_STATS
        ++st_trace;
#endif
    }
#ifdef SBX_EGG_STATS
    if (hit && mat == 3) st_shadow = 1;
#endif
    if (hit) {
        if (mat == 1 || mat == 2) depth = fmax_(depth, hp.z);
        float s = 1.f;
        if (mat == 3) {
            const v3 sh_dir = V3(0, 1, 1);
            s = egg_shadowmarch<CULL>(F, hp + sh_dir * 0.05f, sh_dir, w);
        }
        v3 base = V3(1, 1, 1);                              // illuminate :29-35
        if (mat == 3) base = V3(13.f / 255.f, 104.f / 255.f, 0.f / 255.f);
        else if (mat == 1) base = V3(0.9f, 0.95f, 0.95f);
        else if (mat == 2) base = V3(.2f, .2f, .2f);
        color = base * s;
    }
}

// WIT: 0 = IEEE roots; 1 = witnessed roots (the shipped form); 2 = the same with the witness's lower edge at 1.0, so that waves
// near any primitive's axis DO record and re-run (sbx_set_variant 2: the test of the re-run path — same frame required)
template <bool CULL, int WIT>
__global__ void __launch_bounds__(64 * EGG_TX) k_egg(FrameEgg F, RowMap M, float* __restrict__ out, HotRect hot) {
#ifdef SBX_EGG_STATS
    const unsigned long long st_t0 = __builtin_amdgcn_s_memrealtime();      // census build (tools/egg_census.py): 100 MHz counter
#endif
    int st_trace = 0, st_shadow = 0;
#ifndef EGG_VCONST
#define EGG_VCONST 1       // the constants every sdf() call starts with — the turntable rotation and the cull sphere — in VGPRs: a VALU
#endif                     // instruction with an SGPR source issues at half rate on gfx950 (profiles/r02_ubench_issue.txt)
    if (EGG_VCONST) {
        asm volatile("" : "+v"(F.rot_y.c0.x), "+v"(F.rot_y.c0.y), "+v"(F.rot_y.c0.z), "+v"(F.rot_y.c1.x), "+v"(F.rot_y.c1.y),
                          "+v"(F.rot_y.c1.z), "+v"(F.rot_y.c2.x), "+v"(F.rot_y.c2.y), "+v"(F.rot_y.c2.z));
        asm volatile("" : "+v"(F.ocw.x), "+v"(F.ocw.y), "+v"(F.ocw.z), "+v"(F.orad));
#if EGG_VCONST > 1
        asm volatile("" : "+v"(F.foot_ml.x), "+v"(F.foot_ml.y), "+v"(F.foot_ml.z), "+v"(F.foot_mr.x), "+v"(F.foot_mr.y), "+v"(F.foot_mr.z));
#endif
    }
    int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    if (EGG_HOT_FIRST && hot.w > 0) hot_first_tile(hot, (int)gridDim.x, bx, by);          // wave-uniform
    const Pixel px = pixel_of<EGG_TW, EGG_TX>(M, (int)threadIdx.x, bx, by, (int)gridDim.y);
    if (!px.valid) return;
    const v2 pc = point_cam(F.cam, px.fx, px.fy);
    float depth;
    v3 color;
    if (WIT != 0) {
        Wit<true> w;
        if (WIT == 2) w.lo = 0x3F800000u;
        egg_pixel<CULL>(F, pc, w, color, depth, st_trace, st_shadow);
        if (__builtin_amdgcn_ballot_w64(w.bad) != 0ull) {      // some lane took a root outside the proved interval: the IEEE forms
            Wit<false> w0;
            egg_pixel<CULL>(F, pc, w0, color, depth, st_trace, st_shadow);
        }
    } else {
        Wit<false> w0;
        egg_pixel<CULL>(F, pc, w0, color, depth, st_trace, st_shadow);
    }
    // bars overlay :233-251
    const float bar_factor = 1.0f - smoothstep_(0.0f, 0.01f, abs_((abs_(pc.x) - 0.6f)) - 0.05f);
    const float depth_factor = 1.f - step_(1.f, depth);
    color = abs3(mix3(color, V3(.6f, .6f, .6f), bar_factor * depth_factor));
#ifdef SBX_EGG_STATS
    {   // lane 0 of the wave: start / end time, the wave's longest trace, lanes that ran a shadow march, the wave's place
        int mx = st_trace;
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
        const int nsh = __popcll(__builtin_amdgcn_ballot_w64(st_shadow != 0));
        const unsigned long long st_t1 = __builtin_amdgcn_s_memrealtime();
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        float4 o4;
        o4.x = __uint_as_float((unsigned)(st_t0 & 0xffffffffu));
        o4.y = __uint_as_float((unsigned)(st_t1 - st_t0));
        o4.z = __uint_as_float((unsigned)mx | ((unsigned)nsh << 8) | ((xcc & 0xfu) << 16) | ((hwid & 0xffffu) << 20));
        o4.w = __uint_as_float(hwid);
        reinterpret_cast<float4*>(out)[px.idx] = o4;
        return;
    }
#endif
    store_rgba(M, out, px.idx, to_srgb(color));
}

// The tiles under the projection of the sphere (F.oc, F.orad) around everything but the ground (sdf()'s p space: P = rot_y^T (p +
// (0, .5, 3.5))), for a launch that covers whole rows of the frame from row M.y0 (a contiguous strip; other maps: plain order).
// A hint about cost: off by any amount it only changes the order in which the same workgroups run.
static HotRect egg_hot_rect(const FrameEgg& F, const RowMap& M, dim3 grid) {
    HotRect none{0, 0, 0, 0};
    if (!EGG_HOT_FIRST || M.nranks != 1 || M.frag || M.span_mode || M.r0 != 0) return none;
    const v3 c = mul(transpose(F.rot_y), F.oc + V3(0, 0.5f, 3.5f));
    const v3 v = c - F.cam.eye;
    const float depth = dot(v, F.cam.fwd), r = F.orad;
    if (!(depth > r * 1.05f)) return none;                               // the camera is inside or beside the sphere
    // extent of x = X / Z over the sphere: the planes through the eye that contain the camera's up axis and touch the sphere — in
    // the (right, fwd) plane the sphere is a circle of radius r at (vx, depth), the tangents from the origin are at phi +- asin(r / d)
    const float vx = dot(v, F.cam.right), vy = dot(v, F.cam.up);
    auto extent = [&](float side, float& lo, float& hi) {
        const float d = sqrt_(side * side + depth * depth);
        const float phi = std::atan2(side, depth), al = std::asin(std::min(1.f, r / d));
        const float a = std::max(phi - al, -1.5f), b = std::min(phi + al, 1.5f);
        lo = std::tan(a); hi = std::tan(b);
    };
    float pxa, pxb, pya, pyb;
    extent(vx, pxa, pxb);
    extent(vy, pya, pyb);
    // point_cam = ((2 ndc - 1) * aspect * fov, (2 ndc - 1) * fov)  ->  pixel = ndc * res
    const float sx = F.cam.aspect_x * F.cam.fov, sy = F.cam.fov;
    auto pix = [](float pc, float scale, float res) { return (pc / scale + 1.f) * .5f * res; };
    const float xa = pix(pxa, sx, F.cam.res_x), xb = pix(pxb, sx, F.cam.res_x);
    const float ya = pix(pya, sy, F.cam.res_y) - (float)M.y0, yb = pix(pyb, sy, F.cam.res_y) - (float)M.y0;
    if (!(xa == xa && xb == xb && ya == ya && yb == yb)) return none;
    constexpr int TWP = EGG_TW * EGG_TX, THP = 64 / EGG_TW;
    const int gx = (int)grid.x, gy = (int)grid.y;
    const int x0 = std::max(0, std::min(gx, (int)std::floor(xa / TWP))), x1 = std::max(0, std::min(gx, (int)std::ceil(xb / TWP)));
    const int y0 = std::max(0, std::min(gy, (int)std::floor(ya / THP))), y1 = std::max(0, std::min(gy, (int)std::ceil(yb / THP)));
    if (x1 <= x0 || y1 <= y0) return none;
    return HotRect{x0, y0, x1 - x0, y1 - y0};
}

void launch_egg(const FrameEgg& F, const RowMap& M, float* out, hipStream_t s, int variant) {
    const dim3 grid = grid_for<EGG_TW, EGG_TX>(M);
    const HotRect hot = egg_hot_rect(F, M, grid);
    static const int pad = []() { const char* e = std::getenv("SBX_DEBUG_LDS_PAD"); return e ? std::atoi(e) : EGG_LDS_PAD; }();
    if (variant == 1) hipLaunchKernelGGL((k_egg<false, 0>), grid, dim3(64 * EGG_TX), (size_t)pad, s, F, M, out, hot);
    else if (variant == 2) hipLaunchKernelGGL((k_egg<true, 2>), grid, dim3(64 * EGG_TX), (size_t)pad, s, F, M, out, hot);
    else if (variant == 3) hipLaunchKernelGGL((k_egg<true, 0>), grid, dim3(64 * EGG_TX), (size_t)pad, s, F, M, out, hot);
    else hipLaunchKernelGGL((k_egg<true, EGG_WITNESS>), grid, dim3(64 * EGG_TX), (size_t)pad, s, F, M, out, hot);
}

}  // namespace sbx
