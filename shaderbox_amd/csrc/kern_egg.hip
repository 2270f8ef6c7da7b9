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
#ifndef EGG_COOP
#define EGG_COOP 1         // the four waves of a workgroup finish its LONG rays together, one part of sdf() each (see egg_coop_finish)
#endif
#ifndef EGG_TX
#define EGG_TX (EGG_COOP ? 4 : 1)   // waves per workgroup.  Without the cooperative finish: 1 (4: the same single launch, 7 % slower with
#endif                     // frames in flight at 1080p, 3 % at 4K)
#ifndef EGG_COOP_K0
#define EGG_COOP_K0 8      // trace steps before a workgroup first counts its rays still marching
#endif
#ifndef EGG_COOP_DK
#define EGG_COOP_DK 4      // ... and between later counts
#endif
#ifndef EGG_NUM_VGPR
#define EGG_NUM_VGPR 72    // 7 waves per SIMD: the kernel needs all 7 to fill the pipes (5: -8 % throughput).  As amdgpu_num_vgpr, not as
#endif                     // __launch_bounds__' second argument: that one also takes 12 SGPRs away (94 instead of 106), and this kernel
                           // keeps ~100 frame constants in SGPRs — with 94 the trace loop reloads 136 of them per step through v_readlane
#ifndef EGG_COOP_PRIO
#define EGG_COOP_PRIO 0    // s_setprio of the waves inside the cooperative finish (0: unchanged)
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

// One ray of render_scene's trace loop (:190-231) as state that a loop can leave and another can resume.  The trace only FINDS the
// hit; what the reference does inside the loop at the hit (`:205-228`: depth, the 20-step shadow march of ground pixels, the flat
// colours, `break`) runs after the loop, once per wave with all of its hit lanes, instead of once per distinct hit iteration of the
// wave with the few lanes that hit in that iteration.  Per lane the same operations on the same values in the same order.
// 1920x1080: 0.54 -> 0.27 ms.  (Trace and shadow march as ONE loop around one copy of the sdf — lanes with a ground hit start their
// shadow march while neighbours still trace — is slower: 0.283 vs 0.273 ms, 4K 0.68 vs 0.64; the per-lane phase logic costs more
// than the shorter waves save.)
struct EggRay { float t; bool done, hit; int mat; v3 hp; int steps; int id; };

// trace steps [i0, i1) of the lanes not done yet
template <bool CULL, class W>
__device__ __forceinline__ void egg_trace_steps(const FrameEgg& F, v3 ro, v3 rd, EggRay& r, int i0, int i1, W& w) {
    if (r.done) return;
    for (int i = i0; i < i1; ++i) {                         // render_scene :190-231
        if (EGG_PRIO_STEP > 0 && i == EGG_PRIO_STEP) __builtin_amdgcn_s_setprio(EGG_PRIO);   // a long wave: ahead of the short ones on its SIMD
        const v3 p = ro + rd * r.t;
        const D2 d = egg_sdf<CULL>(F, p, w);
#ifdef SBX_EGG_DEBUG
        if (r.id == SBX_EGG_DEBUG) printf("normal step %d t %.9g -> d %.9g m %g\n", i, r.t, d.d, d.m);
#endif
        if (r.t > 15.f) { r.done = true; break; }
        if (d.d < 0.001f) { r.hit = true; r.mat = (int)d.m; r.hp = p; r.done = true; break; }
        r.t += d.d;
#ifdef SBX_EGG_STATS
        ++r.steps;
#endif
    }
}

// THE COOPERATIVE FINISH (round 6).  One wave issues at most one VALU instruction per ~5 cycles whatever its instruction-level
// parallelism (profiles/r02_ubench_issue.txt, W = 1), so a ray that grazes the egg's legs for all 80 steps — ~600 instructions of
// sdf() per step with nothing left to cull — costs its wave 80 x 1.2 us however few of its lanes still march, and one launch cannot
// end before that wave (profiles/r04_egg_lone_wave.txt: 0.09-0.13 ms ALONE on the chip; the census of round 4: a 125 us tail with
// < 7 % of the wave slots in use).  A step cannot start before the previous one's distance is known; what CAN run in parallel is
// the union inside one step.  So a workgroup is four waves (a 64 x 4 pixel strip), and once at most 64 of its 256 rays are still
// marching (counted after EGG_COOP_K0 steps and every EGG_COOP_DK after that, one LDS word per wave and a barrier), those rays are
// packed into the 64 lanes of EVERY wave and each wave evaluates ONE part of the union for all of them —
//     wave 0: left leg     wave 1: right leg     wave 2: the egg (three spheres, two smooth-mins)     wave 3: wheel and both feet
// — writes its distances to LDS, and after one barrier all four fold the five values with op_add2 in sdf()'s own order (:140-143;
// a strict `<`, so a tie keeps the reference's winner) and advance their identical copies of the rays.  A step costs the slowest
// part (a Bezier tube, ~210 instructions) plus ~60 for the point, the exchange and the fold, instead of the whole union.
// Same bits: every member is evaluated by the same expressions on the same point (-ffp-contract=off: an expression's value does not
// depend on which wave computes it); a part skipped because it is far (bezier_far, egg_far: the culls of egg_sdf, against the
// ground's distance, which bounds the union from above) enters as a value that cannot win, as in egg_sdf.  Steps on which every
// ray left is far from everything but the ground (egg_far) skip the exchange altogether — the four waves decide that from the
// same numbers, so they agree without talking.
struct EggCoopLds {
    int cnt[2][4];              // rays still marching, per wave; two sets: a fast wave may write the next count while a slow one reads
    float rd[3][64], t[64];     // the packed rays: direction and distance marched
    float part[2][5][64];       // left leg, right leg, egg, feet, wheel; two sets, by exchange parity, for the same reason
    float res[4][64];           // hit | material << 1 | steps << 8, hit point
#ifdef SBX_EGG_DEBUG
    int id[64];
#endif
};

template <bool CULL, class W>
__device__ __forceinline__ void egg_coop_finish(const FrameEgg& F, v3 ro, v3 rd, EggRay& r, int i0, bool live, unsigned long long mask,
                                                int base, int S, EggCoopLds& L, W& w) {
    const int lane = (int)threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);  // (an SGPR: the compiler cannot know it is uniform)
    const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    if (live) { L.rd[0][slot] = rd.x; L.rd[1][slot] = rd.y; L.rd[2][slot] = rd.z; L.t[slot] = r.t; }
#ifdef SBX_EGG_DEBUG
    if (live) L.id[slot] = r.id;
#endif
    __syncthreads();
    const bool active = lane < S;
    const v3 crd = V3(L.rd[0][lane], L.rd[1][lane], L.rd[2][lane]);          // (lanes >= S: stale words of a ray never advanced)
    EggRay c;
    c.t = L.t[lane]; c.done = !active; c.hit = false; c.mat = 0; c.steps = 0;       // (c.hp: after the loop, from c.t — registers)
    const float inf = u2f(0x7f800000u), thick = .05f;
    const float mat_egg = 1.f, mat_bike = 2.f, mat_ground = 3.f;              // :17-20
    if (EGG_COOP_PRIO > 0) __builtin_amdgcn_s_setprio(EGG_COOP_PRIO);
    int ex = 0;
    for (int i = i0; i < 80; ++i) {
        if (__builtin_amdgcn_ballot_w64(!c.done) == 0ull) break;             // the same word in all four waves
        const v3 P = ro + crd * c.t;
        const D2 ground = {dot(V3(0.f, 1.f, 0.f), P) + (1.2f + 0.5f), mat_ground};       // sd_plane :136-138
        D2 d = ground;
        const bool need = !c.done && !(CULL && egg_far(F, P, ground.d));
        if (__builtin_amdgcn_ballot_w64(need) != 0ull) {                     // ... and so is this one
            const v3 p = mul(F.rot_y, P) - V3(0, 0.5f, 3.5f);                // :40-41
            float (*part)[64] = L.part[ex & 1];
            if (need) {                   // (a lane that is done or far evaluates nothing: it must not record a root either)
                // One store after the chain, at an address that does not depend on the branch taken: with a store in every branch
                // hipcc (ROCm 7.2) sinks them into one store whose address is a phi, and the structurised code of the LAST branch
                // never sets that address register (seen in the listing: wave 3 stored the wheel's distance through a stale s22;
                // every pixel whose ray met the cooperative finish came out as a wheel hit).
                float val;
                if (wave == 0) {                                             // :102-118.  (Two copies of the tube rather than one with a
                    const bool far = CULL && __builtin_amdgcn_ballot_w64(!bezier_far(F.leg_l, p, thick, ground.d)) == 0ull;   // selected
                    val = far ? inf : sd_bezier_x(F.leg_l, p, thick, w);                            // frame: the select of 18 kernel
                } else if (wave == 1) {                                                            // arguments lands in VGPRs and spills)
                    const bool far = CULL && __builtin_amdgcn_ballot_w64(!bezier_far(F.leg_r, p, thick, ground.d)) == 0ull;
                    val = far ? inf : sd_bezier_x(F.leg_r, p, thick, w);
                } else if (wave == 2) {                                      // :47-53
                    const float egg_y = 0.65f;
                    const float egg_m = w.length(p - V3(0, egg_y, 0)) - 0.475f;
                    const float egg_b = w.length(p - V3(0, egg_y - 0.45f, 0)) - 0.25f;
                    const float egg_t = w.length(p - V3(0, egg_y + 0.45f, 0)) - 0.25f;
                    const float egg_1 = op_blend(egg_m, egg_b, .5f);
                    val = op_blend(egg_1, egg_t, .5f);
                } else {                                                     // :120-134
                    const D2 left_foot = {sd_cylinder0<false>(F.foot_l, p + F.left_foot, thick, w), mat_egg};
                    const D2 right_foot = {sd_cylinder0<false>(F.foot_r, p + F.right_foot, thick, w), mat_egg};
                    part[3][lane] = op_add2(left_foot, right_foot).d;
                    const v3 pw = p + V3(0, 1.2f, 0);
                    val = w.length(V2(w.length(V2(pw.x, pw.y)) - 1.f, pw.z)) - .03f;             // sd_torus sdf.h:75-83
                }
                part[wave + (wave == 3)][lane] = val;
            }
            __syncthreads();
            if (need) {
                const D2 feet = {part[3][lane], mat_egg}, bike = {part[4][lane], mat_bike}, egg = {part[2][lane], mat_egg};
                const D2 _1 = op_add2(feet, bike);                           // :140-143
                const D2 _2 = op_add2(egg, _1);
                const D2 legs = op_add2(D2{part[0][lane], mat_egg}, D2{part[1][lane], mat_egg});
                const D2 _3 = op_add2(legs, _2);
                d = op_add2(ground, _3);
            }
            ++ex;
        }
#ifdef SBX_EGG_DEBUG
        if (!c.done && L.id[lane] == SBX_EGG_DEBUG)
            printf("coop wave %d lane %d of %d step %d t %.9g ground %.9g need %d ex %d parts %.9g %.9g %.9g %.9g %.9g -> d %.9g m %g\n", wave, lane, S, i,
                   c.t, ground.d, (int)need, ex, L.part[(ex - 1) & 1][0][lane], L.part[(ex - 1) & 1][1][lane], L.part[(ex - 1) & 1][2][lane],
                   L.part[(ex - 1) & 1][3][lane], L.part[(ex - 1) & 1][4][lane], d.d, d.m);
#endif
        if (!c.done) {
            if (c.t > 15.f) c.done = true;
            else if (d.d < 0.001f) { c.hit = true; c.mat = (int)d.m; c.done = true; }      // (c.t stays: the hit point is ro + crd * c.t)
            else {
                c.t += d.d;
#ifdef SBX_EGG_STATS
                ++c.steps;
#endif
            }
        }
    }
    if (EGG_COOP_PRIO > 0) __builtin_amdgcn_s_setprio(0);
    if (wave == 0 && active) {
        const v3 hp = ro + crd * c.t;                                        // = P of the step that hit (c.t was not advanced)
        L.res[0][lane] = u2f((c.hit ? 1u : 0u) | ((unsigned)c.mat << 1) | ((unsigned)c.steps << 8));
        L.res[1][lane] = hp.x; L.res[2][lane] = hp.y; L.res[3][lane] = hp.z;
    }
    __syncthreads();
    if (live) {
        const unsigned k = f2u(L.res[0][slot]);
        r.hit = (k & 1u) != 0u; r.mat = (int)((k >> 1) & 0x7fu); r.steps += (int)(k >> 8);
        r.hp = V3(L.res[1][slot], L.res[2][slot], L.res[3][slot]);
    }
    r.done = true;
}

#ifndef EGG_WITNESS
#define EGG_WITNESS 1      // five-instruction square roots with a recorded domain (sbx_sdf.h Wit): 0 = the IEEE roots only
#endif

// One pixel up to (colour, depth) — render_scene :190-231 — with the roots of witness `w`.  `coop` (the same in every thread of the
// workgroup): count the marching rays and finish the last <= 64 together; every thread of the workgroup must then be here.
template <bool CULL, class W>
__device__ __forceinline__ void egg_pixel(const FrameEgg& F, v2 pc, bool valid, bool coop, EggCoopLds& L, W& w, v3& color, float& depth,
                                          int& st_trace, int& st_shadow, int id = 0) {
    const v3 ro = F.cam.eye, rd = primary_dir(F.cam, pc, w);
    depth = -1e8f;                                          // :188, fresh per pixel
    color = V3(.1f, .1f, .7f);                              // background :9-12
    EggRay r;
    r.t = 0.f; r.done = !valid; r.hit = false; r.mat = 0; r.hp = V3(0, 0, 0); r.steps = 0; r.id = id;
    if (!(EGG_COOP && coop)) {
        egg_trace_steps<CULL>(F, ro, rd, r, 0, 80, w);
    } else {
        const int lane = (int)threadIdx.x & 63;
        const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        int i = 0, iend = EGG_COOP_K0, set = 0;
        for (;;) {                                          // every condition below is the same in all four waves
            egg_trace_steps<CULL>(F, ro, rd, r, i, iend, w);
            i = iend;
            if (i >= 80) break;
            const bool live = !r.done;
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(live);
            if (lane == 0) L.cnt[set][wave] = __popcll(mask);
            __syncthreads();
            // (read into SGPRs: the compiler cannot know that an LDS word is the same in every lane, and a branch it takes for
            // divergent is a masked region that a wave may walk through with no lane active — barriers and all)
            const int c0 = __builtin_amdgcn_readfirstlane(L.cnt[set][0]), c1 = __builtin_amdgcn_readfirstlane(L.cnt[set][1]);
            const int c2 = __builtin_amdgcn_readfirstlane(L.cnt[set][2]), c3 = __builtin_amdgcn_readfirstlane(L.cnt[set][3]);
            set ^= 1;
            const int S = c0 + c1 + c2 + c3;
            if (S == 0) break;
            if (S <= 64) {
                const int base = wave == 0 ? 0 : wave == 1 ? c0 : wave == 2 ? c0 + c1 : c0 + c1 + c2;
                egg_coop_finish<CULL>(F, ro, rd, r, i, live, mask, base, S, L, w);
                break;
            }
            iend = i + EGG_COOP_DK < 80 ? i + EGG_COOP_DK : 80;
        }
    }
#ifdef SBX_EGG_STATS
    st_trace = r.steps;
    st_shadow = (r.hit && r.mat == 3) ? 1 : 0;
#endif
    if (r.hit) {
        if (r.mat == 1 || r.mat == 2) depth = fmax_(depth, r.hp.z);
        float s = 1.f;
        if (r.mat == 3) {
            const v3 sh_dir = V3(0, 1, 1);
            s = egg_shadowmarch<CULL>(F, r.hp + sh_dir * 0.05f, sh_dir, w);
        }
        v3 base = V3(1, 1, 1);                              // illuminate :29-35
        if (r.mat == 3) base = V3(13.f / 255.f, 104.f / 255.f, 0.f / 255.f);
        else if (r.mat == 1) base = V3(0.9f, 0.95f, 0.95f);
        else if (r.mat == 2) base = V3(.2f, .2f, .2f);
        color = base * s;
    }
}

// Where the cooperative finish is worth its barriers: workgroups with a pixel under the projection of the sphere around everything
// but the ground (in point_cam units: the same extents as the hot rectangle, egg_extents below) — only there can a ray graze
// anything.  Elsewhere the four waves of a workgroup never meet.  all = 1: no usable projection (the camera is inside the sphere):
// every workgroup counts.
struct CoopBox { float x0, x1, y0, y1; int all; };

// WIT: 0 = IEEE roots; 1 = witnessed roots (the shipped form); 2 = the same with the witness's lower edge at 1.0, so that waves
// near any primitive's axis DO record and re-run (sbx_set_variant 2: the test of the re-run path — same frame required)
template <bool CULL, int WIT>
__global__ void __launch_bounds__(64 * EGG_TX) k_egg(FrameEgg F, RowMap M, float* __restrict__ out, HotRect hot, CoopBox box) {
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
    __shared__ EggCoopLds L;
    int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    if (EGG_HOT_FIRST && hot.w > 0) hot_first_tile(hot, (int)gridDim.x, bx, by);          // wave-uniform
    const Pixel px0 = pixel_of<EGG_TW, EGG_TX>(M, (int)threadIdx.x, bx, by, (int)gridDim.y);
    if (!EGG_COOP && !px0.valid) return;
    const v2 pc0 = point_cam(F.cam, px0.fx, px0.fy);
    bool coop = false;
    if (EGG_COOP)           // (an invalid pixel stays: its wave's barriers need it.  It never marches and never stores.)
        coop = __syncthreads_or(px0.valid && (box.all || (pc0.x >= box.x0 && pc0.x <= box.x1 && pc0.y >= box.y0 && pc0.y <= box.y1))) != 0;
#ifdef EGG_COOP_NEVER      // A/B: the four-wave workgroup and all of its code, but no workgroup ever counts
    coop = false;
#endif
    float depth;
    v3 color;
    if (WIT != 0) {
        Wit<true> w;
        if (WIT == 2) w.lo = 0x3F800000u;
        egg_pixel<CULL>(F, pc0, px0.valid, coop, L, w, color, depth, st_trace, st_shadow, px0.x | (px0.y << 16));
        // some lane took a root outside the proved interval: the IEEE forms.  (With the cooperative finish a wave's roots may have been
        // another wave's rays: the workgroup re-runs together.)
        const bool again = coop ? __syncthreads_or(w.bad) != 0 : __builtin_amdgcn_ballot_w64(w.bad) != 0ull;
        if (again) {
            Wit<false> w0;
            egg_pixel<CULL>(F, pc0, px0.valid, coop, L, w0, color, depth, st_trace, st_shadow, px0.x | (px0.y << 16));
        }
    } else {
        Wit<false> w0;
        egg_pixel<CULL>(F, pc0, px0.valid, coop, L, w0, color, depth, st_trace, st_shadow, px0.x | (px0.y << 16));
    }
    // The pixel's place and point_cam again, from a thread id the compiler cannot recognise: kept across the march they are six
    // VGPRs the whole pixel long, and the march is what needs registers
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const Pixel px = pixel_of<EGG_TW, EGG_TX>(M, tid, bx, by, (int)gridDim.y);
    if (EGG_COOP && !px.valid) return;
    const v2 pc = point_cam(F.cam, px.fx, px.fy);
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
        o4.w = __uint_as_float((unsigned)st_trace | ((unsigned)st_shadow << 8));      // per LANE: its own trace steps, shadow march or not
        reinterpret_cast<float4*>(out)[px.idx] = o4;
        return;
    }
#endif
    store_rgba(M, out, px.idx, to_srgb(color));
}

// The projection of the sphere (F.oc, F.orad) around everything but the ground (sdf()'s p space: P = rot_y^T (p + (0, .5, 3.5))) in
// point_cam units: x = X / Z and y = Y / Z over the sphere.  false: the camera is inside or beside the sphere.
static bool egg_extents(const FrameEgg& F, float& pxa, float& pxb, float& pya, float& pyb) {
    const v3 c = mul(transpose(F.rot_y), F.oc + V3(0, 0.5f, 3.5f));
    const v3 v = c - F.cam.eye;
    const float depth = dot(v, F.cam.fwd), r = F.orad;
    if (!(depth > r * 1.05f)) return false;
    // the planes through the eye that contain the camera's up axis and touch the sphere — in the (right, fwd) plane the sphere is a
    // circle of radius r at (vx, depth), the tangents from the origin are at phi +- asin(r / d)
    const float vx = dot(v, F.cam.right), vy = dot(v, F.cam.up);
    auto extent = [&](float side, float& lo, float& hi) {
        const float d = sqrt_(side * side + depth * depth);
        const float phi = std::atan2(side, depth), al = std::asin(std::min(1.f, r / d));
        const float a = std::max(phi - al, -1.5f), b = std::min(phi + al, 1.5f);
        lo = std::tan(a); hi = std::tan(b);
    };
    extent(vx, pxa, pxb);
    extent(vy, pya, pyb);
    return pxa == pxa && pxb == pxb && pya == pya && pyb == pyb;
}

// The tiles under that projection, for a launch that covers whole rows of the frame from row M.y0 (a contiguous strip; other maps:
// plain order).  A hint about cost: off by any amount it only changes the order in which the same workgroups run.
static HotRect egg_hot_rect(const FrameEgg& F, const RowMap& M, dim3 grid) {
    HotRect none{0, 0, 0, 0};
    if (!EGG_HOT_FIRST || M.nranks != 1 || M.frag || M.span_mode || M.r0 != 0) return none;
    float pxa, pxb, pya, pyb;
    if (!egg_extents(F, pxa, pxb, pya, pyb)) return none;
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
    CoopBox box{0.f, 0.f, 0.f, 0.f, 1};
    if (egg_extents(F, box.x0, box.x1, box.y0, box.y1)) box.all = 0;
    static const int pad = []() { const char* e = std::getenv("SBX_DEBUG_LDS_PAD"); return e ? std::atoi(e) : EGG_LDS_PAD; }();
    if (variant == 1) hipLaunchKernelGGL((k_egg<false, 0>), grid, dim3(64 * EGG_TX), (size_t)pad, s, F, M, out, hot, box);
    else if (variant == 2) hipLaunchKernelGGL((k_egg<true, 2>), grid, dim3(64 * EGG_TX), (size_t)pad, s, F, M, out, hot, box);
    else if (variant == 3) hipLaunchKernelGGL((k_egg<true, 0>), grid, dim3(64 * EGG_TX), (size_t)pad, s, F, M, out, hot, box);
    else hipLaunchKernelGGL((k_egg<true, EGG_WITNESS>), grid, dim3(64 * EGG_TX), (size_t)pad, s, F, M, out, hot, box);
}

}  // namespace sbx
