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
#define EGG_COOP 0         // 1: long, expensive rays leave their wave for a queue that four-wave FINISHER workgroups serve (see "THE
#endif                     // FINISHERS" below).  Bit-exact, measured, and NOT faster on this part: 0.225-0.35 ms against 0.215 for one
                           // 1920x1080 launch, 0.21-0.24 against 0.136 ms per frame with two in flight (profiles/r06_egg_finishers.txt)
#ifndef EGG_TX
#define EGG_TX 1           // waves per workgroup: 1 (4: the same single launch, 7 % slower with frames in flight at 1080p, 3 % at 4K;
#endif                     // census round 6: 4500 instead of 5900 waves resident — a workgroup's slots come back all at once)
#ifndef EGG_COOP_K1
#define EGG_COOP_K1 32     // trace steps before a wave first offers rays to the queue
#endif
#ifndef EGG_COOP_DK
#define EGG_COOP_DK 4      // ... and between later offers
#endif
#ifndef EGG_COOP_NF
#define EGG_COOP_NF 128    // finisher workgroups (four waves each) of one launch
#endif
#ifndef EGG_COOP_DMAX
#define EGG_COOP_DMAX 1e30f   // a ray is offered only if its last step was shorter than this (a grazing ray's steps are short)
#endif
#ifndef EGG_COOP_MINB
#define EGG_COOP_MINB 32   // a finisher takes fewer rays than this only after EGG_COOP_IDLE looks at a queue that did not grow
#endif
#ifndef EGG_COOP_IDLE
#define EGG_COOP_IDLE 3
#endif
#ifndef EGG_COOP_PRIO
#define EGG_COOP_PRIO 0    // s_setprio of the finisher's waves (0: unchanged)
#endif
#ifndef EGG_Q_CAP
#define EGG_Q_CAP 65536    // rays one launch's queue holds (32 B each); a wave that finds it full keeps its rays
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
struct EggRay { float t; bool done, hit; int mat; v3 hp; int steps; float dl; };     // dl: the last step's length

// trace steps [i0, i1) of the lanes not done yet
template <bool CULL, class W>
__device__ __forceinline__ void egg_trace_steps(const FrameEgg& F, v3 ro, v3 rd, EggRay& r, int i0, int i1, W& w) {
    if (r.done) return;
    for (int i = i0; i < i1; ++i) {                         // render_scene :190-231
        if (EGG_PRIO_STEP > 0 && i == EGG_PRIO_STEP) __builtin_amdgcn_s_setprio(EGG_PRIO);   // a long wave: ahead of the short ones on its SIMD
        const v3 p = ro + rd * r.t;
        const D2 d = egg_sdf<CULL>(F, p, w);
        if (r.t > 15.f) { r.done = true; break; }
        if (d.d < 0.001f) { r.hit = true; r.mat = (int)d.m; r.hp = p; r.done = true; break; }
        r.t += d.d;
        if (EGG_COOP) r.dl = d.d;
#ifdef SBX_EGG_STATS
        ++r.steps;
#endif
    }
}

// THE FINISHERS (round 6).  One wave issues at most one VALU instruction per ~5 cycles whatever its instruction-level parallelism
// (profiles/r02_ubench_issue.txt, W = 1), so a ray that grazes the egg's legs for all 80 steps — ~600 instructions of sdf() per step
// with nothing left to cull — costs its wave 80 x 1.2 us however few of its lanes still march, and one launch cannot end before that
// wave (profiles/r04_egg_lone_wave.txt: 0.09-0.13 ms ALONE on the chip; the census: the chip is full for 120 us and then runs a
// 100 us tail with < 7 % of the wave slots in use).  A step cannot start before the previous one's distance is known; what CAN run
// in parallel is the union inside one step.  So:
//   * a wave of k_egg that is still marching after EGG_COOP_K1 steps hands those of its rays that are near the scene (not egg_far:
//     the others take ~28 instructions a step) to a QUEUE in device memory — fragCoord, distance marched, step count, pixel index,
//     32 bytes a ray — and goes on without them;
//   * k_egg_finish, a second launch that runs BESIDE k_egg (its own stream, forked from and joined to the caller's), is EGG_COOP_NF
//     workgroups of FOUR waves.  A workgroup claims 64 queue slots, waits for their rays, and each of its waves evaluates ONE part
//     of the union for all 64 —
//         wave 0: left leg     wave 1: right leg     wave 2: the egg (three spheres, two smooth-mins)     wave 3: wheel and both feet
//     — writes its distances to LDS, and after one barrier all four fold the five values with op_add2 in sdf()'s own order
//     (:140-143; a strict `<`, so a tie keeps the reference's winner) and advance their identical copies of the rays: a step costs
//     the slowest part (a Bezier tube, ~210 instructions) plus ~60 for the point, the exchange and the fold, and it serves 64 rays,
//     where the wave that handed them over paid the whole union for a handful.  The same for the shadow march of those that land on
//     the ground; then the colours, the bars, the store — the rest of the pixel, by the same functions.
// Same bits: a ray's state crosses the queue exactly (its direction is recomputed from fragCoord by the function that computed it);
// every member is evaluated by the same expressions on the same point (-ffp-contract=off: an expression's value does not depend on
// which wave computes it); a part skipped because it is far (bezier_far, egg_far: the culls of egg_sdf, against the ground's
// distance, which bounds the union from above) enters as a value that cannot win, as in egg_sdf.  Steps on which every ray left is
// far from everything but the ground skip the exchange altogether — the four waves decide that from the same numbers, so they
// agree without talking.
// (Round 6 first built the cooperation INSIDE k_egg — four-wave workgroups, the last <= 64 rays of a workgroup finished by its own
// waves.  Bit-exact, and the longest wave fell from 190 to 140 us, but workgroups of four give their wave slots back all at once:
// 4500 instead of 5900 waves resident, the chip-full phase 120 -> 150 us, one launch 0.216 against 0.215 ms.
// profiles/r06_egg_design1.txt.)
struct EggRec { unsigned fx, fy, t, i, idx_lo, idx_hi, tag, pad; };      // (floats as their bits) tag == the launch's sequence number: a ray
constexpr int EGG_Q_SHARDS = 64;                    // the producer waves' reports are spread over this many words, one per 128-byte line:
struct EggQueue {                                   // one word takes ~88 atomics per microsecond, a launch reports 6000 times
    unsigned reserve, pad0[31];                     // slots handed out to producers
    unsigned commit, pad1[31];                      // rays whose words and tag are written (in any order: a claimed slot may lag, briefly)
    unsigned head, out, pad[30];                    // slots claimed by finishers; finishers gone (pad: the census build's sums)
    unsigned alldone, pad2[31];                     // shards of `done` that have all their reports: what the finishers poll — ONE word
    unsigned done[EGG_Q_SHARDS][32];                // producer waves finished, [shard][0]
    EggRec rec[EGG_Q_CAP];
};
static_assert(!EGG_COOP || EGG_TX == 1, "the finishers count k_egg's offering WAVES as workgroups");
struct EggQArg { EggQueue* q; unsigned seq, expected; int nf; };     // q == nullptr: no queue (plain kernel)

// The queue's words cross between compute units (and XCDs, whose L2s do not snoop each other) WITHOUT fences: an agent-scope release
// is a write-back of the whole L2 and an acquire an invalidate of it — six thousand reporting waves doing that took k_egg from 0.20 to
// 0.31 ms.  Instead every queue word is written and read by agent-scope relaxed atomics (write-through / cache-bypassing `sc1`
// accesses), and the ORDER a reader relies on is made by the writer waiting for its stores' acknowledgements (s_waitcnt vmcnt(0))
// before it issues the word that announces them: a ray's words, wait, its tag, wait, the wave's report.
__device__ __forceinline__ unsigned egg_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void egg_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void egg_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// have all producer waves reported?  (the last report of a shard counts the shard in `alldone`: one word to look at)
__device__ __forceinline__ bool egg_all_reported(const EggQArg& A, int lane) {
    unsigned v = 0;
    if (lane == 0) v = egg_ld(&A.q->alldone);
    v = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    return v >= (A.expected < (unsigned)EGG_Q_SHARDS ? A.expected : (unsigned)EGG_Q_SHARDS);
}

// A wave offers the rays in `want` (all lanes of the wave are here).  true: they are in the queue and no longer this wave's.
__device__ __forceinline__ bool egg_export(const EggQArg& A, unsigned long long want, bool mine, float fx, float fy, float t, int i, size_t idx) {
    const int n = __popcll(want);
    const int lane = (int)threadIdx.x & 63;
    unsigned pos = 0;
    if (lane == 0) pos = __hip_atomic_fetch_add(&A.q->reserve, (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pos = (unsigned)__builtin_amdgcn_readfirstlane((int)pos);
    if (pos + (unsigned)n > (unsigned)EGG_Q_CAP) return false;          // full: the slots stay untagged, which a finisher reads as empty
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
    EggRec* r = &A.q->rec[pos + (unsigned)rank];
    if (mine) {
        egg_st(&r->fx, f2u(fx)); egg_st(&r->fy, f2u(fy)); egg_st(&r->t, f2u(t)); egg_st(&r->i, (unsigned)i);
        egg_st(&r->idx_lo, (unsigned)idx); egg_st(&r->idx_hi, (unsigned)((unsigned long long)idx >> 32));
    }
    egg_drain();                                                         // the ray before its tag
    if (mine) egg_st(&r->tag, A.seq);
    egg_drain();
    if (lane == 0) __hip_atomic_fetch_add(&A.q->commit, (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    egg_drain();                                                         // ... the tags before the count, the count before this wave's report
    return true;
}

#ifndef EGG_WITNESS
#define EGG_WITNESS 1      // five-instruction square roots with a recorded domain (sbx_sdf.h Wit): 0 = the IEEE roots only
#endif

// One pixel up to (colour, depth) — render_scene :190-231 — with the roots of witness `w`.  `offer` (wave-uniform): this wave may
// hand rays to the queue; `gone`: this lane's ray is a finisher's now (it computes nothing more and stores nothing).
template <bool CULL, class W>
__device__ __forceinline__ void egg_pixel(const FrameEgg& F, v2 pc, bool valid, bool offer, const EggQArg& A, const Pixel& px, bool& gone,
                                          W& w, v3& color, float& depth, int& st_trace, int& st_shadow) {
    const v3 ro = F.cam.eye, rd = primary_dir(F.cam, pc, w);
    depth = -1e8f;                                          // :188, fresh per pixel
    color = V3(.1f, .1f, .7f);                              // background :9-12
    EggRay r;
    r.t = 0.f; r.done = !valid || gone; r.hit = false; r.mat = 0; r.hp = V3(0, 0, 0); r.steps = 0; r.dl = 0.f;
    // ONE copy of the trace loop (each is a copy of sdf(), ~4.5 KB of a 64 KB instruction cache): a wave that may not offer runs its
    // 80 steps in one go, one that may stops after EGG_COOP_K1 and then every EGG_COOP_DK
    int i = 0, iend = (EGG_COOP && offer) ? EGG_COOP_K1 : 80;
#pragma clang loop unroll(disable)
    for (;;) {                                              // (wave-uniform conditions)
        egg_trace_steps<CULL>(F, ro, rd, r, i, iend, w);
        i = iend;
        if (i >= 80 || __builtin_amdgcn_ballot_w64(!r.done) == 0ull) break;
        // near the scene, hence expensive, hence worth a finisher's lane.  A wave in which ANY lane has taken a root outside the
        // witness's interval keeps its rays: it is going to run again with the IEEE roots (k_egg), from the start.
        const v3 P = ro + rd * r.t;
        const bool want = !r.done && r.dl < EGG_COOP_DMAX && !egg_far(F, P, dot(V3(0.f, 1.f, 0.f), P) + (1.2f + 0.5f));
        const unsigned long long wm = __builtin_amdgcn_ballot_w64(want);
        if (wm != 0ull && __builtin_amdgcn_ballot_w64(w.bad) == 0ull && egg_export(A, wm, want, px.fx, px.fy, r.t, i, px.idx)) {
            if (want) { gone = true; r.done = true; }
        }
        iend = i + EGG_COOP_DK < 80 ? i + EGG_COOP_DK : 80;
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

// bars overlay :233-251
__device__ __forceinline__ v3 egg_bars(v3 color, float pcx, float depth) {
    const float bar_factor = 1.0f - smoothstep_(0.0f, 0.01f, abs_((abs_(pcx) - 0.6f)) - 0.05f);
    const float depth_factor = 1.f - step_(1.f, depth);
    return abs3(mix3(color, V3(.6f, .6f, .6f), bar_factor * depth_factor));
}

// WIT: 0 = IEEE roots; 1 = witnessed roots (the shipped form); 2 = the same with the witness's lower edge at 1.0, so that waves
// near any primitive's axis DO record and re-run (sbx_set_variant 2: the test of the re-run path — same frame required)
template <bool CULL, int WIT>
__global__ void __launch_bounds__(64 * EGG_TX) k_egg(FrameEgg F, RowMap M, float* __restrict__ out, HotRect hot, EggQArg A) {
#ifdef SBX_EGG_STATS
    const unsigned long long st_t0 = __builtin_amdgcn_s_memrealtime();      // census build (tools/egg_census.py): 100 MHz counter
#endif
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();      // (the dispatch order's cost table, RowMap.cost)
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
    // The waves that may hand rays over are those of the hot rectangle — the first hot.w * hot.h workgroups — and the finishers
    // count exactly those home (A.expected): every one of them reports below, whatever its pixels did.
    const bool offer = EGG_COOP && A.q != nullptr && by * (int)gridDim.x + bx < hot.w * hot.h;
    if (EGG_HOT_FIRST && hot.w > 0 && !M.order) hot_first_tile(hot, (int)gridDim.x, bx, by);          // wave-uniform (a dispatch-order table, once there is one, knows better)
    const Pixel px = pixel_of<EGG_TW, EGG_TX>(M, (int)threadIdx.x, bx, by, (int)gridDim.y);
    if (!(EGG_COOP && offer) && !px.valid) return;           // (an offering wave keeps its invalid lanes: lane 0 reports for the wave)
    const v2 pc = point_cam(F.cam, px.fx, px.fy);
    float depth;
    v3 color;
    bool gone = false;
    if (WIT != 0) {
        Wit<true> w;
        if (WIT == 2) w.lo = 0x3F800000u;
        egg_pixel<CULL>(F, pc, px.valid, offer, A, px, gone, w, color, depth, st_trace, st_shadow);
        if (__builtin_amdgcn_ballot_w64(w.bad) != 0ull) {      // some lane took a root outside the proved interval: the IEEE forms
            Wit<false> w0;                                     // (rays handed over before that were exact, and stay handed over)
            egg_pixel<CULL>(F, pc, px.valid, offer, A, px, gone, w0, color, depth, st_trace, st_shadow);
        }
    } else {
        Wit<false> w0;
        egg_pixel<CULL>(F, pc, px.valid, offer, A, px, gone, w0, color, depth, st_trace, st_shadow);
    }
    if (EGG_COOP && offer) {                                   // this wave hands over nothing more (its tags are acknowledged: egg_export)
#ifndef EGG_DBG_NOREPORT
        if (((int)threadIdx.x & 63) == 0) {                    // (the offering waves are workgroups 0 .. expected - 1 of the launch)
            const unsigned id = blockIdx.y * gridDim.x + blockIdx.x, sh = id % EGG_Q_SHARDS;
            const unsigned full = A.expected / EGG_Q_SHARDS + (sh < A.expected % EGG_Q_SHARDS ? 1u : 0u);
            if (__hip_atomic_fetch_add(&A.q->done[sh][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == full)
                __hip_atomic_fetch_add(&A.q->alldone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
        if (!px.valid) return;
    }
    tile_cost_store_at(M, tl_t0, bx, by);                  // (bx, by: after the hot-first mapping)
    color = egg_bars(color, pc.x, depth);
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
        o4.w = __uint_as_float((unsigned)st_trace | ((unsigned)st_shadow << 8) | (gone ? 0x10000u : 0u));   // per LANE: trace steps, shadow march, handed over
        reinterpret_cast<float4*>(out)[px.idx] = o4;
        return;
    }
#endif
    if (gone) return;
    store_rgba(M, out, px.idx, to_srgb(color));
}

// ---- the finishers ---------------------------------------------------------------------------------------------------------
struct EggCoopLds {
    float part[2][5][64];       // left leg, right leg, egg, feet, wheel; two sets, by exchange parity: a fast wave may write the next
                                // exchange while a slow one still reads this one (the barrier of the exchange between orders the reuse)
    float fx[64], fy[64], t[64];
    unsigned i[64];
    unsigned long long idx[64];
    unsigned start, n, last, bad, timeout;
};

// sdf(P) of the lanes in `on`, by the four waves of the workgroup together: every wave calls this with the SAME P and on
template <bool CULL, class W>
__device__ __forceinline__ D2 egg_coop_sdf(const FrameEgg& F, v3 P, bool on, int wave, int lane, EggCoopLds& L, int& ex, W& w) {
    const float inf = u2f(0x7f800000u), thick = .05f;
    const float mat_egg = 1.f, mat_bike = 2.f, mat_ground = 3.f;              // :17-20
    const D2 ground = {dot(V3(0.f, 1.f, 0.f), P) + (1.2f + 0.5f), mat_ground};           // sd_plane :136-138
    D2 d = ground;
    const bool need = on && !(CULL && egg_far(F, P, ground.d));
    if (__builtin_amdgcn_ballot_w64(need) != 0ull) {                         // the same word in all four waves
        const v3 p = mul(F.rot_y, P) - V3(0, 0.5f, 3.5f);                    // :40-41
        float (*part)[64] = L.part[ex & 1];
        if (need) {                       // (a lane that is off or far evaluates nothing: it must not record a root either)
            // One store after the chain, at an address that does not depend on the branch taken: with a store in every branch hipcc
            // (ROCm 7.2) sinks them into one store whose address is a phi, and the structurised code of the LAST branch never sets
            // that address register (seen in the listing: wave 3 stored the wheel's distance through a stale s22; every ray that
            // met the cooperation came out as a wheel hit).
            float val;
            if (wave == 0) {                                                 // :102-118.  (Two copies of the tube rather than one with a
                const bool far = CULL && __builtin_amdgcn_ballot_w64(!bezier_far(F.leg_l, p, thick, ground.d)) == 0ull;   // selected
                val = far ? inf : sd_bezier_x(F.leg_l, p, thick, w);                                // frame: the select of 18 kernel
            } else if (wave == 1) {                                                                // arguments lands in VGPRs and spills)
                const bool far = CULL && __builtin_amdgcn_ballot_w64(!bezier_far(F.leg_r, p, thick, ground.d)) == 0ull;
                val = far ? inf : sd_bezier_x(F.leg_r, p, thick, w);
            } else if (wave == 2) {                                          // :47-53
                const float egg_y = 0.65f;
                const float egg_m = w.length(p - V3(0, egg_y, 0)) - 0.475f;
                const float egg_b = w.length(p - V3(0, egg_y - 0.45f, 0)) - 0.25f;
                const float egg_t = w.length(p - V3(0, egg_y + 0.45f, 0)) - 0.25f;
                const float egg_1 = op_blend(egg_m, egg_b, .5f);
                val = op_blend(egg_1, egg_t, .5f);
            } else {                                                         // :120-134
                const D2 left_foot = {sd_cylinder0<false>(F.foot_l, p + F.left_foot, thick, w), mat_egg};
                const D2 right_foot = {sd_cylinder0<false>(F.foot_r, p + F.right_foot, thick, w), mat_egg};
                part[3][lane] = op_add2(left_foot, right_foot).d;
                const v3 pw = p + V3(0, 1.2f, 0);
                val = w.length(V2(w.length(V2(pw.x, pw.y)) - 1.f, pw.z)) - .03f;                 // sd_torus sdf.h:75-83
            }
            part[wave + (wave == 3)][lane] = val;
        }
        __syncthreads();
        if (need) {
            const D2 feet = {part[3][lane], mat_egg}, bike = {part[4][lane], mat_bike}, egg = {part[2][lane], mat_egg};
            const D2 _1 = op_add2(feet, bike);                               // :140-143
            const D2 _2 = op_add2(egg, _1);
            const D2 legs = op_add2(D2{part[0][lane], mat_egg}, D2{part[1][lane], mat_egg});
            const D2 _3 = op_add2(legs, _2);
            d = op_add2(ground, _3);
        }
        ++ex;
    }
    return d;
}

// The rest of 64 queued pixels, by the four waves together (each holds the same copy; wave 0 stores).  false: a root outside the
// witness's interval was taken somewhere — nothing was stored, run again with the IEEE roots.
template <bool CULL, class W>
__device__ __forceinline__ bool egg_finish_batch(const FrameEgg& F, const RowMap& M, float* __restrict__ out, int n, int wave, int lane,
                                                 EggCoopLds& L, W& w) {
    const bool active = lane < n;
    const float fx = active ? L.fx[lane] : .5f, fy = active ? L.fy[lane] : .5f;
    const v2 pc = point_cam(F.cam, fx, fy);
    const v3 ro = F.cam.eye, rd = primary_dir(F.cam, pc, w);                 // the producer's functions on the producer's numbers
    float t = active ? L.t[lane] : 0.f;
    int i = active ? (int)L.i[lane] : 80;
    bool done = !active, hit = false;
    int mat = 0, ex = 0;
    while (__builtin_amdgcn_ballot_w64(!done) != 0ull) {                     // render_scene :190-231 from step i on
        const v3 P = ro + rd * t;
        const D2 d = egg_coop_sdf<CULL>(F, P, !done, wave, lane, L, ex, w);
        if (!done) {
            if (t > 15.f) done = true;
            else if (d.d < 0.001f) { hit = true; mat = (int)d.m; done = true; }   // (t stays: the hit point is ro + rd * t)
            else { t += d.d; if (++i >= 80) done = true; }
        }
    }
    float depth = -1e8f;
    v3 color = V3(.1f, .1f, .7f);
    const v3 hp = ro + rd * t;
    float s = 1.f;
    {   // shadowmarch :161-186 of the pixels that landed on the ground
        const v3 sh_dir = V3(0, 1, 1), so = hp + sh_dir * 0.05f;
        bool sdone = !(hit && mat == 3);
        float st = 0.f, umbra = 1.f;
        for (int k = 0; k < 20; ++k) {
            if (__builtin_amdgcn_ballot_w64(!sdone) == 0ull) break;
            const v3 P = so + sh_dir * st;
            const D2 d = egg_coop_sdf<CULL>(F, P, !sdone, wave, lane, L, ex, w);
            if (!sdone) {
                if (st > 10.f) sdone = true;
                else if (d.d < 0.001f) { umbra = 0.1f; sdone = true; }
                else { st += d.d; umbra = fmin_(umbra, 15.f * d.d / st); }
            }
        }
        if (hit && mat == 3) s = umbra;
    }
    if (hit) {
        if (mat == 1 || mat == 2) depth = fmax_(depth, hp.z);
        v3 base = V3(1, 1, 1);                              // illuminate :29-35
        if (mat == 3) base = V3(13.f / 255.f, 104.f / 255.f, 0.f / 255.f);
        else if (mat == 1) base = V3(0.9f, 0.95f, 0.95f);
        else if (mat == 2) base = V3(.2f, .2f, .2f);
        color = base * s;
    }
    if (W::fast) {                                          // one verdict for the workgroup
        if (__builtin_amdgcn_ballot_w64(w.bad) != 0ull && lane == 0) L.bad = 1u;
        __syncthreads();
        const bool bad = __builtin_amdgcn_readfirstlane((int)L.bad) != 0;
        __syncthreads();
        if (bad) return false;
    }
    color = egg_bars(color, pc.x, depth);
#ifndef SBX_EGG_STATS                                       // (the census build's frame holds k_egg's per-wave records instead)
    if (wave == 0 && active) store_rgba(M, out, (size_t)L.idx[lane], to_srgb(color));
#endif
    return true;
}

#ifndef EGG_FIN_POLLS
#define EGG_FIN_POLLS (1 << 22)     // polls (~0.5 us each) after which a finisher gives up on the producers: seconds, never reached by a
#endif                              // launch that runs; then the fault word is raised (the frame is incomplete) instead of a hang
__device__ unsigned* g_egg_fault = nullptr;

template <bool CULL, int WIT>
__global__ void __launch_bounds__(256) k_egg_finish(FrameEgg F, RowMap M, float* __restrict__ out, EggQArg A) {
    __shared__ EggCoopLds L;
    const int lane = (int)threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);  // (an SGPR: the compiler cannot know it is uniform)
    if (EGG_COOP_PRIO > 0) __builtin_amdgcn_s_setprio(EGG_COOP_PRIO);
    if (threadIdx.x == 0) { L.bad = 0u; L.timeout = 0u; }
#ifdef SBX_EGG_STATS
    const unsigned long long st_g0 = __builtin_amdgcn_s_memrealtime();
    int st_nb = 0;
    unsigned long long st_first = 0, st_lastb = 0;
#endif
    for (;;) {
#ifdef SBX_EGG_STATS
        const unsigned long long st_c0 = __builtin_amdgcn_s_memrealtime();
#endif
        if (wave == 0) {
            // CLAIM: rays [h, h + k) of the queue, k <= 64 of those written so far (`commit` counts rays written, in whatever order
            // their waves got there; `head` is moved by compare-and-swap).  A finisher does not wait for a full 64 unless rays
            // keep coming: a ray must not sit in the queue while finishers idle — the launch ends with its last ray.  No ray left
            // and every producer wave reported (a wave's count precedes its report): done.
            unsigned h = 0, k = 0;
            int idle = 0, polls = 0;
            unsigned seen = 0;
            for (;;) {
                unsigned c = 0;
                if (lane == 0) { h = egg_ld(&A.q->head); c = egg_ld(&A.q->commit); }
                h = (unsigned)__builtin_amdgcn_readfirstlane((int)h);
                c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
                const unsigned avail = c > h ? c - h : 0u;
                bool take = avail >= (unsigned)EGG_COOP_MINB;
                if (!take) {
                    const bool rep = egg_all_reported(A, lane);               // (read BEFORE the counts are read again)
                    if (rep) {
                        if (lane == 0) { h = egg_ld(&A.q->head); c = egg_ld(&A.q->commit); }
                        h = (unsigned)__builtin_amdgcn_readfirstlane((int)h);
                        c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
                        if (c <= h) { k = 0; break; }                          // final counts: nothing left
                        take = true;
                    } else if (avail > 0u) {
                        idle = (c == seen) ? idle + 1 : 0;
                        seen = c;
                        take = idle >= EGG_COOP_IDLE;
                    }
                }
                if (take) {
                    const unsigned want = (c - h) < 64u ? (c - h) : 64u;
                    unsigned got = 0;
                    if (lane == 0) {
                        unsigned e = h;
                        got = __hip_atomic_compare_exchange_strong(&A.q->head, &e, h + want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
                    }
                    if (__builtin_amdgcn_readfirstlane((int)got) != 0) { k = want; break; }
                    continue;                                               // another finisher moved the head: look again
                }
                if (++polls > EGG_FIN_POLLS) { if (lane == 0) L.timeout = 1u; k = 0; break; }
                __builtin_amdgcn_s_sleep(48);                               // ~1.3 us: 128 finishers looking at three words
            }
            // the k rays: a slot's tag may lag its claim by the moment between a producer's count and a slower producer's tag
            const bool mine = (unsigned)lane < k;
            EggRec* r = &A.q->rec[(h + (unsigned)lane) % (unsigned)EGG_Q_CAP];
            bool have = !mine;
            while (__builtin_amdgcn_ballot_w64(!have) != 0ull) {
                if (!have && egg_ld(&r->tag) == A.seq) have = true;
                if (++polls > EGG_FIN_POLLS) { have = true; if (lane == 0) L.timeout = 1u; }
            }
            if (mine) {
                L.fx[lane] = u2f(egg_ld(&r->fx)); L.fy[lane] = u2f(egg_ld(&r->fy)); L.t[lane] = u2f(egg_ld(&r->t)); L.i[lane] = egg_ld(&r->i);
                L.idx[lane] = (unsigned long long)egg_ld(&r->idx_lo) | ((unsigned long long)egg_ld(&r->idx_hi) << 32);
            }
            if (lane == 0) { L.n = L.timeout ? 0u : k; L.last = (k == 0u) ? 1u : 0u; }
        }
        __syncthreads();
        const int n = __builtin_amdgcn_readfirstlane((int)L.n);
        const bool last = __builtin_amdgcn_readfirstlane((int)L.last) != 0;
#ifdef SBX_EGG_STATS
        const unsigned long long st_b0 = __builtin_amdgcn_s_memrealtime();
#endif
        if (n > 0) {
            bool ok = false;
            if (WIT != 0) {
                Wit<true> w;
                if (WIT == 2) w.lo = 0x3F800000u;
                ok = egg_finish_batch<CULL>(F, M, out, n, wave, lane, L, w);
                if (!ok && threadIdx.x == 0) L.bad = 0u;
            }
            if (!ok) {
                __syncthreads();
                Wit<false> w0;
                egg_finish_batch<CULL>(F, M, out, n, wave, lane, L, w0);
            }
        }
#ifdef SBX_EGG_STATS
        if (n > 0) { if (!st_nb) st_first = st_b0; ++st_nb; st_lastb = __builtin_amdgcn_s_memrealtime(); }
        if (threadIdx.x == 0 && n > 0) {
            __hip_atomic_fetch_add(&A.q->pad[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&A.q->pad[1], (unsigned)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&A.q->pad[2], (unsigned)(__builtin_amdgcn_s_memrealtime() - st_b0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&A.q->pad[3], (unsigned)(st_b0 - st_c0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
        if (last) break;
        __syncthreads();                                    // (the batch's LDS words before the next claim overwrites them)
    }
#ifdef SBX_EGG_STATS
    if (threadIdx.x == 0)
        printf("finisher %3d: start %llu end %llu (%.1f us) batches %d first batch at +%.1f us, last batch done at +%.1f us\n", (int)blockIdx.x, st_g0,
               __builtin_amdgcn_s_memrealtime(), (__builtin_amdgcn_s_memrealtime() - st_g0) * .01, st_nb, st_nb ? (st_first - st_g0) * .01 : 0.,
               st_nb ? (st_lastb - st_g0) * .01 : 0.);
#endif
    if (threadIdx.x == 0) {
        if (L.timeout && g_egg_fault) __hip_atomic_store(g_egg_fault, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // the last finisher out leaves the counters as the next launch on this queue expects them (tags need no reset: the next
        // launch carries another number)
        const unsigned gone = __hip_atomic_fetch_add(&A.q->out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (after this group's last claim returned)
        if (gone + 1u == (unsigned)A.nf) {
#ifdef SBX_EGG_STATS
            printf("finishers: %u batches, %u rays (queue reserve %u), %.1f us in batches, %.1f us waiting for rays (sums over %d workgroups)\n",
                   egg_ld(&A.q->pad[0]), egg_ld(&A.q->pad[1]), egg_ld(&A.q->reserve), egg_ld(&A.q->pad[2]) * .01, egg_ld(&A.q->pad[3]) * .01, A.nf);
            for (int k = 0; k < 4; ++k) egg_st(&A.q->pad[k], 0u);
#endif
            __hip_atomic_store(&A.q->reserve, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&A.q->head, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&A.q->commit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int k = 0; k < EGG_Q_SHARDS; ++k) egg_st(&A.q->done[k][0], 0u);
            egg_st(&A.q->alldone, 0u);
            __hip_atomic_store(&A.q->out, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
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
// plain order).  A hint about cost: off by any amount it only changes the order in which the same workgroups run (and which waves
// may hand rays to the finishers).
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

// ---- the launch's side: queues, the finishers' streams, the events that fork and join them -------------------------------------
// One EggSide per context (sbx_capi.hip), made on first use.  A launch takes the next of EGG_SIDE_RING queues and the next of the
// side streams; a queue is reused only behind the event of the launch that used it last (both of the new launch's streams wait on
// it: nothing on the host blocks).
constexpr int EGG_SIDE_RING = 8, EGG_SIDE_STREAMS = 4;
struct EggSide {
    EggQueue* q[EGG_SIDE_RING] = {};
    unsigned seq[EGG_SIDE_RING] = {};
    hipEvent_t used[EGG_SIDE_RING] = {};
    bool was_used[EGG_SIDE_RING] = {};
    hipStream_t side[EGG_SIDE_STREAMS] = {};
    hipEvent_t fork[EGG_SIDE_RING] = {};
    unsigned next = 0;
    bool ok = false;
};
void* egg_side_create() {
    if (!EGG_COOP) return nullptr;
    EggSide* S = new EggSide;
    bool ok = true;
    for (int k = 0; k < EGG_SIDE_RING && ok; ++k) {
        ok = hipMalloc((void**)&S->q[k], sizeof(EggQueue)) == hipSuccess && hipMemset(S->q[k], 0, sizeof(EggQueue)) == hipSuccess &&
             hipEventCreateWithFlags(&S->used[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&S->fork[k], hipEventDisableTiming) == hipSuccess;
    }
    // The side streams have the device's HIGHEST priority: k_egg is launched first and refills every wave slot it frees from its own
    // 32 400 workgroups; a finisher workgroup needs four free slots on one CU, and at equal priority most of them got theirs only
    // when k_egg had nothing left to launch (trace: the finishers ended 150 us after k_egg).
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
    for (int k = 0; k < EGG_SIDE_STREAMS && ok; ++k) ok = hipStreamCreateWithPriority(&S->side[k], hipStreamNonBlocking, hi) == hipSuccess;
    if (ok) ok = hipDeviceSynchronize() == hipSuccess;       // the zeroed counters, before any stream's first launch
    S->ok = ok;
    if (!ok) (void)hipGetLastError();
    return S;
}
void egg_side_destroy(void* p) {
    EggSide* S = static_cast<EggSide*>(p);
    if (!S) return;
    for (int k = 0; k < EGG_SIDE_RING; ++k) {
        if (S->q[k]) (void)hipFree(S->q[k]);
        if (S->used[k]) (void)hipEventDestroy(S->used[k]);
        if (S->fork[k]) (void)hipEventDestroy(S->fork[k]);
    }
    for (auto& st : S->side) if (st) (void)hipStreamDestroy(st);
    delete S;
}
hipError_t bind_fault_egg(unsigned* word) { return hipMemcpyToSymbol(HIP_SYMBOL(g_egg_fault), &word, sizeof(word)); }

template <bool CULL, int WIT>
static void launch_egg_t(const FrameEgg& F, const RowMap& M, float* out, hipStream_t s, dim3 grid, HotRect hot, size_t pad, EggSide* S) {
    EggQArg A{nullptr, 0u, 0u, 0};
    const int nhot = hot.w * hot.h;
    if (!EGG_COOP || !S || !S->ok || nhot <= 0) {
        hipLaunchKernelGGL((k_egg<CULL, WIT>), grid, dim3(64 * EGG_TX), pad, s, F, M, out, hot, A);
        return;
    }
    if constexpr (EGG_COOP != 0) {                           // (the finisher kernels are compiled only into builds that launch them)
    const int k = (int)(S->next++ % EGG_SIDE_RING);
    hipStream_t side = S->side[k % EGG_SIDE_STREAMS];
    if (++S->seq[k] == 0u) ++S->seq[k];                      // (0 is the tag of a slot never written)
    A.q = S->q[k]; A.seq = S->seq[k]; A.expected = (unsigned)nhot * EGG_TX; A.nf = EGG_COOP_NF;
    if (S->was_used[k]) { (void)hipStreamWaitEvent(s, S->used[k], 0); (void)hipStreamWaitEvent(side, S->used[k], 0); }
    // fork: the finishers start where the caller's stream stands; k_egg FIRST — should both streams share a hardware queue, the
    // finishers then run behind it (late, but they never wait for a kernel that is queued behind them)
    (void)hipEventRecord(S->fork[k], s);
    hipLaunchKernelGGL((k_egg<CULL, WIT>), grid, dim3(64 * EGG_TX), pad, s, F, M, out, hot, A);
#ifdef EGG_DBG_NOFIN
    return;
#endif
    (void)hipStreamWaitEvent(side, S->fork[k], 0);
    hipLaunchKernelGGL((k_egg_finish<CULL, WIT>), dim3(EGG_COOP_NF), dim3(256), 0, side, F, M, out, A);
    (void)hipEventRecord(S->used[k], side);
    (void)hipStreamWaitEvent(s, S->used[k], 0);              // join
    S->was_used[k] = true;
    }
}

dim3 egg_grid(const RowMap& M) { return grid_for<EGG_TW, EGG_TX>(M); }

// side: the context's EggSide, or nullptr (a stream being captured, a launch that must stay one kernel): the plain kernel
void launch_egg(const FrameEgg& F, const RowMap& M, float* out, hipStream_t s, int variant, void* side) {
    const dim3 grid = grid_for<EGG_TW, EGG_TX>(M);
    const HotRect hot = egg_hot_rect(F, M, grid);
    EggSide* S = static_cast<EggSide*>(side);
    static const int pad = []() { const char* e = std::getenv("SBX_DEBUG_LDS_PAD"); return e ? std::atoi(e) : EGG_LDS_PAD; }();
    if (variant == 1) launch_egg_t<false, 0>(F, M, out, s, grid, hot, (size_t)pad, S);
    else if (variant == 2) launch_egg_t<true, 2>(F, M, out, s, grid, hot, (size_t)pad, S);
    else if (variant == 3) launch_egg_t<true, 0>(F, M, out, s, grid, hot, (size_t)pad, S);
    else launch_egg_t<true, EGG_WITNESS>(F, M, out, s, grid, hot, (size_t)pad, S);
}

}  // namespace sbx
