// shaderbox_amd/csrc/sbx_capi.hip — the C ABI of libsbx (include/sbx.h) and the host-side frame setup.
//
// Host side of the drop-in: what the reference's hosts do around mainImage() — own the uniforms
// and the render target, pick the app, issue one draw per frame (util/hlsltoy/src/hlsltoy.cpp:
// 402-426, 494-516) — reduced to: validate, build the per-frame constant block for the app with the
// shared math spec, launch the app's kernel on the caller's stream.  There is NO CPU fallback: without
// a gfx950 device sbx_create fails with SBX_ERR_NO_DEVICE.
#include "../../include/sbx.h"
#include "../../include/sbx_test.h"
#include "sbx_device.h"
#include "sbx_tile_order.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

using namespace sbx;

struct YtabSlot {
    // streams that launched readers of this table (they may still be running), each with the event a rebuild of the slot records on it
    std::vector<std::pair<hipStream_t, hipEvent_t>> users;
};
struct TimingPair { hipEvent_t ev0{}, ev1{}; bool complete = false; };
struct sbx_ctx {
    int device = 0;
    bool timing = false;
    int variant = 0;
    int out_format = 0;                    // sbx_set_output_format: 0 float pixels, 1 R8G8B8A8_UNORM words
    int precision = 0;                     // sbx_set_precision: 0 bit-exact (default), 1 = SBX_PRECISION_1E4 (APP_ATMOSPHERE only)
    int sdf_roots = 0;                     // sbx_set_variant 2 / 3: the SDF kernels' square-root witness test build / IEEE roots
    // APP_CLOUDS y tables: CLOUDS_YTAB_RING slots for eager launches + CLOUDS_YTAB_CAPTURE slots that only launches
    // recorded into a stream capture use (a captured graph bakes the slot pointer in, so eager rebuilds must never
    // touch it, and the build is always part of the graph).
    char* ytab = nullptr;
    unsigned ytab_next = 0, ytab_cap_next = 0;
    // The eager table depends only on (eye.y, wind.y * t, dt, steps): it is rebuilt, into the next ring slot, only when
    // that key changes (default wind has no y component, so an animation reuses one table).  The state below is
    // committed only when a build has actually been ENQUEUED on a stream that is executing (not capturing).
    bool ytab_valid = false;
    float ytab_key[3] = {0, 0, 0};
    int ytab_steps = 0;
    int ytab_slot = 0;
    hipStream_t ytab_stream = nullptr;
    hipEvent_t ytab_ready{};
    bool have_ytab_event = false;
    YtabSlot slots[CLOUDS_YTAB_RING];
    // march lengths beyond the ring's rows (CLOUDS_YTAB_ROWS): ONE table grown on demand (48 B per step), rebuilt when its key
    // changes after waiting for the device — such frames take tens of milliseconds, the wait is noise
    char* ytab_big = nullptr;
    int ytab_big_rows = 0;
    bool ytab_big_valid = false;
    float ytab_big_key[3] = {0, 0, 0};
    int ytab_big_steps = 0;
    hipEvent_t ytab_big_ready{};
    bool have_ytab_big_event = false;
    std::vector<hipEvent_t> event_pool;
    // per-stream timing events (sbx_set_timing): a pair brackets the last launch on its stream
    std::vector<std::pair<hipStream_t, TimingPair>> timers;
    int last_timer = -1;
    // APP_CLOUDS_TEX: the library's R32F copies of the two bound noise volumes (sbx_set_noise_volumes)
    float* noise_tex = nullptr;      // shape (t1)
    float* noise_tex2 = nullptr;     // detail (t2)
    int noise_tex_size = 0, noise_tex2_size = 0;
    unsigned* tex_scan = nullptr;    // 6 device words: min / max keys and NaN flag of the two volumes (sbx_set_noise_volumes)
    float tex_bounds[4] = {0, 0, 0, 0};   // {lo1, hi1, lo2, hi2} of the texels; valid only if tex_bounds_valid
    bool tex_bounds_valid = false;
    // sbx_main_image: the frames of the last MI_FRAMES distinct (app, uniforms, aux) seen, each in pinned host memory behind a
    // sequence lock — host threads that hit read their pixel WITHOUT any lock or shared write (the reference's harness calls
    // mainImage from many threads, src/def.h:7-8); only a miss takes mi_lock and renders.  `gen` is even while an entry is stable
    // and odd while the rendering thread rewrites it; a reader that sees the same even value before and after its reads has read
    // one frame.  The key words are relaxed atomics because readers look at them while a writer may be storing.
    static constexpr int MI_FRAMES = 2;
    static constexpr int MI_KEY_WORDS = 2 + (int)(sizeof(sbx_uniforms) + sizeof(sbx_aux_clouds)) / 4;
    struct MiEntry {
        std::atomic<uint64_t> gen{0};
        std::atomic<float*> host{nullptr};       // pinned (hipHostMalloc): the frame comes back with one asynchronous copy
        std::atomic<uint32_t> key[MI_KEY_WORDS];
        size_t cap_floats = 0;                   // (writer only, under mi_lock)
        uint64_t born = 0;
        bool used = false;
    };
    MiEntry mi[MI_FRAMES];
    uint64_t mi_clock = 0;
    std::vector<float*> mi_retired;              // pinned buffers outgrown by a larger frame: a reader may still be inside one, so
                                                 // they outlive the resize (the oldest goes when eight are waiting; all at destroy)
    // sbx_get_stats; the hit counter is striped over cache lines (it is bumped once per pixel by every host thread)
    struct alignas(64) Stripe { std::atomic<uint64_t> n{0}; };
    Stripe st_hits[16];
    std::atomic<uint64_t> st_launches{0}, st_frames{0}, st_points{0};
    // sbx_main_image_batch / off-centre sbx_main_image: pinned staging of a point list (4 + 2 floats per point), read and written by
    // the kernel where it lies
    float* pt_host = nullptr;
    size_t pt_cap = 0;
    // sbx_render_rows_host: device staging of a host frame, the stream its strips are copied out on, one event per strip
    char* hs_dev = nullptr;
    size_t hs_cap = 0;
    hipStream_t hs_copy = nullptr, hs_render[2] = {nullptr, nullptr};
    bool hs_ready = false;           // every stream and event below exists
    hipEvent_t hs_entry = nullptr;
    hipEvent_t hs_ev[16] = {};
    std::mutex hs_lock;
    std::mutex mi_lock;              // sbx_main_image / sbx_main_image_batch may be called from several host threads
    // span tables of the multi-GPU span exchange (sbx_render_span_*): a few device copies, most recently used first
    struct SpanSlot { std::vector<int> key; std::vector<int> table; int4* dev = nullptr; int max_w = 0; size_t cap = 0; };
    SpanSlot span_slots[4];
    unsigned span_next = 0;
    TileOrderSet tile_orders;              // the dispatch order's tables (sbx_tile_order.h)
    void* egg_side = nullptr;        // kern_egg.hip EggSide: queues, streams and events of APP_EGG's finisher launches
    std::string err;
};

// The sticky fault word of each device (sbx_hashcache.h g_hc_fault): pinned host memory the kernels can write and the host can
// read without synchronising; process-wide, shared by the contexts of a device, never freed.
static unsigned* g_fault_word[64] = {nullptr};
static std::mutex g_fault_lock;
static int bind_fault_word(int device) {
    std::lock_guard<std::mutex> g(g_fault_lock);
    if (device < 0 || device >= 64) return SBX_ERR_ARG;
    if (!g_fault_word[device]) {
        unsigned* w = nullptr;
        if (hipHostMalloc((void**)&w, 64, hipHostMallocMapped) != hipSuccess) return SBX_ERR_HIP;
        *w = 0u;
        g_fault_word[device] = w;
    }
    unsigned* dev = nullptr;
    if (hipHostGetDevicePointer((void**)&dev, g_fault_word[device], 0) != hipSuccess) return SBX_ERR_HIP;
    if (bind_fault_clouds(dev) != hipSuccess || bind_fault_clouds_ue4(dev) != hipSuccess || bind_fault_planet(dev) != hipSuccess ||
        bind_fault_egg(dev) != hipSuccess)
        return SBX_ERR_HIP;
    return SBX_OK;
}

namespace sbx {
unsigned* fault_word_device(int device) {
    unsigned* dev = nullptr;
    if (device < 0 || device >= 64 || !g_fault_word[device]) return nullptr;
    if (hipHostGetDevicePointer((void**)&dev, g_fault_word[device], 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return dev;
}
int ctx_device(const sbx_ctx* ctx) { return ctx->device; }
}  // namespace sbx

static int fail(sbx_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess) {
    if (ctx) {
        ctx->err = what;
        if (e != hipSuccess) { ctx->err += ": "; ctx->err += hipGetErrorString(e); }
    }
    return code;
}

namespace sbx { int ctx_fail(sbx_ctx* ctx, int code, const char* what, hipError_t e) { return fail(ctx, code, what, e); } }

// ---------------------------------------------------------------------------------------------
// frame builders: the frame-constant part of setup_camera()/setup_scene()/sdf() per app
// ---------------------------------------------------------------------------------------------
static FrameClouds build_clouds(const sbx_uniforms& U, const sbx_aux_clouds& A, bool sky_sphere = false) {
    FrameClouds F;
    // setup_camera app_clouds.h:23-30
    const v3 eye = V3(0, -.5f, 0);
    const float angle = U.u_mouse[0] * .5f;
    const v3 look_at = mul(rotate_around_y(angle), V3(0, 0, -1));
    F.cam = make_camera(U.u_res[0], U.u_res[1], 1.f, eye, look_at);   // FOV 1. :219
    F.sun_dir = V3(A.sun_dir[0], A.sun_dir[1], A.sun_dir[2]);
    F.sun_color = V3(A.sun_color[0], A.sun_color[1], A.sun_color[2]);
    const v3 wind = V3(A.wind_dir[0], A.wind_dir[1], A.wind_dir[2]);
    F.wind_off = wind * U.u_time * (1.f / .001f);                      // :167
    F.sun_power = A.sun_power;
    F.sigma = A.sigma_scattering;
    F.steps = A.cld_march_steps;
    F.lsteps = A.illum_march_steps;
    F.dt = A.cld_thick / (float)A.cld_march_steps;                     // :98,180
    F.cov = 1.f - A.cld_coverage;                                      // :83
    F.cov_hi = F.cov + .0135f;                                         // :84
    F.cov_rd = recip64(F.cov_hi - F.cov);
    F.cov_d = F.cov_hi - F.cov;
    F.cov_r = 1.0f / F.cov_d;
    F.lip_ok = 0;                                                      // decided per launch (launch_clouds)
    F.exp_small = 0;
    F.thr1 = F.thr2 = 0.f;                                             // set per launch (launch_clouds)
    // SKY_SPHERE (:8,14-19,154-162)
    F.sky = sky_sphere ? 1 : 0;
    F.atm_y = A.atm_ground_y;
    F.atm_r = A.atm_radius;
    F.nf = sky_sphere ? ((1.f / A.atm_radius) * 10.f) : .001f;        // cld_noise_factor :18 / :20
    F.sky_rot = rotate_around_x(U.u_time);                            // :160
    return F;
}

static BezierFrame bezier_frame(v3 a, v3 b, v3 c) {                    // sdf.h:147-153
    BezierFrame B;
    B.b = b;
    B.w = normalize(cross(c - b, a - b));
    B.u = normalize(c - b);
    B.v = normalize(cross(B.w, B.u));
    B.a2 = V2(dot(a - b, B.u), dot(a - b, B.v));
    B.c2 = V2(dot(c - b, B.u), dot(c - b, B.v));
    B.bc = (a + b + c) * (1.f / 3.f);              // any point works; the radius below is measured from it
    B.br = fmax_(fmax_(length(a - B.bc), length(b - B.bc)), length(c - B.bc)) * 1.001f + 1e-4f;
    return B;
}
static CylFrame cyl_frame(v3 P0, v3 P1) {                              // sdf.h:104,106-107
    CylFrame C;
    C.dir = normalize(P1 - P0);
    C.len1 = length(P1);
    C.len0 = length(P0);
    return C;
}
static v3 ik_solver(v3 start, v3 goal_abs, float L1, float L2) {       // IK.h:5-52
    const v3 goal = goal_abs - start;
    const float G = length(goal);
    const float cos_theta = (L1 * L1 + G * G - L2 * L2) / (2.f * L1 * G);
    const float sin_theta = sqrt_(1.f - cos_theta * cos_theta);
    const m3 rot = M3(cos_theta, -sin_theta, 0, sin_theta, cos_theta, 0, 0, 0, 1.f);
    return start + mul(rot, normalize(goal) * L1);
}
static FrameEgg build_egg(const sbx_uniforms& U) {
    FrameEgg F;
    F.cam = make_camera(U.u_res[0], U.u_res[1], 1.f, V3(.0f, .25f, 5.25f), V3(.0f, .25f, .0f));   // app_egg.h:23-27,253
    const float t = U.u_time;
    F.rot_y = rotate_around_y(t * -100.0f);                            // :40
    const v3 wheel_pos = V3(0, 1.2f, 0);
    const float pedal_radius = 0.3f, pedal_speed = 400.f, pedal_off = 0.2f;
    const m3 rot_z = rotate_around_z(-t * pedal_speed);                // :73,76
    F.left_foot = wheel_pos + mul(rot_z, V3(0.f, pedal_radius, pedal_off));
    F.right_foot = wheel_pos + mul(rot_z, V3(0.f, -pedal_radius, -pedal_off));
    const v3 side = V3(0, 0, pedal_off);
    const float femur = 0.8f, tibia = 0.75f;
    const v3 zero = V3(0.f, 0.f, 0.f);
    const v3 knee_l = ik_solver(zero + side, F.left_foot, femur, tibia);   // :84-85
    const v3 knee_r = ik_solver(zero - side, F.right_foot, femur, tibia);  // :95-96
    F.leg_l = bezier_frame(-(zero + side), -knee_l, -F.left_foot);         // :111-113
    F.leg_r = bezier_frame(-(zero - side), -knee_r, -F.right_foot);        // :114-116
    const v3 left_toe = normalize(V3(F.left_foot.y - knee_l.y, knee_l.x - F.left_foot.x, 0));     // :120
    const v3 right_toe = normalize(V3(F.right_foot.y - knee_r.y, knee_r.x - F.right_foot.x, 0));  // :125
    F.foot_l = cyl_frame(zero, left_toe / 8.f);
    F.foot_r = cyl_frame(zero, right_toe / 8.f);
    // Bounding sphere of the egg, legs, feet and wheel (everything but the ground), see kern_egg.hip egg_far: each
    // member m has a centre c and a radius rho such that its sdf value is >= .7 * (|p - c| - rho) wherever that is
    // positive: egg spheres r + .36 (two smooth-mins of k = .5 lower the union by <= .25), tubes br + .061 (the .85
    // factor and the thickness), toe cylinders .161 around their midpoint (max(axis, slabs) >= |.|/sqrt2 - 1/16),
    // wheel 1.03.
    F.foot_ml = -F.left_foot + left_toe * (-1.f / 16.f);
    F.foot_mr = -F.right_foot + right_toe * (-1.f / 16.f);
    const float egg_y = 0.65f;
    const v3 cs[8] = {V3(0, egg_y, 0), V3(0, egg_y - 0.45f, 0), V3(0, egg_y + 0.45f, 0), F.leg_l.bc, F.leg_r.bc,
                      -F.left_foot + left_toe * (-1.f / 16.f), -F.right_foot + right_toe * (-1.f / 16.f), -wheel_pos};
    const float rs[8] = {.475f + .36f, .25f + .36f, .25f + .36f, F.leg_l.br + .061f, F.leg_r.br + .061f, .161f, .161f, 1.03f};
    v3 c = V3(0, 0, 0);
    for (int i = 0; i < 8; ++i) c = c + cs[i] * .125f;
    float R = 0.f;
    for (int i = 0; i < 8; ++i) R = fmax_(R, length(cs[i] - c) + rs[i]);
    F.oc = c;
    F.orad = R * 1.001f + 1e-3f;
    F.ocw = mul(transpose(F.rot_y), c + V3(0, 0.5f, 3.5f));        // p = rot_y P - (0, .5, 3.5)  <=>  P = rot_y^T (p + (0, .5, 3.5))
    return F;
}

static FrameRaytracer build_raytracer(const sbx_uniforms& U) {
    FrameRaytracer F;
    const float cb = 2.f;                                              // cb_plane_dist cornell_box.h:62
    // setup_camera app_raytracer.h:38-44
    v2 mouse = V2(0, 0);
    if (!(U.u_mouse[0] < 1e-4f)) mouse = V2(2.f * (U.u_res[0] / U.u_mouse[0]) - 1.f, 2.f * (U.u_res[1] / U.u_mouse[1]) - 1.f);
    const m3 rot_y = rotate_around_y(mouse.x * 30.f);
    const v3 eye = mul(rot_y, V3(0, cb, 2.333f * cb));
    F.cam = make_camera(U.u_res[0], U.u_res[1], tan_(radians_(30.f)), eye, V3(0, cb, 0));   // FOV :138
    // materials: zero-initialised slots (App. B5), mat_debug :20-25, cornell box cornell_box.h:47-55
    for (int i = 0; i < 8; ++i) F.mats[i] = RtMaterial{V3(0, 0, 0), 0.f, 0.f, 0.f, 0.f};
    F.mats[0] = RtMaterial{V3(1.f, 1.f, 1.f), 0.f, 1.f, 0.f, 0.f};
    F.mats[1] = RtMaterial{V3(0.7913f, 0.7913f, 0.7913f), .5f, 1.f, 0.f, 0.f};
    F.mats[2] = RtMaterial{V3(0.6795f, 0.0612f, 0.0529f), .5f, 1.f, 0.f, 0.f};
    F.mats[3] = RtMaterial{V3(0.1878f, 0.1274f, 0.4287f), .5f, 1.f, 0.f, 0.f};
    F.mats[4] = RtMaterial{V3(0.95f, 0.64f, 0.54f), .1f, 1.f, 1.f, 0.f};
    F.mats[5] = RtMaterial{V3(1.f, 0.77f, 0.345f), .05f, 1.333f, 1.f, 0.f};
    for (int i = 0; i < 8; ++i) {                                      // util_optics.h:10-11 with n1 = 1, per material
        const float Rn = (1.f - F.mats[i].ior) / (1.f + F.mats[i].ior);
        F.mats[i].r0 = Rn * Rn;
    }
    // planes, in array-index order ground, behind, front, ceiling, left, right  cornell_box.h:57-69
    F.planes[0] = RtPlane{V3(0, -1, 0), 0.f, 1};
    F.planes[1] = RtPlane{V3(0, 0, -1), -cb, 1};
    F.planes[2] = RtPlane{V3(0, 0, 1), cb, 1};
    F.planes[3] = RtPlane{V3(0, 1, 0), 2.f * cb, 1};
    F.planes[4] = RtPlane{V3(1, 0, 0), cb, 2};
    F.planes[5] = RtPlane{V3(-1, 0, 0), -cb, 3};
    // spheres cornell_box.h:71-82 + animation app_raytracer.h:29-34
    const float s = sin_(U.u_time), c = cos_(U.u_time);
    F.spheres[0] = RtSphere{V3(0, 2.5f * cb + 0.4f, 0), 1.5f, 0, recip64(1.5f)};
    F.spheres[1] = RtSphere{V3(0.75f, 1, -0.75f) + V3(0, abs_(s), c + 1.f), 0.75f, 4, recip64(0.75f)};
    F.spheres[2] = RtSphere{V3(-0.75f, 0.75f, 0.f), 0.75f, 5, recip64(0.75f)};
    F.light = V3(0, 2.f * cb - 0.2f, 1.5f);
    return F;
}

static FrameAtmosphere build_atmosphere(const sbx_uniforms& U) {
    FrameAtmosphere F;
    F.cam = make_camera(U.u_res[0], U.u_res[1], 1.f, V3(0, 0, 0), V3(0, 1, 0));   // app_atmosphere.h:164-175,230
    const m3 rot = rotate_around_x(-abs_(sin_(U.u_time / 2.f)) * 90.f);             // :179
    F.sun_dir = mul(V3(0, 1, 0), rot);                                              // :180 (v * M)
    return F;
}

static FrameSdfAo build_sdf_ao(const sbx_uniforms& U, const sbx_aux_sdf_ao& A) {
    FrameSdfAo F;
    const m3 rot = rotate_around_y(U.u_time * 50.f);                   // app_sdf_ao.h:45-50
    F.cam = make_camera(U.u_res[0], U.u_res[1], 1.f, mul(rot, V3(0, 3, 5)), V3(0, 0, 0));
    F.rx_m90 = rotate_around_x(-90.f);
    F.ry_180 = rotate_around_y(180.f);
    F.sun_dir = normalize(V3(1, 2, 1));
    F.fog_density = A.fog_density;
    F.fog_falloff = A.fog_falloff;
    return F;
}

static Capsule capsule(v3 a, v3 b) {                                   // sdf.h:168-169
    Capsule c;
    c.a = a;
    c.ab = b - a;
    c.rd = recip64(dot(c.ab, c.ab));
    return c;
}
static FrameVinyl build_vinyl(const sbx_uniforms& U, int steps) {
    FrameVinyl F;
    F.steps = steps;                                                   // :411-416
    const float t = U.u_time;
    F.cam = make_camera(U.u_res[0], U.u_res[1], 1.f, V3(0, 5.75f, 6.75f), V3(0, -2.5f, 0));   // app_vinyl.h:56-64,459
    F.platter_rot = mul(rotate_around_y(t * 200.f), rotate_around_x(sin_(t) * .1f));          // :417,424-426
    F.sun_dir = normalize(V3(-1, 4, -3));
    F.ry30 = rotate_around_y(30.f);
    F.rym30 = rotate_around_y(-30.f);
    F.wobble = rotate_around_x(sin_(t * 3.6758f) * .1f);
    const float R = .1f, H = .8f;
    const v3 base_p = V3(-7, 0, -5);
    const v3 a1 = V3(-6, H, -3), a11 = V3(-4.25f, H, 2), a2 = V3(-4.1f, H, 2.45f), a33 = V3(-3.5f, H, 3), a3 = V3(-2, H, 4);
    F.arm1 = capsule(base_p + V3(-1, H, -2), a1);
    F.arm2 = capsule(a1, a11);
    F.arm3 = capsule(a33, a3);
    F.armb = bezier_frame(a11, a2, a33);
    F.a3 = a3;
    const v3 arm_fwd = normalize(a3 - a33);
    const v3 arm_up = V3(0, 1, 0);
    const v3 arm_right = cross(arm_fwd, arm_up);
    F.arm_xform = m3{arm_fwd, arm_up, arm_right};
    const float clr_r = R * 1.5f;
    F.collar = cyl_frame(V3(0, 0, 0), V3(0, 0, 0) + arm_fwd * .05f);
    F.fl_rot = mul(F.arm_xform, rotate_around_x(45.f));
    F.fl_sub1 = arm_right * clr_r;
    F.fl_sub2 = arm_up * clr_r;
    F.fl_rot2 = rotate_around_x(-45.f);
    F.ctg_rot = rotate_around_z(44.f);
    F.cut_rx10 = rotate_around_x(10.f);
    F.cut_rym5 = rotate_around_y(-5.f);
    F.cut2_rz10 = rotate_around_z(10.f);
    (void)R;
    return F;
}

static FramePlanet build_planet(const sbx_uniforms& U, bool atm_sky = false) {
    FramePlanet F;
    F.atm_sky = atm_sky ? 1 : 0;
    F.atm_sun = build_atmosphere(U).sun_dir;
    F.cam = make_camera(U.u_res[0], U.u_res[1], tan_(radians_(30.f)), V3(0, 0, -2.5f), V3(0, 0, 2));   // app_planet.h:47-58,368
    const m3 rot_y = rotate_around_y(27.f);                            // :307
    F.rot = mul(rotate_around_x(U.u_time * -12.f), rot_y);             // :308
    F.rot_cloud = mul(rotate_around_x(U.u_time * 8.f), rot_y);         // :309
    F.rot_t = transpose(F.rot);                                        // :356
    F.L = mul(F.rot, normalize(V3(1, 1, 0)));                          // :289
    return F;
}

static FrameCloudsBest build_clouds_best(const sbx_uniforms& U) {
    FrameCloudsBest F;
    F.cam = make_camera(U.u_res[0], U.u_res[1], 1.f, V3(0, 1.f, 0), V3(0, 1.6f, -1));   // app_clouds_best.h:635-641,663
    F.sun_dir = normalize(V3(0, 0, -1));                               // :415
    F.wind_z = -U.u_time * .2f;                                        // :414
    const float cld_thick = 90.f;                                      // :412
    F.march_step = cld_thick / float(CB_STEPS);                        // :603
    F.cov = .3125f;                                                    // :411
    F.cov_rd = recip64((F.cov + .035f) - F.cov);                       // smoothstep(cov, cov + .035, dens) :583
    // y of the march: projection.y = dir.y / dir.y = 1, so origin.y = eye.y + 1 * 100 and iter.y = 1 * march_step
    const float origin_y = F.cam.eye.y + 1.0f * 100.f;                 // :611
    const float iter_y = 1.0f * F.march_step;                          // :606
    float pos_y = origin_y;
    for (int i = 0; i < CB_STEPS; ++i) {
        const float height = (pos_y - origin_y) / cld_thick;           // :619-620
        F.row[i].illum = exp_(height) / 1.95f;                         // illuminate_volume :591-597
        float q = (pos_y * .001f + 0.f) * 2.032f;                      // density_func :581-582 (wind.y = 0)
        for (int k = 0; k < 5; ++k) { F.row[i].qy[k] = q; q = q * 2.6434f; }   // fbm: p *= lacunarity
        pos_y = pos_y + iter_y;                                        // :628
    }
    return F;
}

static FrameCloudsUe4 build_clouds_ue4(const sbx_uniforms& U, const sbx_aux_clouds_ue4& A) {
    FrameCloudsUe4 F;
    const v3 eye = V3(0, -.5f, 0);                                      // host mapping: src/app_clouds.h:23-30
    const v3 look_at = mul(rotate_around_y(U.u_mouse[0] * .5f), V3(0, 0, -1));
    F.cam = make_camera(U.u_res[0], U.u_res[1], 1.f, eye, look_at);
    if (A.use_dirs) {
        F.sun_dir = V3(A.sun_dir[0], A.sun_dir[1], A.sun_dir[2]);
        F.wind_dir = V3(A.wind_dir[0], A.wind_dir[1], A.wind_dir[2]);
    } else {
        F.sun_dir = normalize(V3(0, abs_(sin_(U.u_time * .3f)), -1));    // SUN_DIR app_clouds.usf:14
        F.wind_dir = V3(0, 0, -U.u_time * .2f);                          // WIND_DIR :13
    }
    F.march_step = A.thickness / float(UE4_STEPS);                      // :199
    F.absorbtion = A.absorbtion;
    F.cov = 1.f - A.coverage;                                           // :256
    F.cov_rd = recip64((F.cov + A.fuzziness) - F.cov);                  // :175
    F.cov_d = (F.cov + A.fuzziness) - F.cov;
    F.cov_r = 1.0f / F.cov_d;
    for (int i = 0; i < UE4_STEPS; ++i) F.eh[i] = exp_(float(i) / float(UE4_STEPS)) / 1.75f;   // :213,221
    return F;
}

// ---------------------------------------------------------------------------------------------
extern "C" {

void sbx_aux_clouds_ue4_defaults(sbx_aux_clouds_ue4* a) {              // app_clouds.usf:4-7
    if (!a) return;
    std::memset(a, 0, sizeof(*a));
    a->coverage = .50f; a->thickness = 15.f; a->absorbtion = 1.030725f; a->fuzziness = 0.035f;
    a->sun_dir[2] = -1.f; a->use_dirs = 0;
}

void sbx_aux_clouds_defaults(sbx_aux_clouds* a) {                      // uniform_buffer.h:39-55
    if (!a) return;
    std::memset(a, 0, sizeof(*a));
    a->wind_dir[2] = .2f;
    a->sun_dir[2] = -1.f;
    a->sun_color[0] = 1.f; a->sun_color[1] = .7f; a->sun_color[2] = .55f;
    a->sun_power = 8.f;
    a->cld_march_steps = 100;
    a->illum_march_steps = 6;
    a->sigma_scattering = .15f;
    a->cld_coverage = .535f;
    a->cld_thick = 125.f;
    a->atm_radius = 5000.f;
    a->atm_ground_y = 4750.f;
}
void sbx_aux_sdf_ao_defaults(sbx_aux_sdf_ao* a) {                      // uniform_buffer.h:56-60
    if (!a) return;
    std::memset(a, 0, sizeof(*a));
    a->fog_density = .1f;
    a->fog_falloff = .5f;
}

int sbx_create(int device, sbx_ctx** out) {
    if (!out) return SBX_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SBX_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return SBX_ERR_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SBX_ERR_HIP;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SBX_ERR_NO_DEVICE;   // kernels exist for gfx950 only
    sbx_ctx* ctx = new sbx_ctx();
    ctx->device = device;
    ctx->tile_orders.pool = &ctx->event_pool;
    for (auto& en : ctx->mi) for (auto& w : en.key) w.store(0xffffffffu, std::memory_order_relaxed);
    if (hipSetDevice(device) != hipSuccess ||
        hipMalloc((void**)&ctx->ytab, (size_t)(CLOUDS_YTAB_RING + CLOUDS_YTAB_CAPTURE) * CLOUDS_YTAB_BYTES) != hipSuccess ||
        bind_fault_word(device) != SBX_OK) {
        if (ctx->ytab) (void)hipFree(ctx->ytab);
        delete ctx;
        return SBX_ERR_HIP;
    }
    *out = ctx;
    return SBX_OK;
}

void sbx_destroy(sbx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->ytab) (void)hipFree(ctx->ytab);
    if (ctx->ytab_big) (void)hipFree(ctx->ytab_big);
    if (ctx->have_ytab_big_event) (void)hipEventDestroy(ctx->ytab_big_ready);
    for (auto& en : ctx->mi) if (en.host.load()) (void)hipHostFree(en.host.load());
    for (float* h : ctx->mi_retired) (void)hipHostFree(h);
    if (ctx->pt_host) (void)hipHostFree(ctx->pt_host);
    egg_side_destroy(ctx->egg_side);
    tile_order_destroy(ctx->tile_orders);
    if (ctx->hs_dev) (void)hipFree(ctx->hs_dev);
    if (ctx->hs_copy) (void)hipStreamDestroy(ctx->hs_copy);
    for (auto& st : ctx->hs_render) if (st) (void)hipStreamDestroy(st);
    if (ctx->hs_entry) (void)hipEventDestroy(ctx->hs_entry);
    for (auto& e : ctx->hs_ev) if (e) (void)hipEventDestroy(e);
    for (auto& sl : ctx->span_slots) if (sl.dev) (void)hipFree(sl.dev);
    if (ctx->noise_tex) (void)hipFree(ctx->noise_tex);
    if (ctx->noise_tex2) (void)hipFree(ctx->noise_tex2);
    if (ctx->tex_scan) (void)hipFree(ctx->tex_scan);
    if (ctx->have_ytab_event) (void)hipEventDestroy(ctx->ytab_ready);
    for (auto& sl : ctx->slots) for (auto& u : sl.users) (void)hipEventDestroy(u.second);
    for (auto& e : ctx->event_pool) (void)hipEventDestroy(e);
    for (auto& t : ctx->timers) { (void)hipEventDestroy(t.second.ev0); (void)hipEventDestroy(t.second.ev1); }
    delete ctx;
}

static bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

// APP_CLOUDS launch with the y-table bookkeeping.  Three cases:
//  (1) no table (per-lane variant, or a step count the table does not cover): nothing cached, nothing touched;
//  (2) the stream is being captured: the build goes into the capture, into a slot of the capture ring, together with
//      the render kernel; cache key, events and the eager ring are left alone (nothing has executed yet);
//  (3) eager: rebuild into the next ring slot only when the key changed — after waiting for every launch that may
//      still be reading that slot — and record, per stream, an event behind each consumer of the current slot.
static int render_clouds(sbx_ctx* ctx, const FrameClouds& F, const RowMap& M, float* rgba, hipStream_t s, bool capturing) {
    const bool uses_table = ctx->variant == 0 && F.steps > 0 && F.steps <= CLOUDS_YTAB_ROWS;
    if (!uses_table && ctx->variant == 0 && F.steps > CLOUDS_YTAB_ROWS && F.steps <= CLOUDS_YTAB_BIG_MAX && !capturing) {
        // (4) a march longer than the ring's tables: the context's one big table (round 3 fell back to the table-less kernels
        //     here, ~2x slower per step)
        const float key[3] = {F.cam.eye.y, F.wind_off.y, F.dt};
        hipError_t e;
        if (!ctx->have_ytab_big_event) {
            if ((e = hipEventCreateWithFlags(&ctx->ytab_big_ready, hipEventDisableTiming)) != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipEventCreate", e);
            ctx->have_ytab_big_event = true;
        }
        const bool rebuild = !ctx->ytab_big_valid || ctx->ytab_big_steps != F.steps || std::memcmp(key, ctx->ytab_big_key, sizeof(key)) != 0;
        if (rebuild) {
            ctx->ytab_big_valid = false;
            if ((e = hipDeviceSynchronize()) != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipDeviceSynchronize", e);   // readers of the old table
            if (F.steps > ctx->ytab_big_rows) {
                if (ctx->ytab_big) (void)hipFree(ctx->ytab_big);
                ctx->ytab_big = nullptr; ctx->ytab_big_rows = 0;
                const int rows = (F.steps + 4095) / 4096 * 4096;
                if ((e = hipMalloc((void**)&ctx->ytab_big, (size_t)rows * 48)) != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipMalloc", e);
                ctx->ytab_big_rows = rows;
            }
        } else {
            (void)hipStreamWaitEvent(s, ctx->ytab_big_ready, 0);
        }
        launch_clouds(F, M, rgba, s, 0, ctx->ytab_big, ctx->ytab_big_rows, rebuild);
        if (rebuild) {
            (void)hipEventRecord(ctx->ytab_big_ready, s);
            std::memcpy(ctx->ytab_big_key, key, sizeof(key));
            ctx->ytab_big_steps = F.steps;
            ctx->ytab_big_valid = true;
        }
        return SBX_OK;
    }
    if (!uses_table) {
        launch_clouds(F, M, rgba, s, ctx->variant, nullptr, 0, false);
        return SBX_OK;
    }
    if (capturing) {
        char* tab = ctx->ytab + (size_t)(CLOUDS_YTAB_RING + (ctx->ytab_cap_next++ % CLOUDS_YTAB_CAPTURE)) * CLOUDS_YTAB_BYTES;
        launch_clouds(F, M, rgba, s, 0, tab, CLOUDS_YTAB_ROWS, true);
        return SBX_OK;
    }
    const float key[3] = {F.cam.eye.y, F.wind_off.y, F.dt};
    const bool rebuild = !ctx->ytab_valid || F.steps != ctx->ytab_steps || std::memcmp(key, ctx->ytab_key, sizeof(key)) != 0;
    if (!ctx->have_ytab_event) {
        if (hipEventCreateWithFlags(&ctx->ytab_ready, hipEventDisableTiming) != hipSuccess)
            return fail(ctx, SBX_ERR_HIP, "hipEventCreate");
        ctx->have_ytab_event = true;
    }
    int slot = ctx->ytab_slot;
    if (rebuild) {
        slot = (int)(ctx->ytab_next++ % CLOUDS_YTAB_RING);
        YtabSlot& sl = ctx->slots[slot];
        // Earlier readers of the slot we are about to overwrite: every stream that launched a reader of it.  The event is recorded
        // NOW, on the reader's stream — behind everything that stream holds, its readers included — and this stream waits for it.
        // (Until round 6 every launch re-recorded its stream's event right after the kernel: a barrier packet per frame, 26 us
        // between two back-to-back 2.4 ms launches of one stream — profiles/r06_streams3_trace.txt — to protect a rebuild that an
        // animation with the default wind never does.)
        for (auto& u : sl.users) {
            if (u.first != s) {
                const bool ok = !stream_is_capturing(u.first) && hipEventRecord(u.second, u.first) == hipSuccess &&
                                hipStreamWaitEvent(s, u.second, 0) == hipSuccess;
                if (!ok) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); (void)hipGetLastError(); }   // (a stream that is gone)
            }
            ctx->event_pool.push_back(u.second);
        }
        sl.users.clear();
    } else if (s != ctx->ytab_stream) {
        (void)hipStreamWaitEvent(s, ctx->ytab_ready, 0);           // table was built on another stream
    }
    char* tab = ctx->ytab + (size_t)slot * CLOUDS_YTAB_BYTES;
    launch_clouds(F, M, rgba, s, 0, tab, CLOUDS_YTAB_ROWS, rebuild);
    if (rebuild) {
        (void)hipEventRecord(ctx->ytab_ready, s);                  // the build is enqueued: now the cache state is true
        std::memcpy(ctx->ytab_key, key, sizeof(key));
        ctx->ytab_steps = F.steps;
        ctx->ytab_slot = slot;
        ctx->ytab_stream = s;
        ctx->ytab_valid = true;
    }
    YtabSlot& sl = ctx->slots[slot];                               // this stream reads the slot: remembered, nothing recorded
    bool found = false;
    for (auto& u : sl.users) if (u.first == s) { found = true; break; }
    if (!found) {
        hipEvent_t ev{};
        if (!ctx->event_pool.empty()) { ev = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
        else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipEventCreate");
        sl.users.emplace_back(s, ev);
    }
    return SBX_OK;
}

static const char* kFaultText = "an earlier launch on this device hit the bound of the hash cache's miss loop (sbx_hashcache.h): its "
                                "pixels are invalid; re-render after sbx_clear_fault";
static const char* kFaultTextWait = "a wait of the store exchange (sbx_shared_frame_begin / _end) on this device gave up: a rank's signal did not "
                                    "arrive in time, the frame is incomplete; re-render after sbx_clear_fault";
static unsigned device_fault_code(const sbx_ctx* ctx) {
    const unsigned* w = (ctx->device >= 0 && ctx->device < 64) ? g_fault_word[ctx->device] : nullptr;
    return w ? *(volatile const unsigned*)w : 0u;
}
static bool device_fault(const sbx_ctx* ctx) { return device_fault_code(ctx) != 0u; }
static const char* kFaultTextEgg = "a finisher of an APP_EGG launch on this device gave up waiting for the launch's own waves (kern_egg.hip "
                                   "k_egg_finish): the frame is incomplete; re-render after sbx_clear_fault";
static const char* fault_text(const sbx_ctx* ctx) {
    const unsigned c = device_fault_code(ctx);
    return c == 2u ? kFaultTextWait : c == 3u ? kFaultTextEgg : kFaultText;
}

// Domain of the margin-based culls of EGG / SDF_AO / VINYL (bounding spheres and boxes around members placed by rotations) and
// of PLANET's |o|^2 band test (a rotation preserves the norm): their proofs take the frame's rotations to BE rotations.  The
// spec's sin / cos reduce their argument with a two-term pi, which is accurate while |angle| < 2^30 or so; the fastest angle
// of any app is 400 degrees = 7 rad per second of u_time (src/app_egg.h:73), so |u_time| <= 1e8 keeps every matrix entry the
// correctly rounded sin / cos of its angle (orthonormal to 2e-7; the margins are >= 3e-4 relative).  Beyond that, and for
// NaN / inf, the reduction returns what it returns (the oracle's does the same) and the PLAIN kernels run: same operations
// as the reference on whatever the matrices are.  (APP_CLOUDS has its own per-launch checks: clouds_regular,
// clouds_lip_domain in kern_clouds.hip.)
static bool tame_time(float t) { return std::fabs(t) <= 1e8f; }      // NaN compares false

static int render_mapped(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, const RowMap& M_in,
                         float* rgba, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    if (M_in.nrows == 0) return SBX_OK;
    if (device_fault(ctx)) return fail(ctx, SBX_ERR_FAULT, fault_text(ctx));
    // argument checks come before anything is enqueued or recorded
    if (app < SBX_APP_PLANET || app > SBX_APP_PLANET_ATMOSPHERE) return fail(ctx, SBX_ERR_UNSUPPORTED, "app is not on the accelerated path");
    sbx_aux_clouds AC;
    if (app == SBX_APP_CLOUDS || app == SBX_APP_CLOUDS_TEX || app == SBX_APP_CLOUDS_SKY) {
        if (aux) AC = *(const sbx_aux_clouds*)aux; else sbx_aux_clouds_defaults(&AC);
        if (AC.cld_march_steps < 0 || AC.illum_march_steps < 0) return fail(ctx, SBX_ERR_ARG, "negative march steps");
    }
    if (app == SBX_APP_CLOUDS_TEX && !ctx->noise_tex) return fail(ctx, SBX_ERR_ARG, "APP_CLOUDS with USE_NOISE_TEX needs sbx_set_noise_volume first");
    const bool capturing = stream_is_capturing(s);
    TimingPair* tp = nullptr;
    if (ctx->timing && !capturing) {                               // events recorded into a capture would time nothing
        int idx = -1;
        for (size_t i = 0; i < ctx->timers.size(); ++i) if (ctx->timers[i].first == s) idx = (int)i;
        if (idx < 0) {
            TimingPair p;
            if ((e = hipEventCreate(&p.ev0)) != hipSuccess || (e = hipEventCreate(&p.ev1)) != hipSuccess)
                return fail(ctx, SBX_ERR_HIP, "hipEventCreate", e);
            ctx->timers.emplace_back(s, p);
            idx = (int)ctx->timers.size() - 1;
        }
        tp = &ctx->timers[idx].second;
        tp->complete = false;
        ctx->last_timer = idx;
        (void)hipEventRecord(tp->ev0, s);
    }
    int rc = SBX_OK;
    ctx->st_launches.fetch_add(1, std::memory_order_relaxed);
    const int cull_variant = tame_time(uni->u_time) ? ctx->variant : 1;
    const int sdf_variant = cull_variant == 1 ? 1 : ctx->sdf_roots;      // EGG / SDF_AO / VINYL: 2 / 3 = the witness's test build / IEEE roots
    // The dispatch order of this launch (TileOrder): the tiles by the cost earlier frames of this app and shape measured, longest
    // first.  For the kernels it was measured to help (profiles/r06_tile_order.txt: CLOUDS 4K 2.41 -> 2.23 ms, CLOUDS_SKY 5.14 ->
    // 4.91, VINYL 1.25 -> 1.19, VINYL_GPU 1.24 -> 1.14, and APP_EGG 1920x1080 0.205 -> 0.134 ms back to back — its launch drags a tail
    // of ~350 silhouette waves of up to 175 us behind it, some of which the hot-first order starts 40 us into the launch; under the
    // table they are the first to start); within +-0.6 % for ATMOSPHERE, PLANET, RAYTRACER, SDF_AO, +1.6 % for CLOUDS_BEST: those
    // keep their own order.  (Until the table is there, and with frames in flight, k_egg's hot-first order stands.)
    dim3 og(0, 0, 1);
    switch (app) {
    case SBX_APP_CLOUDS: case SBX_APP_CLOUDS_SKY: og = ctx->variant == 1 ? og : clouds_grid(M_in); break;
    case SBX_APP_VINYL: case SBX_APP_VINYL_GPU: og = vinyl_grid(M_in); break;
    case SBX_APP_EGG: og = egg_grid(M_in); break;
    default: break;
    }
    RowMap M = M_in;
    unsigned long long scene = 1469598103934665603ull;           // FNV-1a of the frame's uniforms (and APP_CLOUDS' aux block): "the same scene"
    if (og.x) {
        auto mix = [&scene](const void* p, size_t n) { for (size_t i = 0; i < n; ++i) scene = (scene ^ ((const unsigned char*)p)[i]) * 1099511628211ull; };
        mix(uni, sizeof(*uni));
        if (app == SBX_APP_CLOUDS || app == SBX_APP_CLOUDS_SKY) mix(&AC, sizeof(AC));
    }
    const int scene_policy = app == SBX_APP_EGG ? TILE_SCENE_KEYED : (app == SBX_APP_CLOUDS || app == SBX_APP_CLOUDS_SKY) ? TILE_SCENE_REFRESH : TILE_SCENE_FREE;
    TileOrder* const ordered = tile_order_begin(ctx->tile_orders, app, M, og, s, capturing, scene, scene_policy);
    switch (app) {
    case SBX_APP_CLOUDS: rc = render_clouds(ctx, build_clouds(*uni, AC), M, rgba, s, capturing); break;
    case SBX_APP_CLOUDS_SKY: launch_clouds(build_clouds(*uni, AC, true), M, rgba, s, ctx->variant, nullptr, 0, false); break;
    case SBX_APP_CLOUDS_TEX: launch_clouds_tex(build_clouds(*uni, AC), M, rgba, s, ctx->noise_tex, ctx->noise_tex_size, ctx->noise_tex2, ctx->noise_tex2_size,
                                                    // a captured launch is replayed later, possibly over volumes re-bound in place with
                                                    // other texel ranges: no bounds baked into a graph (the plain exp_ / IEEE divide)
                                                    (ctx->tex_bounds_valid && !capturing) ? ctx->tex_bounds : nullptr); break;
    case SBX_APP_EGG:
        // the finishers' queues and streams, on first use; a stream being captured gets the plain kernel (a graph would bake a queue
        // and its sequence number in)
        if (!ctx->egg_side && !capturing) ctx->egg_side = egg_side_create();
        launch_egg(build_egg(*uni), M, rgba, s, sdf_variant, capturing ? nullptr : ctx->egg_side);
        break;
    case SBX_APP_RAYTRACER: launch_raytracer(build_raytracer(*uni), M, rgba, s, ctx->variant == 1 ? 1 : ctx->sdf_roots); break;
    case SBX_APP_ATMOSPHERE: launch_atmosphere(build_atmosphere(*uni), M, rgba, s, ctx->precision); break;
    case SBX_APP_SDF_AO: {
        sbx_aux_sdf_ao A;
        if (aux) A = *(const sbx_aux_sdf_ao*)aux; else sbx_aux_sdf_ao_defaults(&A);
        launch_sdf_ao(build_sdf_ao(*uni, A), M, rgba, s, sdf_variant);
        break;
    }
    case SBX_APP_PLANET: launch_planet(build_planet(*uni), M, rgba, s, cull_variant); break;
    case SBX_APP_PLANET_ATMOSPHERE: launch_planet(build_planet(*uni, true), M, rgba, s, cull_variant); break;
    case SBX_APP_VINYL: launch_vinyl(build_vinyl(*uni, 60), M, rgba, s, sdf_variant); break;
    case SBX_APP_VINYL_GPU: launch_vinyl(build_vinyl(*uni, 180), M, rgba, s, sdf_variant); break;
    case SBX_APP_CLOUDS_BEST: launch_clouds_best(build_clouds_best(*uni), M, rgba, s); break;
    case SBX_APP_CLOUDS_UE4: {
        sbx_aux_clouds_ue4 A;
        if (aux) A = *(const sbx_aux_clouds_ue4*)aux; else sbx_aux_clouds_ue4_defaults(&A);
        launch_clouds_ue4(build_clouds_ue4(*uni, A), M, rgba, s);
        break;
    }
    default: break;
    }
    if (tp) { (void)hipEventRecord(tp->ev1, s); tp->complete = true; }
    if (ordered) tile_order_end(ctx->tile_orders, ordered, s);           // (a table due for its refresh is rebuilt behind the launch)
    if (rc != SBX_OK) return rc;
    e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "kernel launch", e);
    return SBX_OK;
}

static int check_common(sbx_ctx* ctx, const sbx_uniforms* uni, const float* rgba, int& W, int& H, unsigned align_mask = 15u) {
    if (!ctx) return SBX_ERR_ARG;
    if (!uni || !rgba) return fail(ctx, SBX_ERR_ARG, "NULL uniforms or framebuffer");
    W = (int)uni->u_res[0]; H = (int)uni->u_res[1];
    if (W <= 0 || H <= 0 || (float)W != uni->u_res[0] || (float)H != uni->u_res[1] || W > 65536 || H > 65536)
        return fail(ctx, SBX_ERR_ARG, "u_res must be positive integers <= 65536");
    // float4 pixels are stored with one 16-byte store; 3-channel slabs (RowMap.rgb) with three 4-byte stores: a slab piece that
    // starts at row r0 of an odd-width frame is only 4-byte aligned, and that is fine
    if (ctx->out_format) align_mask = 3u;                     // one 4-byte store per pixel
    if (((uintptr_t)rgba & align_mask) != 0) return fail(ctx, SBX_ERR_ARG, align_mask == 15u ? "framebuffer must be 16-byte aligned" : "slab must be 4-byte aligned");
    return SBX_OK;
}

// what RowMap.rgb says about the output of a frame-granular entry point: the context's format wins (sbx_set_output_format)
static int out_rgb(const sbx_ctx* ctx, int rgb) { return ctx->out_format ? 2 : rgb; }

static int render_rows(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int y0, int y1, float* rgba,
                       void* stream, bool float_pixels) {
    int W, H;
    if (ctx && uni && y0 == y1 && y0 >= 0 && (float)y0 <= uni->u_res[1]) return SBX_OK;   // empty strip: nothing to write
    int rc = check_common(ctx, uni, rgba, W, H);
    if (rc != SBX_OK) return rc;
    if (y0 < 0 || y1 < y0 || y1 > H) return fail(ctx, SBX_ERR_ARG, "bad row range");
    RowMap M{W, H, y0, (y1 - y0) > 0 ? (y1 - y0) : 1, 1, 0, y1 - y0, 0, 1, 1, 0, float_pixels ? 0 : out_rgb(ctx, 0)};
    return render_mapped(ctx, app, uni, aux, M, rgba, stream);
}
int sbx_render_rows(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int y0, int y1, float* rgba,
                    void* stream) {
    return render_rows(ctx, app, uni, aux, y0, y1, rgba, stream, false);
}

// Rows [y0, y1) into HOST memory: rendered strip by strip into the context's staging buffer, every strip copied out on the context's
// copy stream while the next ones render (a frame's copy costs about what its kernel does: CLOUDS 4K 2.3 ms of copy at 57 GB/s
// beside 2.4 ms of kernel, profiles/r05_host_boundary.txt).
int sbx_render_rows_host(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int y0, int y1, void* rgba_host,
                         void* stream) {
    int W, H;
    if (ctx && uni && y0 == y1 && y0 >= 0 && (float)y0 <= uni->u_res[1]) return SBX_OK;   // empty strip: nothing to write
    int rc = check_common(ctx, uni, reinterpret_cast<float*>(rgba_host), W, H, 3u);      // (a host frame needs no 16-byte alignment)
    if (rc != SBX_OK) return rc;
    if (y0 < 0 || y1 < y0 || y1 > H) return fail(ctx, SBX_ERR_ARG, "bad row range");
    if (stream_is_capturing((hipStream_t)stream)) return fail(ctx, SBX_ERR_ARG, "sbx_render_rows_host cannot be captured (it waits for its copies)");
    std::lock_guard<std::mutex> lock(ctx->hs_lock);
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    const size_t row_bytes = (size_t)W * (ctx->out_format ? 4 : 16);
    const size_t need = row_bytes * (size_t)(y1 - y0);
    if (need > ctx->hs_cap) {
        if (ctx->hs_dev) { (void)hipDeviceSynchronize(); (void)hipFree(ctx->hs_dev); }
        ctx->hs_dev = nullptr; ctx->hs_cap = 0;
        if ((e = hipMalloc((void**)&ctx->hs_dev, need)) != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipMalloc (host-frame staging)", e);
        ctx->hs_cap = need;
    }
    if (!ctx->hs_ready) {
        // all of the streams and events, or none: a set-up that stopped half way must not leave the next call a null stream
        // (the old test was `!hs_copy`, which is set first)
        auto undo = [&]() {
            if (ctx->hs_copy) { (void)hipStreamDestroy(ctx->hs_copy); ctx->hs_copy = nullptr; }
            for (auto& st : ctx->hs_render) if (st) { (void)hipStreamDestroy(st); st = nullptr; }
            if (ctx->hs_entry) { (void)hipEventDestroy(ctx->hs_entry); ctx->hs_entry = nullptr; }
            for (auto& ev : ctx->hs_ev) if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
        };
        undo();
        if ((e = hipStreamCreateWithFlags(&ctx->hs_copy, hipStreamNonBlocking)) != hipSuccess ||
            (e = hipStreamCreateWithFlags(&ctx->hs_render[0], hipStreamNonBlocking)) != hipSuccess ||
            (e = hipStreamCreateWithFlags(&ctx->hs_render[1], hipStreamNonBlocking)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->hs_entry, hipEventDisableTiming)) != hipSuccess) {
            undo();
            return fail(ctx, SBX_ERR_HIP, "hipStreamCreate", e);
        }
        for (auto& ev : ctx->hs_ev)
            if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) { undo(); return fail(ctx, SBX_ERR_HIP, "hipEventCreate", e); }
        ctx->hs_ready = true;
    }
    // Pinned (or registered) memory: the copies are asynchronous, so the frame goes out in strips — whole 64-row bands (a multiple of
    // every kernel's tile height), sixteen at most — rendered alternately on two streams of the context (one strip's tail overlaps the
    // next one's ramp; one stream and eight strips: CLOUDS 4K 4.3 ms instead of 3.55, SBX_HOST_STRIPS sets the number for measurements) behind whatever `stream` holds, each copied out as soon as it is there.  Pageable memory: the runtime stages
    // such a copy synchronously, nothing overlaps, and one strip is fastest (CLOUDS 4K 5.3 ms against 6.6 with eight).
    hipPointerAttribute_t at{};
    const bool pinned = hipPointerGetAttributes(&at, rgba_host) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();                                      // (an unregistered pointer is reported as an error: it is the pageable case)
    const int rows = y1 - y0;
    static const int want = [] { const char* v = getenv("SBX_HOST_STRIPS"); const int n = v ? atoi(v) : 16; return n < 1 ? 1 : (n > 16 ? 16 : n); }();
    const int strips = !pinned ? 1 : (rows >= 64 * want ? want : (rows >= 128 ? (rows / 64 < want ? rows / 64 : want) : 1));
    const int per = ((rows + strips - 1) / strips + 63) / 64 * 64;
    if ((e = hipEventRecord(ctx->hs_entry, (hipStream_t)stream)) != hipSuccess ||
        (e = hipStreamWaitEvent(ctx->hs_render[0], ctx->hs_entry, 0)) != hipSuccess ||
        (e = hipStreamWaitEvent(ctx->hs_render[1], ctx->hs_entry, 0)) != hipSuccess)
        return fail(ctx, SBX_ERR_HIP, "sbx_render_rows_host: entry event", e);
    int g = 0;
    for (int a = 0; a < rows; a += per, ++g) {
        const int b = a + per < rows ? a + per : rows;
        char* dev = ctx->hs_dev + (size_t)a * row_bytes;
        hipStream_t rs = ctx->hs_render[g & 1];
        rc = render_rows(ctx, app, uni, aux, y0 + a, y0 + b, reinterpret_cast<float*>(dev), rs, false);
        if (rc != SBX_OK) break;
        if ((e = hipEventRecord(ctx->hs_ev[g], rs)) != hipSuccess ||
            (e = hipStreamWaitEvent(ctx->hs_copy, ctx->hs_ev[g], 0)) != hipSuccess ||
            (e = hipMemcpyAsync(static_cast<char*>(rgba_host) + (size_t)a * row_bytes, dev, (size_t)(b - a) * row_bytes, hipMemcpyDeviceToHost,
                                ctx->hs_copy)) != hipSuccess) {
            rc = fail(ctx, SBX_ERR_HIP, "sbx_render_rows_host: strip copy", e);
            break;
        }
    }
    (void)hipStreamSynchronize(ctx->hs_render[0]);
    (void)hipStreamSynchronize(ctx->hs_render[1]);
    e = hipStreamSynchronize(ctx->hs_copy);                       // the copies wait for their strips: all of it has run
    if (rc == SBX_OK && e != hipSuccess) rc = fail(ctx, SBX_ERR_HIP, "sbx_render_rows_host: copy", e);
    return rc;
}

// mainImage at `n` arbitrary fragCoords (device arrays): one launch laid out as a pseudo-frame (RowMap.frag)
static const int POINTS_ROW = 256;         // a multiple of every kernel's workgroup width in pixels
int sbx_render_points(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, size_t n, const float* frag,
                      float* rgba, void* stream) {
    if (!ctx) return SBX_ERR_ARG;
    if (!uni) return fail(ctx, SBX_ERR_ARG, "NULL uniforms");
    if (n == 0) return SBX_OK;
    if (!frag || !rgba) return fail(ctx, SBX_ERR_ARG, "NULL point list or output");
    if (n > (size_t)0x7fffffff - POINTS_ROW) return fail(ctx, SBX_ERR_ARG, "too many points");
    // u_res is what fragCoord is divided by (main.h:40) and what the aspect ratio comes from (:33): any positive finite
    // numbers do; only the frame-granular entry points need a whole number of pixels
    if (!(uni->u_res[0] > 0.f) || !(uni->u_res[1] > 0.f) || std::isinf(uni->u_res[0]) || std::isinf(uni->u_res[1]))
        return fail(ctx, SBX_ERR_ARG, "u_res must be positive and finite");
    if (((uintptr_t)rgba & 15u) != 0) return fail(ctx, SBX_ERR_ARG, "output must be 16-byte aligned");
    // the pseudo-frame: 256 columns, wider for very long lists so that the row count stays far below the grid's y limit (65535
    // blocks of as little as 2 rows)
    const int width = POINTS_ROW * (int)((n + (size_t)POINTS_ROW * 100000 - 1) / ((size_t)POINTS_ROW * 100000));
    const int rows = (int)((n + width - 1) / width);
    RowMap M{width, rows, 0, rows, 1, 0, rows, 0, 1, 1, 0, 0, frag, (int)n};
    return render_mapped(ctx, app, uni, aux, M, rgba, stream);
}

static int stage_points(sbx_ctx* ctx, size_t n) {
    if (n <= ctx->pt_cap) return SBX_OK;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    if (ctx->pt_host) (void)hipHostFree(ctx->pt_host);
    ctx->pt_host = nullptr; ctx->pt_cap = 0;
    const size_t cap = n < 1024 ? 1024 : n;
    if ((e = hipHostMalloc((void**)&ctx->pt_host, cap * 6 * sizeof(float), hipHostMallocDefault)) != hipSuccess)
        return fail(ctx, SBX_ERR_HIP, "hipHostMalloc", e);
    ctx->pt_cap = cap;
    return SBX_OK;
}
static int main_image_points(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, size_t n, const float* frag,
                             float* colors) {
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    int rc = stage_points(ctx, n);
    if (rc != SBX_OK) return rc;
    // The pinned staging buffer is device-accessible at its host address: the kernel reads the coordinates from it and stores the
    // colours into it — one launch, one wait, no transfer before or after (round 5; until then H2D copy + launch + D2H copy).
    // Layout: the n colours (16-byte aligned) first, then the n coordinates.
    float* hcol = ctx->pt_host; float* hfrag = ctx->pt_host + 4 * ctx->pt_cap;
    std::memcpy(hfrag, frag, n * 2 * sizeof(float));
    rc = sbx_render_points(ctx, app, uni, aux, n, hfrag, hcol, nullptr);
    if (rc != SBX_OK) return rc;
    if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) return fail(ctx, SBX_ERR_HIP, "point list launch", e);
    std::memcpy(colors, hcol, n * 4 * sizeof(float));
    return SBX_OK;
}
int sbx_main_image_batch(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, size_t n, const float* fragCoords,
                         float* fragColors) {
    if (!ctx) return SBX_ERR_ARG;
    std::lock_guard<std::mutex> g(ctx->mi_lock);
    if (!uni) return fail(ctx, SBX_ERR_ARG, "NULL uniforms");
    if (n == 0) return SBX_OK;
    if (!fragCoords || !fragColors) return fail(ctx, SBX_ERR_ARG, "NULL argument");
    return main_image_points(ctx, app, uni, aux, n, fragCoords, fragColors);
}

// ---- sbx_main_image: the per-pixel entry over cached frames ----------------------------------------------------------------
static int mi_aux_bytes(int app, const void* aux) {
    return !aux ? 0 : ((app == SBX_APP_CLOUDS || app == SBX_APP_CLOUDS_TEX || app == SBX_APP_CLOUDS_SKY) ? (int)sizeof(sbx_aux_clouds)
                       : (app == SBX_APP_SDF_AO ? (int)sizeof(sbx_aux_sdf_ao)
                       : (app == SBX_APP_CLOUDS_UE4 ? (int)sizeof(sbx_aux_clouds_ue4) : 0)));
}
// the cache key of a frame as words: app, aux size, the uniforms, the aux block (zero padded)
static void mi_make_key(int app, const sbx_uniforms* uni, const void* aux, uint32_t key[sbx_ctx::MI_KEY_WORDS]) {
    std::memset(key, 0, sizeof(uint32_t) * sbx_ctx::MI_KEY_WORDS);
    const int ab = mi_aux_bytes(app, aux);
    key[0] = (uint32_t)app; key[1] = (uint32_t)ab;
    std::memcpy(key + 2, uni, sizeof(*uni));
    if (ab) std::memcpy(key + 2 + sizeof(*uni) / 4, aux, (size_t)ab);
}
// Lock-free lookup: true and the pixel if some entry holds this frame and stayed untouched while it was read.
static bool mi_lookup(sbx_ctx* ctx, const uint32_t* key, size_t pixel, float out[4]) {
    for (auto& en : ctx->mi) {
        const uint64_t g1 = en.gen.load(std::memory_order_acquire);
        if (g1 & 1u) continue;                                                   // being rewritten
        bool same = true;
        for (int i = 0; i < sbx_ctx::MI_KEY_WORDS && same; ++i) same = en.key[i].load(std::memory_order_relaxed) == key[i];
        if (!same) continue;
        const float* h = en.host.load(std::memory_order_relaxed);
        if (!h) continue;
        float c[4];
        std::memcpy(c, h + pixel * 4, sizeof(c));
        std::atomic_thread_fence(std::memory_order_acquire);
        if (en.gen.load(std::memory_order_relaxed) != g1) continue;              // rewritten under us: not a hit
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
        return true;
    }
    return false;
}
static unsigned mi_stripe() {
    static std::atomic<unsigned> next{0};
    static thread_local unsigned mine = next.fetch_add(1, std::memory_order_relaxed) & 15u;
    return mine;
}

int sbx_main_image(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, const float fragCoord[2],
                   float fragColor[4]) {
    if (!ctx) return SBX_ERR_ARG;
    if (!uni || !fragCoord || !fragColor) { std::lock_guard<std::mutex> g(ctx->mi_lock); return fail(ctx, SBX_ERR_ARG, "NULL argument"); }
    // Is fragCoord the centre of a pixel of the frame?  Then the pixel comes from the frame cached for (app, uniforms, aux).
    // ANY other coordinate — off-centre (a supersampling host), outside the frame, NaN, or a frame whose u_res is not a whole
    // number of pixels — is evaluated exactly where it is, by a one-point launch: mainImage is a function of fragCoord
    // (src/main.h:40), it never snaps or clamps.
    const float W_f = uni->u_res[0], H_f = uni->u_res[1];
    const int W = (int)W_f, H = (int)H_f;
    const float fx = fragCoord[0], fy = fragCoord[1];
    const bool whole = W > 0 && H > 0 && (float)W == W_f && (float)H == H_f && W <= 65536 && H <= 65536;
    const float cx = std::floor(fx), cy = std::floor(fy);
    const bool centre = whole && fx == cx + .5f && fy == cy + .5f && cx >= 0.f && cy >= 0.f && cx < W_f && cy < H_f;
    if (!centre) {
        std::lock_guard<std::mutex> g(ctx->mi_lock);
        ctx->st_points.fetch_add(1, std::memory_order_relaxed);
        return main_image_points(ctx, app, uni, aux, 1, fragCoord, fragColor);
    }
    uint32_t key[sbx_ctx::MI_KEY_WORDS];
    mi_make_key(app, uni, aux, key);
    const size_t pixel = (size_t)(int)cy * (size_t)W + (size_t)(int)cx;
    // the hit path: no lock, no shared write but a striped counter
    if (mi_lookup(ctx, key, pixel, fragColor)) {
        ctx->st_hits[mi_stripe()].n.fetch_add(1, std::memory_order_relaxed);
        return SBX_OK;
    }
    std::lock_guard<std::mutex> g(ctx->mi_lock);
    if (mi_lookup(ctx, key, pixel, fragColor)) {                                 // another thread rendered it while we waited
        ctx->st_hits[mi_stripe()].n.fetch_add(1, std::memory_order_relaxed);
        return SBX_OK;
    }
    // miss: render the frame into the entry that was filled longest ago (an unused one first)
    sbx_ctx::MiEntry* en = &ctx->mi[0];
    for (auto& c : ctx->mi) if (!c.used || (en->used && c.born < en->born)) { en = &c; if (!c.used) break; }
    const size_t n = (size_t)W * (size_t)H * 4;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    const uint64_t g0 = en->gen.load(std::memory_order_relaxed);
    en->gen.store(g0 + 1, std::memory_order_relaxed);                            // odd: readers stay away / discard what they read
    std::atomic_thread_fence(std::memory_order_release);
    for (auto& w : en->key) w.store(0xffffffffu, std::memory_order_relaxed);     // (no frame has app = -1)
    en->used = false;
    if (n > en->cap_floats) {
        float* fresh = nullptr;
        if ((e = hipHostMalloc((void**)&fresh, n * sizeof(float), hipHostMallocDefault)) != hipSuccess) { en->gen.store(g0 + 2, std::memory_order_release); return fail(ctx, SBX_ERR_HIP, "hipHostMalloc", e); }
        if (float* old = en->host.load(std::memory_order_relaxed)) {
            ctx->mi_retired.push_back(old);                                      // a reader may still be copying its pixel out of it
            if (ctx->mi_retired.size() > 8) { (void)hipHostFree(ctx->mi_retired.front()); ctx->mi_retired.erase(ctx->mi_retired.begin()); }
        }
        en->host.store(fresh, std::memory_order_relaxed);
        en->cap_floats = n;
    }
    float* host = en->host.load(std::memory_order_relaxed);
    // The kernel stores STRAIGHT into the pinned frame (hipHostMalloc memory is device-accessible at its host address): one launch,
    // no device copy of the frame, no transfer behind it — the stores cross PCIe while the rest of the frame is computed (CLOUDS 4K
    // 3.6 ms against 5.1 for launch + copy, PLANET 8K 12.2 against 15.8; profiles/r05_host_boundary.txt).  (The cache holds float
    // pixels whatever the context's output format.)
    const int rc = render_rows(ctx, app, uni, aux, 0, H, host, nullptr, true);
    if (rc != SBX_OK) { en->gen.store(g0 + 2, std::memory_order_release); return rc; }
    if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) {
        en->gen.store(g0 + 2, std::memory_order_release);
        return fail(ctx, SBX_ERR_HIP, "frame render", e);
    }
    ctx->st_frames.fetch_add(1, std::memory_order_relaxed);
    for (int i = 0; i < sbx_ctx::MI_KEY_WORDS; ++i) en->key[i].store(key[i], std::memory_order_relaxed);
    en->used = true; en->born = ++ctx->mi_clock;
    en->gen.store(g0 + 2, std::memory_order_release);                            // even again: published
    std::memcpy(fragColor, host + pixel * 4, 4 * sizeof(float));
    return SBX_OK;
}

int sbx_get_stats(sbx_ctx* ctx, sbx_stats* out) {
    if (!ctx || !out) return SBX_ERR_ARG;
    std::memset(out, 0, sizeof(*out));
    out->render_launches = ctx->st_launches.load(std::memory_order_relaxed);
    for (auto& st : ctx->st_hits) out->main_image_hits += st.n.load(std::memory_order_relaxed);
    out->main_image_frames = ctx->st_frames.load(std::memory_order_relaxed);
    out->main_image_points = ctx->st_points.load(std::memory_order_relaxed);
    return SBX_OK;
}
int sbx_reset_stats(sbx_ctx* ctx) {
    if (!ctx) return SBX_ERR_ARG;
    ctx->st_launches.store(0); ctx->st_frames.store(0); ctx->st_points.store(0);
    for (auto& st : ctx->st_hits) st.n.store(0);
    return SBX_OK;
}

static bool split_ok(int height, int block_rows, int nranks, int root_rounds, int rounds) {
    return height > 0 && block_rows > 0 && nranks > 0 && rounds >= 1 && root_rounds >= 0 && root_rounds <= rounds &&
           (nranks > 1 || root_rounds == rounds);          // a lone rank cannot be relieved
}
int sbx_split_rank_rows(int height, int block_rows, int rank, int nranks, int root_rounds, int rounds) {
    if (!split_ok(height, block_rows, nranks, root_rounds, rounds) || rank < 0 || rank >= nranks) return SBX_ERR_ARG;
    const int nblocks = (height + block_rows - 1) / block_rows;
    const int V = split_cycle_blocks(nranks, root_rounds, rounds);
    const int cnt = rank == 0 ? root_rounds : rounds;
    int rows = 0;
    for (int cycle = 0; cycle * V < nblocks; ++cycle)
        for (int round = 0; round < cnt; ++round) {
            const int v = round < root_rounds ? round * nranks + rank
                                              : root_rounds * nranks + (round - root_rounds) * (nranks - 1) + (rank - 1);
            const int b = cycle * V + v;
            if (b >= nblocks) continue;
            const int y = b * block_rows;
            rows += (y + block_rows <= height) ? block_rows : (height - y);
        }
    return rows;
}
int sbx_split_rows_max(int height, int block_rows, int nranks, int root_rounds, int rounds) {
    if (!split_ok(height, block_rows, nranks, root_rounds, rounds)) return SBX_ERR_ARG;
    int mx = 0;                                             // the fullest slab (equal-count gather: every slab is this tall)
    for (int r = 0; r < nranks; ++r) {
        const int rows = sbx_split_rank_rows(height, block_rows, r, nranks, root_rounds, rounds);
        if (rows > mx) mx = rows;
    }
    return ((mx + block_rows - 1) / block_rows) * block_rows;
}
int sbx_rank_rows(int height, int block_rows, int rank, int nranks) {
    return sbx_split_rank_rows(height, block_rows, rank, nranks, 1, 1);
}
int sbx_rank_rows_max(int height, int block_rows, int nranks) {
    return sbx_split_rows_max(height, block_rows, nranks, 1, 1);
}


// ---------------------------------------------------------------------------------------------
// Spans: which part of a row-block is worth sending to another GPU (include/sbx.h "span exchange")
// ---------------------------------------------------------------------------------------------
// Several apps leave mainImage through an early exit for a large part of the frame: APP_CLOUDS below the horizon
// (src/app_clouds.h:212), APP_ATMOSPHERE outside the dome (acos of an argument below -1 is a NaN direction, the atmosphere test
// fails, src/app_atmosphere.h:196-207,85-88), APP_PLANET where the view ray misses the atmosphere shell (src/app_planet.h:315-321).
// Such pixels cost a few hundred instructions; shipping them costs 12 bytes each on the one xGMI link between their renderer
// and the frame's owner, which at 7680x4320 is the slower of the two by far.  So the split deals out only the SPAN of each
// row-block that the host expects to be expensive, and the owner renders the rest of every block itself, in place.
// The predicate below is a HINT: it repeats the kernels' own early-exit tests with the same math spec on the host, at tile
// corners only, and widens the result by a tile — whoever renders a pixel runs the full kernel on it, so a wrong hint costs
// balance, never a pixel.
namespace {
constexpr int SPAN_ALIGN = 64;        // spans start and end on multiples of this many pixels (every kernel's workgroup width divides it)

struct SpanProbe {
    int app;
    Camera cam;
    bool model;                       // false: no early-exit model for this app, every block's span is the whole row
};
static SpanProbe span_probe(int app, const sbx_uniforms& U, const void* aux) {
    SpanProbe P;
    P.app = app;
    P.model = true;
    switch (app) {
    case SBX_APP_CLOUDS: case SBX_APP_CLOUDS_TEX: case SBX_APP_CLOUDS_SKY: {
        sbx_aux_clouds A;
        if (aux) A = *(const sbx_aux_clouds*)aux; else sbx_aux_clouds_defaults(&A);
        P.cam = build_clouds(U, A).cam;
        break;
    }
    case SBX_APP_ATMOSPHERE: P.cam = build_atmosphere(U).cam; break;
    case SBX_APP_PLANET: P.cam = build_planet(U).cam; break;
    default: P.model = false; P.cam = Camera{}; break;
    }
    return P;
}
// may mainImage at fragCoord (fx, fy) get past the app's early exit?
static bool span_heavy(const SpanProbe& P, float fx, float fy) {
    const v2 pc = point_cam(P.cam, fx, fy);
    switch (P.app) {
    case SBX_APP_CLOUDS: case SBX_APP_CLOUDS_TEX: case SBX_APP_CLOUDS_SKY: {
        const v3 dir = primary_dir(P.cam, pc);
        return !(dir.y < 0.05f);                                   // app_clouds.h:212
    }
    case SBX_APP_ATMOSPHERE: {
        // app_atmosphere.h:195-207: acos(1 - z2) is a NaN beyond z2 = 2 and the atmosphere test then fails (:85-88): free pixels.
        // Below the horizon (1 < z2 <= 2) the view ray dives into the planet from 1 m above the ground: under-ground samples are
        // not lit (get_sun_light returns at its first sample, :65-67), and once exp(-height / hM) has overflowed (106 km down)
        // the kernel is finished with the ray (sbx_atmosphere.h "dead rays").  "Heavy" = some view sample is above the ground
        // BEFORE the optical depth overflows: the same march with the host copy of the math spec.
        const float z2 = pc.x * pc.x + pc.y * pc.y;
        if (z2 > 2.0f) return false;
        const float phi = atan2_(pc.y, pc.x), theta = acos_(1.0f - z2);
        const v3 rd = V3(sin_(theta) * cos_(phi), cos_(theta), sin_(theta) * sin_(phi));
        const float Re = 6360e3f, Ra = 6420e3f;
        const v3 ro = V3(0, Re + 1.f, 0);
        const float tca = dot(V3(0, 0, 0) - ro, rd);
        const float d2 = dot(ro, ro) - tca * tca;
        if (!(d2 < Ra * Ra)) return false;
        const float t1 = tca + sqrt_(Ra * Ra - d2);
        const float step = t1 / 16.f;
        float odM = 0.f;
        for (int i = 0; i < 16; ++i) {
            const v3 sp = ro + rd * (((float)i + .5f) * step);
            const float height = length(sp) - Re;
            odM += exp_(-height / 1200.0f) * step;                 // hM: the first of the two optical depths to overflow
            if (odM > 3e38f) return false;                         // (+inf, or within a rounding of it: the ray is finished)
            if (!(height < 0.f)) return true;
        }
        return false;
    }
    case SBX_APP_PLANET: {                                         // intersect_sphere(eye, {0, 1 + max_height}), kern_planet.hip
        const v3 ro = P.cam.eye, rd = primary_dir(P.cam, pc);
        const float radius = 1.f + .4f;                            // planet.radius + max_height   app_planet.h:16-20,311-312
        const v3 rc = V3(0, 0, 0) - ro;
        const float tca = dot(rc, rd);
        if (tca < 0.f) return false;
        const float d2 = dot(rc, rc) - tca * tca;
        return !(d2 > radius * radius * 1.01f);
    }
    default: return true;
    }
}
}  // namespace

extern "C" int sbx_span_table(int app, const sbx_uniforms* uni, const void* aux, int block_rows, int nranks, int root_rounds,
                              int rounds, int32_t* table, int64_t* rank_pixels, int32_t* max_width) {
    if (!uni) return SBX_ERR_ARG;
    const int W = (int)uni->u_res[0], H = (int)uni->u_res[1];
    if (W <= 0 || H <= 0 || (float)W != uni->u_res[0] || (float)H != uni->u_res[1] || W > 65536 || H > 65536) return SBX_ERR_ARG;
    if (!split_ok(H, block_rows, nranks, root_rounds, rounds)) return SBX_ERR_ARG;
    if (app < SBX_APP_PLANET || app > SBX_APP_PLANET_ATMOSPHERE) return SBX_ERR_UNSUPPORTED;
    const int nblocks = (H + block_rows - 1) / block_rows;
    const int ntiles = (W + SPAN_ALIGN - 1) / SPAN_ALIGN;
    const SpanProbe P = span_probe(app, *uni, aux);
    std::vector<int> x0(nblocks, 0), x1(nblocks, W);
    // The intervals depend on (app, u_res, u_mouse, block_rows) — not on the split — and finding them is the expensive part (tile by
    // tile, nine probes each, a 16-step march per ATMOSPHERE probe: 14 ms for the 8K dome), while a host asks for the same
    // frame's table once per candidate relief and once per rank context: the last few results are kept, process-wide.
    struct Intervals { std::vector<uint32_t> key; std::vector<int> x0, x1; };
    static std::mutex cache_lock;
    static std::vector<Intervals> cache;
    std::vector<uint32_t> key(6, 0u);
    key[0] = (uint32_t)app; key[1] = (uint32_t)block_rows;
    std::memcpy(&key[2], uni->u_res, 8);
    std::memcpy(&key[4], uni->u_mouse, 8);
    bool cached = false;
    if (P.model) {
        std::lock_guard<std::mutex> g(cache_lock);
        for (const Intervals& c : cache)
            if (c.key == key && (int)c.x0.size() == nblocks) { x0 = c.x0; x1 = c.x1; cached = true; break; }
    }
    if (P.model && !cached) {
        // the heavy part of a row-block is taken to be ONE interval of tiles (a disc, a horizon)
        int prev_lo = 1, prev_hi = 0;
        for (int g = 0; g < nblocks; ++g) {
            const int ya = g * block_rows, yb = std::min(H, ya + block_rows) - 1;
            auto tile_heavy = [&](int k) {
                const int xa = k * SPAN_ALIGN, xb = std::min(W, xa + SPAN_ALIGN) - 1;
                const int xs[3] = {xa, (xa + xb) / 2, xb}, ys[3] = {ya, (ya + yb) / 2, yb};
                for (int j = 0; j < 3; ++j)
                    for (int i = 0; i < 3; ++i)
                        if (span_heavy(P, (float)xs[i] + .5f, (float)ys[j] + .5f)) return true;
                return false;
            };
            // The interval of a disc or a horizon moves by a tile or two from one row-block to the next: start from the previous
            // block's ends and walk outwards while heavy, inwards while not (a few probes per block instead of a scan of the whole
            // row: 1.6 s -> tens of ms for the 8K dome on a slow host); blocks after an EMPTY one scan the row on a coarse grid first
            // (every fourth tile: the table is a hint about cost — a span narrower than that, missed, is rendered by the frame's
            // owner, same pixels).
            int lo, hi;
            if (g > 0 && prev_lo <= prev_hi) {
                lo = prev_lo; hi = prev_hi;
                if (tile_heavy(lo)) { while (lo > 0 && tile_heavy(lo - 1)) --lo; }
                else { while (lo <= hi && !tile_heavy(lo)) ++lo; }
                if (lo <= hi) {
                    if (tile_heavy(hi)) { while (hi < ntiles - 1 && tile_heavy(hi + 1)) ++hi; }
                    else { while (hi > lo && !tile_heavy(hi)) --hi; }
                }
            } else {
                lo = 0; hi = ntiles - 1;
                bool any = g == 0;                                 // (the first block: the full scan)
                for (int k = 0; k < ntiles && !any; k += 4) any = tile_heavy(k);
                if (!any && ntiles > 1) any = tile_heavy(ntiles - 1);
                if (!any) { lo = 1; hi = 0; }
                else {
                    while (lo <= hi && !tile_heavy(lo)) ++lo;
                    while (hi > lo && !tile_heavy(hi)) --hi;
                }
            }
            prev_lo = lo; prev_hi = hi;
            if (lo > hi) { x0[g] = 0; x1[g] = 0; continue; }
            lo = std::max(0, lo - 1); hi = std::min(ntiles - 1, hi + 1);          // one tile of slack either side
            x0[g] = lo * SPAN_ALIGN;
            x1[g] = std::min(W, (hi + 1) * SPAN_ALIGN);
        }
        std::lock_guard<std::mutex> g(cache_lock);
        if (cache.size() >= 8) cache.erase(cache.begin());
        cache.push_back(Intervals{key, x0, x1});
    }
    // owner and slab offset of every block: the ranks' local blocks in slab order (sbx_split_rank_rows' enumeration)
    std::vector<int> owner(nblocks, 0), off(nblocks, 0);
    std::vector<long long> pix(nranks, 0);
    const int V = split_cycle_blocks(nranks, root_rounds, rounds);
    int maxw = 0;
    for (int rank = 0; rank < nranks; ++rank) {
        const int cnt = rank == 0 ? root_rounds : rounds;
        for (int cycle = 0; cycle * V < nblocks; ++cycle)
            for (int round = 0; round < cnt; ++round) {
                const int v = round < root_rounds ? round * nranks + rank
                                                  : root_rounds * nranks + (round - root_rounds) * (nranks - 1) + (rank - 1);
                const int g = cycle * V + v;
                if (g >= nblocks) continue;
                const int rows = std::min(H, (g + 1) * block_rows) - g * block_rows;
                owner[g] = rank;
                if (pix[rank] + (long long)rows * W > 0x7fffffffLL) return SBX_ERR_ARG;      // offsets are 32-bit pixel counts
                off[g] = (int)pix[rank];
                pix[rank] += (long long)rows * (x1[g] - x0[g]);
                if (rank > 0) maxw = std::max(maxw, x1[g] - x0[g]);
            }
    }
    if (table)
        for (int g = 0; g < nblocks; ++g) { table[4 * g] = x0[g]; table[4 * g + 1] = x1[g]; table[4 * g + 2] = off[g]; table[4 * g + 3] = owner[g]; }
    if (rank_pixels) for (int r = 0; r < nranks; ++r) rank_pixels[r] = pix[r];
    if (max_width) *max_width = maxw;
    return nblocks;
}

// the device copy of the span table of (app, uniforms, aux, split): looked up among the context's few slots, else built and
// uploaded.  An upload overwrites the least recently filled slot; launches that may still read that slot are waited for
// (hipDeviceSynchronize: a table changes when the frame's geometry does, not per frame).
static int span_table_device(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int nranks,
                             int root_rounds, int rounds, hipStream_t s, const int4** dev, int* max_w, const std::vector<int>** host) {
    // the table depends on the camera (u_res, u_mouse) and the split, not on u_time or the rest of the aux block
    std::vector<int> key = {app, block_rows, nranks, root_rounds, rounds};
    int bits[4];
    std::memcpy(bits, uni->u_res, 8); std::memcpy(bits + 2, uni->u_mouse, 8);
    key.insert(key.end(), bits, bits + 4);
    for (auto& sl : ctx->span_slots)
        if (sl.dev && sl.key == key) { *dev = sl.dev; *max_w = sl.max_w; if (host) *host = &sl.table; return SBX_OK; }
    if (stream_is_capturing(s)) return fail(ctx, SBX_ERR_ARG, "the span table of this frame is not on the device yet: render it once outside the stream capture");
    const int H = (int)uni->u_res[1];
    if (!split_ok(H, block_rows, nranks, root_rounds, rounds)) return fail(ctx, SBX_ERR_ARG, "bad rank split");
    const int nblocks = (H + block_rows - 1) / block_rows;
    sbx_ctx::SpanSlot& sl = ctx->span_slots[ctx->span_next++ % 4];
    sl.key.clear();
    sl.table.assign((size_t)nblocks * 4, 0);
    int mw = 0;
    const int rc = sbx_span_table(app, uni, aux, block_rows, nranks, root_rounds, rounds, sl.table.data(), nullptr, &mw);
    if (rc < 0) return fail(ctx, rc, "bad span table arguments");
    hipError_t e;
    if (sl.dev) (void)hipDeviceSynchronize();                      // earlier launches may still read the slot's old table
    if (sl.cap < (size_t)nblocks) {
        if (sl.dev) (void)hipFree(sl.dev);
        sl.dev = nullptr; sl.cap = 0;
        if ((e = hipMalloc((void**)&sl.dev, (size_t)nblocks * sizeof(int4))) != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipMalloc", e);
        sl.cap = (size_t)nblocks;
    }
    if ((e = hipMemcpy(sl.dev, sl.table.data(), (size_t)nblocks * sizeof(int4), hipMemcpyHostToDevice)) != hipSuccess)
        return fail(ctx, SBX_ERR_HIP, "span table upload", e);
    sl.max_w = mw;
    sl.key = key;
    *dev = sl.dev; *max_w = mw; if (host) *host = &sl.table;
    return SBX_OK;
}

extern "C" int sbx_render_span_peer(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int rank,
                                    int nranks, int root_rounds, int rounds, int r0, int r1, float* rgb, void* stream) {
    int W, H;
    int rc = check_common(ctx, uni, rgb, W, H, 3u);
    if (rc != SBX_OK) return rc;
    if (rank < 1 || rank >= nranks) return fail(ctx, SBX_ERR_ARG, "span slabs are rendered by ranks 1 .. nranks-1");
    const int rows = sbx_split_rank_rows(H, block_rows, rank, nranks, root_rounds, rounds);
    if (rows < 0) return fail(ctx, SBX_ERR_ARG, "bad rank split");
    if (r0 < 0 || r1 < r0 || (r0 % block_rows) != 0) return fail(ctx, SBX_ERR_ARG, "bad slab row range (whole blocks)");
    if (r1 > rows) r1 = rows;
    if (r0 >= r1) return SBX_OK;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    const int4* dev = nullptr; int mw = 0;
    rc = span_table_device(ctx, app, uni, aux, block_rows, nranks, root_rounds, rounds, (hipStream_t)stream, &dev, &mw, nullptr);
    if (rc != SBX_OK) return rc;
    if (mw <= 0) return SBX_OK;                                    // every span of the peers is empty: nothing to render or send
    // `rgb` is the start of the rank's packed slab: the table's offsets are absolute within it
    RowMap M{mw, H, 0, block_rows, nranks, rank, r1 - r0, r0, root_rounds, rounds, 0, out_rgb(ctx, 1), nullptr, 0, dev, 1};
    return render_mapped(ctx, app, uni, aux, M, rgb, stream);
}
extern "C" int sbx_render_span_peer_in_place(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int rank,
                                             int nranks, int root_rounds, int rounds, int channels, float* frame, void* stream) {
    int W, H;
    int rc = check_common(ctx, uni, frame, W, H);
    if (rc != SBX_OK) return rc;
    if (rank < 1 || rank >= nranks) return fail(ctx, SBX_ERR_ARG, "sbx_render_span_peer_in_place is for ranks 1 .. nranks-1");
    if (channels != 3 && channels != 4) return fail(ctx, SBX_ERR_ARG, "channels must be 3 or 4");
    const int rows = sbx_split_rank_rows(H, block_rows, rank, nranks, root_rounds, rounds);
    if (rows < 0) return fail(ctx, SBX_ERR_ARG, "bad rank split");
    if (rows == 0) return SBX_OK;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    const int4* dev = nullptr; int mw = 0;
    rc = span_table_device(ctx, app, uni, aux, block_rows, nranks, root_rounds, rounds, (hipStream_t)stream, &dev, &mw, nullptr);
    if (rc != SBX_OK) return rc;
    if (mw <= 0) return SBX_OK;                                    // every span of the peers is empty
    RowMap M{W, H, 0, block_rows, nranks, rank, rows, 0, root_rounds, rounds, 1, out_rgb(ctx, channels == 3 ? 3 : 0), nullptr, 0, dev, 3};
    return render_mapped(ctx, app, uni, aux, M, frame, stream);
}
extern "C" int sbx_render_span_root(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int nranks,
                                    int root_rounds, int rounds, float* frame, void* stream) {
    int W, H;
    int rc = check_common(ctx, uni, frame, W, H);
    if (rc != SBX_OK) return rc;
    if (!split_ok(H, block_rows, nranks, root_rounds, rounds)) return fail(ctx, SBX_ERR_ARG, "bad rank split");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    const int4* dev = nullptr; int mw = 0;
    rc = span_table_device(ctx, app, uni, aux, block_rows, nranks, root_rounds, rounds, (hipStream_t)stream, &dev, &mw, nullptr);
    if (rc != SBX_OK) return rc;
    // one launch over the whole frame, in place: rank 0's blocks in full, everybody else's outside their spans
    RowMap M{W, H, 0, H, 1, 0, H, 0, 1, 1, 1, out_rgb(ctx, 0), nullptr, 0, dev, 2};
    M.block_rows = block_rows;                                     // (row_to_y of a contiguous map: y = r for any block size)
    return render_mapped(ctx, app, uni, aux, M, frame, stream);
}
extern "C" int sbx_assemble_spans(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int nranks,
                                  int root_rounds, int rounds, const float* peers, int64_t stride_pixels, float* frame, void* stream) {
    int W, H;
    int rc = check_common(ctx, uni, frame, W, H);
    if (rc != SBX_OK) return rc;
    if (!split_ok(H, block_rows, nranks, root_rounds, rounds) || stride_pixels < 0) return fail(ctx, SBX_ERR_ARG, "bad assemble arguments");
    if (nranks == 1) return SBX_OK;
    if (!peers) return fail(ctx, SBX_ERR_ARG, "NULL peers");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    const int4* dev = nullptr; int mw = 0;
    rc = span_table_device(ctx, app, uni, aux, block_rows, nranks, root_rounds, rounds, (hipStream_t)stream, &dev, &mw, nullptr);
    if (rc != SBX_OK) return rc;
    if (mw <= 0) return SBX_OK;
    launch_assemble_spans(W, H, block_rows, dev, peers, (size_t)stride_pixels, frame, (hipStream_t)stream, ctx->out_format != 0);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "assemble launch", e);
    return SBX_OK;
}

static int render_split(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int rank, int nranks,
                        int root_rounds, int rounds, int r0, int r1, int rgb, float* rgba, void* stream) {
    int W, H;
    if (ctx && uni && r0 == r1 && r0 >= 0) return SBX_OK;
    int rc = check_common(ctx, uni, rgba, W, H, rgb ? 3u : 15u);
    if (rc != SBX_OK) return rc;
    const int rows = sbx_split_rank_rows(H, block_rows, rank, nranks, root_rounds, rounds);
    if (rows < 0) return fail(ctx, SBX_ERR_ARG, "bad rank split");
    if (r0 < 0 || r1 < r0) return fail(ctx, SBX_ERR_ARG, "bad slab row range");
    if (r1 > rows) r1 = rows;                 // the slab is padded to the split's rows_max; the tail has no pixels
    if (r0 >= r1) return SBX_OK;
    RowMap M{W, H, 0, block_rows, nranks, rank, r1 - r0, r0, root_rounds, rounds, 0, out_rgb(ctx, rgb)};
    return render_mapped(ctx, app, uni, aux, M, rgba, stream);
}
int sbx_render_split(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int rank,
                     int nranks, int root_rounds, int rounds, int r0, int r1, float* rgba, void* stream) {
    return render_split(ctx, app, uni, aux, block_rows, rank, nranks, root_rounds, rounds, r0, r1, 0, rgba, stream);
}
int sbx_render_split_rgb(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int rank,
                         int nranks, int root_rounds, int rounds, int r0, int r1, float* rgb, void* stream) {
    return render_split(ctx, app, uni, aux, block_rows, rank, nranks, root_rounds, rounds, r0, r1, 1, rgb, stream);
}
int sbx_render_split_in_place(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int rank,
                              int nranks, int root_rounds, int rounds, float* frame, void* stream) {
    int W, H;
    int rc = check_common(ctx, uni, frame, W, H);
    if (rc != SBX_OK) return rc;
    const int rows = sbx_split_rank_rows(H, block_rows, rank, nranks, root_rounds, rounds);
    if (rows < 0) return fail(ctx, SBX_ERR_ARG, "bad rank split");
    if (rows == 0) return SBX_OK;
    RowMap M{W, H, 0, block_rows, nranks, rank, rows, 0, root_rounds, rounds, 1, out_rgb(ctx, 0)};
    return render_mapped(ctx, app, uni, aux, M, frame, stream);
}
int sbx_render_split_in_place_rgb(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int rank,
                                  int nranks, int root_rounds, int rounds, float* frame, void* stream) {
    int W, H;
    int rc = check_common(ctx, uni, frame, W, H);
    if (rc != SBX_OK) return rc;
    const int rows = sbx_split_rank_rows(H, block_rows, rank, nranks, root_rounds, rounds);
    if (rows < 0) return fail(ctx, SBX_ERR_ARG, "bad rank split");
    if (rows == 0) return SBX_OK;
    RowMap M{W, H, 0, block_rows, nranks, rank, rows, 0, root_rounds, rounds, 1, out_rgb(ctx, 3)};      // 3: R, G, B of float4 pixels
    return render_mapped(ctx, app, uni, aux, M, frame, stream);
}
int sbx_render_rank(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int rank,
                    int nranks, float* rgba, void* stream) {
    return sbx_render_split(ctx, app, uni, aux, block_rows, rank, nranks, 1, 1, 0, 0x7fffffff, rgba, stream);
}
int sbx_render_rank_rows(sbx_ctx* ctx, int app, const sbx_uniforms* uni, const void* aux, int block_rows, int rank,
                         int nranks, int r0, int r1, float* rgba, void* stream) {
    return sbx_render_split(ctx, app, uni, aux, block_rows, rank, nranks, 1, 1, r0, r1, rgba, stream);
}

int sbx_assemble_split(sbx_ctx* ctx, int width, int height, int block_rows, int nranks, int root_rounds, int rounds,
                       const float* gathered, float* frame, void* stream) {
    if (!ctx) return SBX_ERR_ARG;
    if (!gathered || !frame || width <= 0 || !split_ok(height, block_rows, nranks, root_rounds, rounds))
        return fail(ctx, SBX_ERR_ARG, "bad assemble arguments");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    launch_assemble(width, height, block_rows, nranks, root_rounds, rounds,
                    sbx_split_rows_max(height, block_rows, nranks, root_rounds, rounds), gathered, frame, (hipStream_t)stream,
                    ctx->out_format != 0);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "assemble launch", e);
    return SBX_OK;
}
int sbx_assemble_peers(sbx_ctx* ctx, int width, int height, int block_rows, int nranks, int root_rounds, int rounds,
                       int channels, const float* peers, float* frame, void* stream) {
    if (!ctx) return SBX_ERR_ARG;
    if (!frame || width <= 0 || !split_ok(height, block_rows, nranks, root_rounds, rounds) || (channels != 3 && channels != 4) ||
        (nranks > 1 && !peers))
        return fail(ctx, SBX_ERR_ARG, "bad assemble arguments");
    if (nranks == 1) return SBX_OK;                           // no peers: the frame is the root's in-place render
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    launch_assemble_peers(width, height, block_rows, nranks, root_rounds, rounds,
                          sbx_split_rows_max(height, block_rows, nranks, root_rounds, rounds), ctx->out_format ? 1 : channels, peers, frame,
                          (hipStream_t)stream);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "assemble launch", e);
    return SBX_OK;
}
int sbx_assemble(sbx_ctx* ctx, int width, int height, int block_rows, int nranks, const float* gathered,
                 float* frame, void* stream) {
    return sbx_assemble_split(ctx, width, height, block_rows, nranks, 1, 1, gathered, frame, stream);
}

int sbx_pack_unorm8(sbx_ctx* ctx, int width, int rows, const float* rgba, unsigned char* out, int flip_y, void* stream) {
    if (!ctx) return SBX_ERR_ARG;
    if (width <= 0 || rows < 0 || (rows > 0 && (!rgba || !out))) return fail(ctx, SBX_ERR_ARG, "bad pack arguments");
    if (rows == 0) return SBX_OK;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    launch_pack_unorm8(width, rows, flip_y != 0, rgba, out, (hipStream_t)stream);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "pack launch", e);
    return SBX_OK;
}

int sbx_set_output_format(sbx_ctx* ctx, int format) {
    if (!ctx) return SBX_ERR_ARG;
    if (format != SBX_FORMAT_RGBA32F && format != SBX_FORMAT_RGBA8) return fail(ctx, SBX_ERR_ARG, "unknown output format");
    ctx->out_format = format;
    return SBX_OK;
}
int sbx_set_precision(sbx_ctx* ctx, int precision) {
    if (!ctx) return SBX_ERR_ARG;
    if (precision != SBX_PRECISION_EXACT && precision != SBX_PRECISION_1E4) return fail(ctx, SBX_ERR_ARG, "unknown precision tier");
    ctx->precision = precision;
    {   // frames cached by sbx_main_image were rendered in the other tier
        std::lock_guard<std::mutex> g(ctx->mi_lock);
        for (auto& en : ctx->mi) {
            const uint64_t g0 = en.gen.load(std::memory_order_relaxed);
            en.gen.store(g0 + 1, std::memory_order_relaxed);
            std::atomic_thread_fence(std::memory_order_release);
            for (auto& w : en.key) w.store(0xffffffffu, std::memory_order_relaxed);
            en.used = false;
            en.gen.store(g0 + 2, std::memory_order_release);
        }
    }
    return SBX_OK;
}
int sbx_set_variant(sbx_ctx* ctx, int variant) {
    if (!ctx) return SBX_ERR_ARG;
    if (variant < 0 || variant > 3) return fail(ctx, SBX_ERR_ARG, "unknown kernel variant");
    ctx->variant = variant == 1 ? 1 : 0;
    ctx->sdf_roots = variant >= 2 ? variant : 0;
    return SBX_OK;
}
int sbx_set_timing(sbx_ctx* ctx, int enabled) {
    if (!ctx) return SBX_ERR_ARG;
    ctx->timing = enabled != 0;
    return SBX_OK;
}
int sbx_last_kernel_ms(sbx_ctx* ctx, float* ms) {
    if (!ctx || !ms) return SBX_ERR_ARG;
    if (ctx->last_timer < 0 || !ctx->timers[ctx->last_timer].second.complete) return fail(ctx, SBX_ERR_ARG, "no timed launch yet");
    TimingPair& p = ctx->timers[ctx->last_timer].second;          // the pair of the stream of the last timed launch
    hipError_t e = hipEventSynchronize(p.ev1);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipEventSynchronize", e);
    e = hipEventElapsedTime(ms, p.ev0, p.ev1);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipEventElapsedTime", e);
    return SBX_OK;
}

int sbx_math_eval(sbx_ctx* ctx, const char* fn, const float* a, const float* b, float* out, size_t n, void* stream) {
    if (!ctx) return SBX_ERR_ARG;
    if (!fn || !a || !out) return fail(ctx, SBX_ERR_ARG, "NULL argument");
    static const char* names[] = {"sin", "cos", "tan", "exp", "pow", "acos", "atan2", "hash", "div", "div_rd", "exp_h13", "pow_h", "sqrt_n", "sqrt_ieee", "exp_reg", "exp_reg_plain", "exp_reg64", "exp_reg64_plain", "exp_small", "exp_small_plain", "exp_reg4k", "sin_b40", "div3", "sqrt_rs", "divn", "srgb_pow", "pow_spec"};
    int id = -1;
    for (int i = 0; i < 27; ++i) if (std::strcmp(fn, names[i]) == 0) id = i;
    if (id < 0) return fail(ctx, SBX_ERR_ARG, "unknown math function");
    if ((id == 4 || id == 6 || id == 8 || id == 9 || id == 11 || id == 22 || id == 24 || id == 26) && !b) return fail(ctx, SBX_ERR_ARG, "binary function needs b");
    if (n == 0) return SBX_OK;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    if (id >= 21) { if (launch_math_eval(id, a, b, out, n, (hipStream_t)stream) != 0) return fail(ctx, SBX_ERR_ARG, "math function not in the kernel"); }
    else if (id == 20) launch_exp4k_eval(a, out, n, (hipStream_t)stream);                 // k_atmosphere's 4096-entry form
    else if (id >= 14) launch_cl_exp_eval(a, out, n, (hipStream_t)stream, id - 14);   // exp_reg_ of sbx_math.h (|x| <= 80): k_clouds' / k_atmosphere's form
    else if (launch_math_eval(id, a, b, out, n, (hipStream_t)stream) != 0) return fail(ctx, SBX_ERR_ARG, "math function not in the kernel");
    e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "math_eval launch", e);
    return SBX_OK;
}

int sbx_noise_eval(sbx_ctx* ctx, const char* fn, const float* xyz, const float* params, float* out, size_t n,
                   void* stream) {
    if (!ctx) return SBX_ERR_ARG;
    if (!fn || !xyz || !out) return fail(ctx, SBX_ERR_ARG, "NULL argument");
    static const char* names[] = {"noise_iq", "hash_w", "noise_w", "fbm_worley_tile", "normalize", "wit_normalize", "wit_record"};
    int id = -1;
    for (int i = 0; i < 7; ++i) if (std::strcmp(fn, names[i]) == 0) id = i;
    if (id < 0) return fail(ctx, SBX_ERR_ARG, "unknown noise function");
    const float zero[3] = {0.f, 0.f, 0.f};
    if ((id == 2 || id == 3) && !params) return fail(ctx, SBX_ERR_ARG, "noise_w / fbm_worley_tile need params");
    if (n == 0) return SBX_OK;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    launch_noise_eval(id, xyz, params ? params : zero, out, n, (hipStream_t)stream);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "noise_eval launch", e);
    return SBX_OK;
}

int sbx_worley_volume(sbx_ctx* ctx, int size, float* rgba, void* stream) {
    if (!ctx) return SBX_ERR_ARG;
    if (!rgba || size <= 0 || size > 1024) return fail(ctx, SBX_ERR_ARG, "bad volume arguments");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    launch_worley_volume(size, rgba, (hipStream_t)stream);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "worley_volume launch", e);
    return SBX_OK;
}

int sbx_set_noise_volumes(sbx_ctx* ctx, int shape_size, const float* shape_rgba, int detail_size, const float* detail_rgba,
                          void* stream) {
    if (!ctx) return SBX_ERR_ARG;
    if (!shape_rgba || !detail_rgba || shape_size <= 0 || detail_size <= 0 || shape_size > 1024 || detail_size > 1024)
        return fail(ctx, SBX_ERR_ARG, "bad noise volume arguments");
    ctx->tex_bounds_valid = false;                                 // whatever happens below, the old volumes' bounds are gone
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    const size_t n1 = (size_t)shape_size * shape_size * shape_size, n2 = (size_t)detail_size * detail_size * detail_size;
    // (re)allocation frees buffers an in-flight render may read: hipFree synchronises the device first
    if (shape_size != ctx->noise_tex_size || !ctx->noise_tex) {
        if (ctx->noise_tex) (void)hipFree(ctx->noise_tex);
        ctx->noise_tex = nullptr; ctx->noise_tex_size = 0;
        if ((e = hipMalloc((void**)&ctx->noise_tex, n1 * sizeof(float))) != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipMalloc", e);
        ctx->noise_tex_size = shape_size;
    }
    if (detail_size != ctx->noise_tex2_size || !ctx->noise_tex2) {
        if (ctx->noise_tex2) (void)hipFree(ctx->noise_tex2);
        ctx->noise_tex2 = nullptr; ctx->noise_tex2_size = 0;
        if ((e = hipMalloc((void**)&ctx->noise_tex2, n2 * sizeof(float))) != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipMalloc", e);
        ctx->noise_tex2_size = detail_size;
    }
    {   // cached sbx_main_image frames may have used the old volumes
        std::lock_guard<std::mutex> g(ctx->mi_lock);
        for (auto& en : ctx->mi) {
            const uint64_t g0 = en.gen.load(std::memory_order_relaxed);
            en.gen.store(g0 + 1, std::memory_order_relaxed);
            std::atomic_thread_fence(std::memory_order_release);
            for (auto& w : en.key) w.store(0xffffffffu, std::memory_order_relaxed);
            en.used = false;
            en.gen.store(g0 + 2, std::memory_order_release);
        }
    }
    launch_extract_r(shape_rgba, ctx->noise_tex, n1, (hipStream_t)stream);
    launch_extract_r(detail_rgba, ctx->noise_tex2, n2, (hipStream_t)stream);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "noise volume copy launch", e);
    // The range of the texel values: k_clouds_tex derives a bound on the density from it and, inside that bound, uses the cheaper
    // exp form that is equal to exp_ there (kern_clouds_tex.hip clouds_tex_density_bound).  The scan is read back here, so the call
    // waits for its own copies; inside a stream capture nothing can be read back and the volumes stay without bounds (exp_ itself).
    ctx->tex_bounds_valid = false;
    if (!stream_is_capturing((hipStream_t)stream)) {
        if (!ctx->tex_scan && (e = hipMalloc((void**)&ctx->tex_scan, 6 * sizeof(unsigned))) != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipMalloc", e);
        launch_minmax_r(ctx->noise_tex, n1, ctx->tex_scan, (hipStream_t)stream);
        launch_minmax_r(ctx->noise_tex2, n2, ctx->tex_scan + 3, (hipStream_t)stream);
        unsigned h[6];
        if ((e = hipMemcpyAsync(h, ctx->tex_scan, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream)) != hipSuccess ||
            (e = hipStreamSynchronize((hipStream_t)stream)) != hipSuccess) return fail(ctx, SBX_ERR_HIP, "noise volume scan", e);
        if (!h[2] && !h[5] && h[0] <= h[1] && h[3] <= h[4]) {
            ctx->tex_bounds[0] = minmax_key_to_float(h[0]); ctx->tex_bounds[1] = minmax_key_to_float(h[1]);
            ctx->tex_bounds[2] = minmax_key_to_float(h[3]); ctx->tex_bounds[3] = minmax_key_to_float(h[4]);
            ctx->tex_bounds_valid = true;
        }
    }
    return SBX_OK;
}

int sbx_tex3d_eval(sbx_ctx* ctx, int size, const float* rgba, const float* xyz, float* out, size_t n, void* stream) {
    if (!ctx) return SBX_ERR_ARG;
    if (!rgba || !xyz || !out || size <= 0 || size > 1024) return fail(ctx, SBX_ERR_ARG, "bad tex3d arguments");
    if (n == 0) return SBX_OK;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    launch_tex3d_eval(size, rgba, xyz, out, n, (hipStream_t)stream);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "tex3d_eval launch", e);
    return SBX_OK;
}

int sbx_fault_status(sbx_ctx* ctx) {
    if (!ctx) return SBX_ERR_ARG;
    return device_fault(ctx) ? fail(ctx, SBX_ERR_FAULT, fault_text(ctx)) : SBX_OK;
}
int sbx_clear_fault(sbx_ctx* ctx) {
    if (!ctx) return SBX_ERR_ARG;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    if ((e = hipDeviceSynchronize()) != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipDeviceSynchronize", e);   // no launch may still set it
    if (ctx->device >= 0 && ctx->device < 64 && g_fault_word[ctx->device]) *(volatile unsigned*)g_fault_word[ctx->device] = 0u;
    return SBX_OK;
}
int sbx_debug_raise_fault(sbx_ctx* ctx, void* stream) {
    if (!ctx) return SBX_ERR_ARG;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "hipSetDevice", e);
    launch_raise_fault(1u, (hipStream_t)stream);
    if ((e = hipGetLastError()) != hipSuccess) return fail(ctx, SBX_ERR_HIP, "fault kernel launch", e);
    return SBX_OK;
}

int sbx_debug_tile_order(sbx_ctx* ctx, int app, int* tables_built, int* launches_since, unsigned* table, size_t capacity) {
    if (!ctx || app < 0 || app >= 16) return SBX_ERR_ARG;
    TileOrder& T = tile_order_latest(ctx->tile_orders, app);   // the shape used last
    if (tables_built) *tables_built = T.built;
    if (launches_since) *launches_since = T.age;
    if (!table || T.cur < 0) return 0;
    const size_t n = (size_t)T.key[4] * (size_t)T.key[5];
    if (n > capacity) return fail(ctx, SBX_ERR_ARG, "sbx_debug_tile_order: table larger than the buffer");
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(table, T.mem + T.cap * (size_t)(2 + T.cur), n * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(ctx, SBX_ERR_HIP, "sbx_debug_tile_order", e);
    return (int)n;
}

const char* sbx_last_error(sbx_ctx* ctx) {
    if (!ctx) return "no context";
    if (device_fault(ctx) && ctx->err.find(fault_text(ctx)) == std::string::npos) { ctx->err += ctx->err.empty() ? "" : "; "; ctx->err += fault_text(ctx); }
    return ctx->err.c_str();
}
const char* sbx_version(void) { return "libsbx 0.2 (gfx950, ABI 2)"; }
int sbx_abi_version(void) { return SBX_ABI_VERSION; }

}  // extern "C"
