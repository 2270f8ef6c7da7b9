// shaderbox_amd/csrc/sbx_frame.h — per-frame constant blocks and the row map.
//
// The reference recomputes frame constants per pixel (mainImage calls setup_camera() and
// setup_scene() for every fragment, /root/reference/src/main.h:35-38; APP_EGG even rebuilds its
// rotations and IK inside every sdf() call, src/app_egg.h:40-128).  They depend only on the
// uniforms, so the host evaluates them ONCE per frame with the same math spec (sbx_math.h is
// bit-identical on host and device) and passes them as kernel arguments: on gfx950 kernel
// arguments are wave-uniform and live in SGPRs, which is the "per-tile camera/uniform
// constants" staging the design calls for without spending LDS or VGPRs on them.
#pragma once
#include <hip/hip_runtime.h>
#include "sbx_vec.h"

namespace sbx {

// local row r of a launch -> global row y of the frame.
//   contiguous strip:  block_rows = nrows, nranks = 1, rank = 0      -> y = y0 + r
//   cyclic row-blocks: rank owns blocks rank, rank+nranks, ...        
//   cyclic with root relief: the blocks are dealt in cycles of `rounds` rounds; a round deals one block to every
//     rank, except that rank 0 (the gather's root, which also receives and assembles the frame) is left out of the
//     rounds >= root_rounds.  root_rounds = rounds = 1 is the plain cyclic split.
struct RowMap {
    int width, height;
    int y0, block_rows, nranks, rank;
    int nrows;   // local rows in this launch
    int r0;      // first local row of this launch within the rank's slab (sub-range launches)
    int root_rounds, rounds;
    int in_place;   // 0: rows are written densely (slab row r); 1: at their global row y of a full-size frame
    int rgb;        // 1: the output holds 3 floats per pixel (a peer's slab on its way to the root: alpha is the constant 1 of
                    //    main.h:52 and need not cross xGMI); 0: float4 pixels; 2: one R8G8B8A8_UNORM word per pixel
                    //    (sbx_set_output_format: the display format of the reference's hosts, 4 bytes per pixel everywhere);
                    //    3: float4 pixels of which only R, G, B are written (sbx_render_split_in_place_rgb)
    // POINT LIST (sbx_render_points / sbx_main_image): frag != NULL makes the launch evaluate mainImage at `npoints` arbitrary
    // fragCoords read from device memory (x, y interleaved) instead of at the pixel centres of a row range: the launch is laid
    // out as a pseudo-frame of `width` columns whose "pixel" (x, r) is point r * width + x; results are written densely, 4
    // (or 3) floats per point.  u_res stays the real frame's: fragCoord / u_res is what mainImage computes (main.h:40).
    const float* frag;
    int npoints;
    // SPANS (sbx_render_span_peer / sbx_render_span_root; the multi-GPU exchange that ships only the expensive part of a
    // row-block): span[g] = {x0, x1, off, owner} for every GLOBAL row-block g of the frame — pixels x0 <= x < x1 of the block
    // are the part that is sharded ("heavy": the host expects real work there), the rest leaves mainImage through an early
    // exit and is rendered by the frame's owner itself; `off` = where the block's span starts in its owner's packed slab
    // (in pixels), `owner` = the rank the split deals the block to.
    //   span_mode 1 (a peer): the launch covers the rank's local rows x the widest span; a pixel is rendered iff it lies in
    //                 its block's span and lands at slab pixel off + row_in_block * (x1 - x0) + (x - x0);
    //   span_mode 2 (the owner of the frame, in place over the WHOLE frame): a pixel is rendered iff its block is rank 0's
    //                 or it lies OUTSIDE its block's span.
    //   span_mode 3 (a peer of the store exchange, sbx_render_span_peer_in_place): the launch covers the rank's rows at the frame's
    //                 full width; a pixel is rendered iff it lies in its block's span and lands at its GLOBAL position of the owner's
    //                 full-size frame (waves outside the span exit at once).
    // Either way every pixel is computed by the same kernel from its global coordinates: the spans decide who renders a
    // pixel, never what it looks like.
    const int4* span;
    int span_mode;
    // DISPATCH ORDER (round 6; a hint about cost, never about pixels): order != NULL makes workgroup i of the launch render tile
    // order[i] = bx | by << 16 instead of tile i — a permutation of the launch's tiles, built from the previous frame's per-tile
    // cost, longest first (sbx_tile_order.h, kern_util.hip k_order_count / _scan / _place); cost != NULL: lane 0 of every wave's first tile
    // column writes its wave's duration (100 MHz ticks) to cost[tile].
    const unsigned* order = nullptr;
    unsigned* cost = nullptr;
};
// blocks in one cycle, and the position of (round, rank) in it
SBX_HD int split_cycle_blocks(int nranks, int root_rounds, int rounds) {
    return root_rounds * nranks + (rounds - root_rounds) * (nranks - 1);
}
SBX_HD int row_to_y(const RowMap& m, int r) {
    const int rr = r + m.r0;
    const int blk = rr / m.block_rows;                       // local block index in the rank's slab
    const int cnt = (m.rank == 0) ? m.root_rounds : m.rounds;    // blocks this rank gets per cycle
    const int cycle = blk / cnt, round = blk - cycle * cnt;
    const int v = (round < m.root_rounds) ? round * m.nranks + m.rank
                                          : m.root_rounds * m.nranks + (round - m.root_rounds) * (m.nranks - 1) + (m.rank - 1);
    const int gblk = cycle * split_cycle_blocks(m.nranks, m.root_rounds, m.rounds) + v;
    return m.y0 + gblk * m.block_rows + (rr - blk * m.block_rows);
}

// Camera part of mainImage (src/main.h:33-48) + get_primary_ray (src/util.h:5-20)
struct Camera {
    float res_x, res_y;
    float aspect_x;   // u_res.x / u_res.y
    float fov;
    v3 eye, fwd, up, right;
    double rres_x, rres_y;   // recip64(u_res): fragCoord / u_res as an exact multiply (sbx_math.h div_by)
};
SBX_HD Camera make_camera(float res_x, float res_y, float fov, v3 eye, v3 look_at) {
    Camera c;
    c.res_x = res_x; c.res_y = res_y;
    c.aspect_x = res_x / res_y;            // main.h:33
    c.fov = fov;
    c.rres_x = recip64(res_x); c.rres_y = recip64(res_y);
    c.eye = eye;
    c.fwd = normalize(look_at - eye);      // util.h:10
    v3 up = V3(0, 1, 0);
    c.right = cross(up, c.fwd);            // util.h:12
    c.up = cross(c.fwd, c.right);          // util.h:13
    return c;
}
// point_cam.xy for fragCoord (main.h:40,44-46); point_cam.z = -1
SBX_HD v2 point_cam(const Camera& c, float fx, float fy) {
    float nx = div_by(fx, c.rres_x), ny = div_by(fy, c.rres_y);   // main.h:40
    return V2(((2.0f * nx - 1.0f) * c.aspect_x) * c.fov, ((2.0f * ny - 1.0f) * 1.0f) * c.fov);
}
SBX_HD v3 primary_dir(const Camera& c, v2 pc) {
    return normalize(c.fwd + c.up * pc.y + c.right * pc.x);   // util.h:17
}
// linear_to_srgb (src/util.h:72-77): p = 1/2.2 in binary32
#ifndef SBX_SRGB_FAST
#define SBX_SRGB_FAST 1     // device: srgb_pow_ of sbx_math.h (pow_'s operations in ~45 instead of ~100 instructions; equal on all 2^32 arguments)
#endif
SBX_HD v3 to_srgb(v3 c) {
#if SBX_SRGB_FAST
    return V3(srgb_pow_(c.x), srgb_pow_(c.y), srgb_pow_(c.z));
#else
    const float p = 1.f / 2.2f;
    return V3(pow_(c.x, p), pow_(c.y, p), pow_(c.z, p));
#endif
}

// ---- APP_CLOUDS (src/app_clouds.h, src/uniform_buffer.h:39-55) ----------------------------
struct FrameClouds {
    Camera cam;
    v3 sun_dir, sun_color;
    v3 wind_off;          // wind_dir * u_time * (1/cld_noise_factor)   app_clouds.h:167
    float sun_power, sigma;
    int steps, lsteps;
    float dt;             // cld_thick / float(cld_march_steps)         app_clouds.h:98,180
    float cov, cov_hi;    // 1 - cld_coverage, cov + .0135              app_clouds.h:83-84
    double cov_rd;        // recip64(cov_hi - cov): the smoothstep division becomes an exact multiply
    float cov_d, cov_r;   // cov_hi - cov and RN(1 / that): the same division through div3_ (sbx_math.h) in k_clouds' SM kernels
    float thr1, thr2;     // cov - .1876, cov - .06255: the staged main sample's two cut-offs (launch_clouds)
    int lip_ok;           // 1: the march positions are small enough for k_clouds' Lipschitz sample skip (set per launch by
                          //    launch_clouds / clouds_lip_domain; 0 disables the skip, nothing else)
    int exp_small;        // 1: every exp argument of the REG kernels lies in [-0.205, -0] (sigma, dt >= 0, .94 sigma dt <= .205): they
                          //    use exp_small_ of sbx_math.h instead of exp_reg64_ (set per launch by launch_clouds; same bits)
    // SKY_SPHERE build (app_clouds.h:8,14-19,154-162; SBX_APP_CLOUDS_SKY): the march starts where the view ray meets a sphere
    // around the viewer (intersect_sphere_from_inside) and runs along the view ray itself; the table-less kernels only
    int sky;              // 1: SKY_SPHERE
    float nf;             // cld_noise_factor: .001, or (1 / atm_radius) * 10 for SKY_SPHERE            app_clouds.h:18,20
    float atm_y, atm_r;   // atmosphere = sphere((0, atm_ground_y, 0), atm_radius)                     app_clouds.h:15-17
    m3 sky_rot;           // rotate_around_x(u_time)                                                   app_clouds.h:160
};

// ---- APP_EGG (src/app_egg.h) ----------------------------------------------------------------
struct BezierFrame {      // the P-independent part of sd_bezier (src/sdf.h:147-153)
    v3 b, u, v, w;
    v2 a2, c2;
    v3 bc; float br;      // a sphere around the control triangle (hence around the curve), radius rounded up: sbx_sdf.h bezier_far
};
struct CylFrame {         // the P-independent part of sd_cylinder with P0 = 0 (src/sdf.h:104,106-107)
    v3 dir;
    float len1, len0;
};
struct FrameEgg {
    Camera cam;
    m3 rot_y;             // rotate_around_y(u_time * -100)            app_egg.h:40
    v3 left_foot, right_foot;   //                                      app_egg.h:73-77
    BezierFrame leg_l, leg_r;   //                                      app_egg.h:111-116
    CylFrame foot_l, foot_r;    //                                      app_egg.h:120-128
    v3 foot_ml, foot_mr;        // midpoints of the toe cylinders in p space: -foot - toe/16 (kern_egg.hip)
    v3 oc; float orad;          // sphere around everything but the ground plane, in sdf()'s p space (kern_egg.hip egg_far)
    v3 ocw;                     // the same centre in WORLD space, rot_y^T (oc + (0, .5, 3.5)): the cull test needs no rotation
};

// ---- APP_RAYTRACER (src/app_raytracer.h, cornell_box.h) -------------------------------------
struct RtPlane { v3 n; float d; int mat; };
struct RtSphere { v3 o; float r; int mat; double rr; };   // rr = recip64(r): the normal's three divisions by r (intersect.h:32)
struct RtMaterial { v3 base_color; float roughness, ior, reflectivity;
                    float r0; };   // fresnel_factor(1, ior, .)'s R0 = ((1 - ior) / (1 + ior))^2, util_optics.h:10-11
struct FrameRaytracer {
    Camera cam;
    RtPlane planes[6];
    RtSphere spheres[3];
    RtMaterial mats[8];
    v3 light;             // lights[0].L
};

// ---- APP_ATMOSPHERE (src/app_atmosphere.h) --------------------------------------------------
struct FrameAtmosphere {
    Camera cam;
    v3 sun_dir;           // (0,1,0) * rotate_around_x(-|sin(t/2)|*90)  app_atmosphere.h:177-181
};

// ---- APP_SDF_AO (src/app_sdf_ao.h) ----------------------------------------------------------
struct FrameSdfAo {
    Camera cam;
    m3 rx_m90;            // rotate_around_x(-90)   app_sdf_ao.h:63,77
    m3 ry_180;            // rotate_around_y(180)   app_sdf_ao.h:134
    v3 sun_dir;           // normalize(1,2,1)       app_sdf_ao.h:209
    float fog_density, fog_falloff;
};

// ---- APP_PLANET (src/app_planet.h) ----------------------------------------------------------
struct FramePlanet {
    Camera cam;
    m3 rot, rot_cloud, rot_t;   // app_planet.h:307-309, transpose(rot) :356
    v3 L;                       // rot * normalize(1,1,0)   app_planet.h:289
    int atm_sky;                // 1: SBX_APP_PLANET_ATMOSPHERE — background() is APP_ATMOSPHERE's get_incident_light (kern_planet.hip)
    v3 atm_sun;                 // its sun: (0,1,0) * rotate_around_x(-|sin(t/2)|*90)   app_atmosphere.h:177-181
};

// ---- "clouds_best" (src/app_clouds_best.h, the stand-alone cloud shader) ------
// The march runs along projection = dir / dir.y, whose y component is exactly 1, from origin.y = eye.y + 100:
// sample height, `cloud.height`, illuminate_volume() = exp(height)/1.95 and the y coordinate of the five
// noise octaves are the same for every pixel.  One row per march step, built on the host with the math spec.
constexpr int CB_STEPS = 50;                                   // cld_march_steps :410
struct CBRow { float qy[5]; float illum; };
struct FrameCloudsBest {
    Camera cam;
    v3 sun_dir;               // normalize(0, 0, -1)                                          :415
    float wind_z;             // -u_time * .2                                                 :414
    float march_step;         // cld_thick / float(steps)                                     :603
    float cov;                // cld_coverage                                                 :411
    double cov_rd;            // recip64((cov + .035) - cov)                                  :583
    CBRow row[CB_STEPS];
};

// ---- the UE4 cloud variant (ue4/volumetric_clouds/Shaders/app_clouds.usf) ----------------
constexpr int UE4_STEPS = 25;                                  // STEPS :16
struct FrameCloudsUe4 {
    Camera cam;               // host mapping: APP_CLOUDS' camera (src/app_clouds.h:23-30), FOV 1
    v3 sun_dir, wind_dir;     // SUN_DIR / WIND_DIR :13-14 or the aux block's
    float march_step;         // thickness / float(steps)                                     :199
    float absorbtion;
    float cov;                // 1 - coverage                                                 :256
    double cov_rd;            // recip64((cov + fuzziness) - cov)                             :175
    float cov_d, cov_r;       // (cov + fuzziness) - cov and RN(1 / that): the same division through div3_ (sbx_math.h)
    float eh[UE4_STEPS];      // exp(h) / 1.75 with h = float(i) / float(steps)               :213,221
};

// ---- APP_VINYL (src/app_vinyl.h; C++ build) -------------------------------------------------
struct Capsule { v3 a, ab; double rd; };   // sd_capsule(p, a, b, r): ab = b - a, rd = recip64(dot(ab, ab))
struct FrameVinyl {
    Camera cam;
    m3 platter_rot;            // rotate_around_y(200 t) * rotate_around_x(sin(t) * .1)      app_vinyl.h:424-426
    v3 sun_dir;                // normalize(-1, 4, -3)                                        :285-286
    m3 ry30, rym30;            // logo                                                        :74,77
    m3 wobble;                 // rotate_around_x(sin(3.6758 t) * .1)                         :141-142
    Capsule arm1, arm2, arm3;  //                                                             :151-153
    BezierFrame armb;          // sd_bezier(a11, a2, a33, ., R)                               :154
    v3 a3;                     //                                                             :150
    m3 arm_xform, fl_rot, fl_rot2, ctg_rot, cut_rx10, cut_rym5, cut2_rz10;   // :163-232
    v3 fl_sub1, fl_sub2;       // arm_right * clr_r, arm_up * clr_r                           :191-193
    CylFrame collar;           // sd_cylinder(., 0, arm_fwd * .05, .)                         :173-176
    int steps;                 // march steps: 60 (the C++ build) or 180 (the GLSL / HLSL builds, SBX_APP_VINYL_GPU)   :411-416
};

}  // namespace sbx
