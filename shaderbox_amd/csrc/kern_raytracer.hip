// shaderbox_amd/csrc/kern_raytracer.hip — APP_RAYTRACER: Cornell box, Cook-Torrance, 2 bounces.
//
// Follows /root/reference/src/app_raytracer.h (render :88-136, raytrace_iteration :70-86,
// illuminate :46-68), src/intersect.h (intersect_plane :61-77, intersect_sphere :7-33),
// src/light.h (illum_cook_torrance :64-92), src/util_optics.h (fresnel_factor :5-14, reflect
// :17-22), src/material.h, src/cornell_box.h.  The whole scene (6 planes, 3 spheres, 8 material
// slots, the light) is frame-constant: it is built on the host (setup_scene :18-36 +
// setup_cornell_box cornell_box.h:39-87) and arrives as kernel arguments in SGPRs.
#include <cmath>
#include "sbx_device.h"
#include "sbx_ldsframe.h"
#include "sbx_witness.h"

#ifndef RT_LDS_FRAME
#define RT_LDS_FRAME 1      // the scene block (130 floats: does not fit the SGPR file) in LDS, read at its uses (sbx_ldsframe.h)
#endif

namespace sbx {

#if RT_LDS_FRAME
__device__ __forceinline__ RtPlane rt_ld(const RtPlane& r) { return RtPlane{lds_ld(r.n), lds_ld(r.d), lds_ld(r.mat)}; }
__device__ __forceinline__ RtSphere rt_ld(const RtSphere& r) { return RtSphere{lds_ld(r.o), lds_ld(r.r), lds_ld(r.mat), lds_ld(r.rr)}; }
__device__ __forceinline__ RtMaterial rt_ld(const RtMaterial& r) {
    return RtMaterial{lds_ld(r.base_color), lds_ld(r.roughness), lds_ld(r.ior), lds_ld(r.reflectivity), lds_ld(r.r0)};
}
__device__ __forceinline__ v3 rt_ld(const v3& r) { return lds_ld(r); }
__device__ __forceinline__ int rt_ld(const int& r) { return lds_ld(r); }
#else
template <class T> __device__ __forceinline__ const T& rt_ld(const T& r) { return r; }
#endif

struct Hit { float t; int mat; v3 n, o; };

__device__ __forceinline__ float fresnel_factor(float n1, float n2, float VdotH) {   // util_optics.h:5-14
    float Rn = (n1 - n2) / (n1 + n2);
    float R0 = Rn * Rn;
    float F = 1.f - VdotH;
    return R0 + (1.f - R0) * (F * F * F * F * F);
}

__device__ __forceinline__ void hit_plane(v3 ro, v3 rd, const RtPlane& p, Hit& hit) {  // intersect.h:61-77
    const float denom = dot(p.n, rd);
    if (denom < 1e-6f) return;
    const v3 P0 = V3(p.d, p.d, p.d);
    const float t = dot(P0 - ro, p.n) / denom;
    if (t < 0.f || t > hit.t) return;
    hit.t = t;
    hit.mat = p.mat;
    hit.o = ro + rd * t;
    hit.n = dot(p.n, rd) < 0 ? p.n : -p.n;          // faceforward(N, I, Nref = N)  util.h:85-93
}
template <class W>      // W: the witness of the fast roots and normalisations (sbx_witness.h)
__device__ __forceinline__ void hit_sphere(v3 ro, v3 rd, const RtSphere& s, Hit& hit, W& w) {   // intersect.h:7-33
    const v3 rc = s.o - ro;
    const float radius2 = s.r * s.r;
    const float tca = dot(rc, rd);
    if (tca < 0.f) return;
    const float d2 = dot(rc, rc) - tca * tca;
    if (d2 > radius2) return;
    const float thc = w.sqrt(radius2 - d2);
    float t0 = tca - thc;
    const float t1 = tca + thc;
    if (t0 < 0.f) t0 = t1;
    if (t0 > hit.t) return;
    const v3 impact = ro + rd * t0;
    hit.t = t0;
    hit.mat = s.mat;
    hit.o = impact;
    const v3 dn = impact - s.o;
    hit.n = V3(div_by(dn.x, s.rr), div_by(dn.y, s.rr), div_by(dn.z, s.rr));     // (impact - origin) / radius, exact (sbx_math.h)
}
#ifndef RT_AXIS_PLANES
#define RT_AXIS_PLANES 1   // the six walls as three axis pairs, one division per axis (below); 0 = six hit_plane from the scene block
#endif
// THE CORNELL BOX'S WALLS AS AXIS PAIRS (round 5).  cornell_box.h:57-69 builds, in array order, the planes
//   0: n (0,-1,0) d 0    1: n (0,0,-1) d -2    2: n (0,0,1) d 2    3: n (0,1,0) d 4    4: n (1,0,0) d 2 (mat 2)    5: n (-1,0,0) d -2 (mat 3)
// (mat 1 for 0..3) for every frame — launch_raytracer checks the frame holds exactly these, else the generic kernel runs.  For a
// normal n = s e_c (s = +-1) and FINITE ro, rd with rd_c != 0, intersect_plane's two dot products are exact single terms: the
// other two products are zeros, and a zero added to a nonzero number changes nothing:
//   denom = dot(n, rd) = s rd_c         num = dot(P0 - ro, n) = s (d - ro_c)   (when that is nonzero)      t = num / denom = (d - ro_c) / rd_c
// (IEEE division is sign-symmetric).  `denom < 1e-6 -> no hit` admits at most ONE plane of each pair — the one rd_c points at — so a
// ray costs three divisions, not six, no normals or distances read from LDS and one hit point.  The reference's loop takes the
// planes in array order and a later plane replaces an earlier one at EQUAL t (`t > hit.t` is false), so the candidates are taken
// in that order: plane 0 (if rd_y < 0), the z plane, plane 3 (if rd_y > 0), the x plane.  hit.n = -n (faceforward: dot(n, rd) =
// denom > 0) keeps its negated zeros.  What the shortcut needs is what the witness already records for the normalisation that
// made rd (every component's square a normal number: finite, nonzero) plus a finite frame (checked on the host: then ro = eye or
// hit.o + dir * 1e-4 stays finite, t <= 1e8 + 10); a num of zero — its sign would be the generic sum's, not s * 0 — is recorded too
// and the wave re-runs with the generic IEEE kernel body like for any other record.
template <class W>
__device__ __forceinline__ void hit_walls(v3 ro, v3 rd, Hit& hit, W& w) {
    const float ty = ((rd.y > 0.f ? 4.f : 0.f) - ro.y) / rd.y;
    const float tz = ((rd.z > 0.f ? 2.f : -2.f) - ro.z) / rd.z;
    const float tx = ((rd.x > 0.f ? 2.f : -2.f) - ro.x) / rd.x;
    w.bad |= (ty == 0.f) || (tz == 0.f) || (tx == 0.f);          // a zero quotient here is a zero num (rd finite, nonzero)
    float t = hit.t;
    int ax = -1;
    const bool cy = !(abs_(rd.y) < 1e-6f) && !(ty < 0.f);
    if (cy && rd.y < 0.f && !(ty > t)) { t = ty; ax = 0; }
    if (!(abs_(rd.z) < 1e-6f) && !(tz < 0.f) && !(tz > t)) { t = tz; ax = 1; }
    if (cy && rd.y > 0.f && !(ty > t)) { t = ty; ax = 0; }
    if (!(abs_(rd.x) < 1e-6f) && !(tx < 0.f) && !(tx > t)) { t = tx; ax = 2; }
    if (ax >= 0) {
        hit.t = t;
        hit.mat = ax == 2 ? (rd.x > 0.f ? 2 : 3) : 1;
        hit.o = ro + rd * t;
        hit.n = V3(ax == 2 ? (rd.x > 0.f ? -1.f : 1.f) : -0.f, ax == 0 ? (rd.y > 0.f ? -1.f : 1.f) : -0.f, ax == 1 ? (rd.z > 0.f ? -1.f : 1.f) : -0.f);
    }
}
inline bool rt_walls_are_cornell(const FrameRaytracer& F) {
    static const float n[6][3] = {{0, -1, 0}, {0, 0, -1}, {0, 0, 1}, {0, 1, 0}, {1, 0, 0}, {-1, 0, 0}};
    static const float d[6] = {0.f, -2.f, 2.f, 4.f, 2.f, -2.f};
    static const int mat[6] = {1, 1, 1, 1, 2, 3};
    for (int i = 0; i < 6; ++i) {
        const RtPlane& p = F.planes[i];
        // (== on the components: a -0 in a normal would change hit.n's zeros)
        if (f2u(p.n.x) != f2u(n[i][0]) || f2u(p.n.y) != f2u(n[i][1]) || f2u(p.n.z) != f2u(n[i][2]) || f2u(p.d) != f2u(d[i]) || p.mat != mat[i]) return false;
    }
    return true;
}

template <bool WALLS, class W>
__device__ __forceinline__ Hit trace(const FrameRaytracer& F, v3 ro, v3 rd, int mat_to_ignore, W& w) {   // :70-86
    Hit hit;
    hit.t = (float)(1e8f + 1e1f); hit.mat = -1; hit.n = V3(0, 0, 0); hit.o = V3(0, 0, 0);   // no_hit def.h:78-83
    if (WALLS && W::fast) {
        hit_walls(ro, rd, hit, w);
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) hit_plane(ro, rd, rt_ld(F.planes[i]), hit);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
    {
        const RtSphere sp = rt_ld(F.spheres[i]);
        if (sp.mat != mat_to_ignore) hit_sphere(ro, rd, sp, hit, w);
    }
    return hit;
}
// get_material: linear scan; an id outside 0..7 yields the zero-initialised material (App. B5)
__device__ __forceinline__ RtMaterial material_of(const FrameRaytracer& F, int id) {
    RtMaterial m;
    m.base_color = V3(0, 0, 0); m.roughness = 0.f; m.ior = 0.f; m.reflectivity = 0.f;
    m.r0 = 1.f;                                                  // ior 0: ((1 - 0) / (1 + 0))^2
#if RT_LDS_FRAME
    if (id >= 0 && id < 8) m = rt_ld(F.mats[id]);               // one LDS read per member at the lane's own address
#else
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i == id) m = F.mats[i];
#endif
    return m;
}
template <class W>
__device__ __forceinline__ v3 cook_torrance(v3 V, v3 L, const Hit& hit, const RtMaterial& mat, W& w) {   // light.h:64-92
    const v3 H = w.normalize(L + V);
    const float NdotL = dot(hit.n, L), NdotH = dot(hit.n, H), NdotV = dot(hit.n, V), VdotH = dot(V, H);
    const float geo_a = (2.f * NdotH * NdotV) / VdotH;
    const float geo_b = (2.f * NdotH * NdotL) / VdotH;
    const float geo_term = fmin_(1.f, fmin_(geo_a, geo_b));
    const float rough_sq = mat.roughness * mat.roughness;
    const float rough_a = 1.f / (rough_sq * NdotH * NdotH * NdotH * NdotH);
    const float rough_exp = (NdotH * NdotH - 1.f) / (rough_sq * NdotH * NdotH);
    const float rough_term = rough_a * exp_(rough_exp);
    const float Fc = 1.f - VdotH;                                 // fresnel_factor(1, ior, VdotH) with R0 from the frame
    const float fresnel_term = mat.r0 + (1.f - mat.r0) * (Fc * Fc * Fc * Fc * Fc);
    const float specular = (geo_term * rough_term * fresnel_term) / (3.14159265359f * NdotV * NdotL);
    return fmax_(0.f, NdotL) * (specular + mat.base_color);
}
// (mat = material_of(hit.mat) and L = normalize(light - hit.o), light.h:18-27, come from the caller, which needs both again:
// the scene block's reads are volatile, so the compiler would not share them)
template <class W>
__device__ __forceinline__ v3 rt_illuminate(const FrameRaytracer& F, v3 eye, const Hit& hit, const RtMaterial& mat, v3 L, W& w) {   // :46-68
    if (hit.mat == 0) return rt_ld(F.mats[0].base_color);        // mat_debug: flat
    v3 accum = V3(.01f, .01f, .01f);                              // ambient_light light.h:16
    const v3 V = w.normalize(eye - hit.o);
    accum = accum + cook_torrance(V, L, hit, mat, w);
    return accum;
}

#ifndef RT_WITNESS
#define RT_WITNESS 1       // fast roots and normalisations with a recorded domain (sbx_witness.h): 0 = the IEEE forms only
#endif

// One pixel's colour — render :88-136 — with the roots / normalisations of witness `w`.  Fs: the scene block (LDS copy or F).
template <bool WALLS, class W>
__device__ __forceinline__ v3 rt_pixel(const FrameRaytracer& F, const FrameRaytracer& Fs, v2 pc, W& w) {
    const v3 eye = F.cam.eye;
    v3 ro = eye, rd = primary_dir(F.cam, pc, w);

    v3 color = V3(0, 0, 0), accum = V3(1, 1, 1);
    for (int i = 0; i < 2; ++i) {                                 // :96-133
        const Hit hit = trace<WALLS>(Fs, ro, rd, -1, w);
        if (hit.t >= 1e8f) {
            color = color + accum * V3(0, 0, 0);                  // background :13-16
            break;
        }
        const float f = fresnel_factor(1.f, 1.f, dot(hit.n, -rd));
        const RtMaterial mat = material_of(Fs, hit.mat);
        const v3 shadow_line = rt_ld(Fs.light) - hit.o;           // = illuminate's light vector (point light, light.h:18-27)
        const v3 shadow_dir = w.normalize(shadow_line);
        color = color + (1.f - f) * accum * rt_illuminate(Fs, eye, hit, mat, shadow_dir, w);   // primary origin on every bounce (:105)
        if (i == 0) {                                             // shadow ray :108-121
            const Hit sh = trace<WALLS>(Fs, hit.o + shadow_dir * 1e-4f, shadow_dir, 0, w);
            if (sh.t < w.length(shadow_line)) color = color * 0.1f;
        }
        if (mat.reflectivity > 0.f) {
            accum = accum * f;
            // reflect(hit.normal, ray.direction): arguments swapped in the reference (:127), kept
            const v3 refl = w.normalize(hit.n - 2.f * dot(rd, hit.n) * rd);
            ro = hit.o + refl * 1e-4f;
            rd = refl;
        } else {
            break;
        }
    }
    return color;
}

template <int WIT, bool WALLS = false>   // WIT: 0 IEEE forms, 1 witnessed fast forms, 2 the witness's test edge (sbx_set_variant 2); WALLS: hit_walls
__global__ void __launch_bounds__(WG_THREADS) k_raytracer(FrameRaytracer F, RowMap M, float* __restrict__ out) {
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();      // (the dispatch order's cost table, RowMap.cost)
#if RT_LDS_FRAME
    __shared__ FrameRaytracer Fs;
    lds_frame_fill<FrameRaytracer, WG_THREADS>(Fs);
#define RT_F Fs
#else
#define RT_F F
#endif
    const Pixel px = pixel_of_thread(M);
    if (!px.valid) return;
    const v2 pc = point_cam(F.cam, px.fx, px.fy);
    v3 color;
    if (WIT != 0) {
        Wit<true> w;
        if (WIT == 2) w.lo = 0x3F800000u;
        color = rt_pixel<WALLS>(F, RT_F, pc, w);
        if (__builtin_amdgcn_ballot_w64(w.bad) != 0ull) {      // some lane left the fast forms' proved domain: the IEEE forms
            Wit<false> w0;
            color = rt_pixel<false>(F, RT_F, pc, w0);
        }
    } else {
        Wit<false> w0;
        color = rt_pixel<false>(F, RT_F, pc, w0);
    }
    tile_cost_store(M, tl_t0);
    store_rgba(M, out, px.idx, to_srgb(color));
}

dim3 raytracer_grid(const RowMap& M) { return grid_for(M); }

void launch_raytracer(const FrameRaytracer& F, const RowMap& M, float* out, hipStream_t s, int variant) {
    // hit_walls' conditions on the frame: the six planes are cornell_box.h's and every number of the frame is finite
    bool walls = RT_AXIS_PLANES && rt_walls_are_cornell(F);
    for (int i = 0; i < 3; ++i) walls = walls && std::isfinite(F.spheres[i].o.x) && std::isfinite(F.spheres[i].o.y) && std::isfinite(F.spheres[i].o.z) && std::isfinite(F.spheres[i].r);
    const float chk[] = {F.cam.eye.x, F.cam.eye.y, F.cam.eye.z, F.light.x, F.light.y, F.light.z};
    for (float v : chk) walls = walls && std::isfinite(v);
    if (variant == 2 && walls) hipLaunchKernelGGL((k_raytracer<2, true>), grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);     // (the witness's test build
    else if (variant == 2) hipLaunchKernelGGL((k_raytracer<2, false>), grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);         //  keeps hit_walls' domain too)
    else if (variant == 1 || variant == 3) hipLaunchKernelGGL(k_raytracer<0>, grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
    else if (!walls) hipLaunchKernelGGL((k_raytracer<RT_WITNESS, false>), grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
    else hipLaunchKernelGGL((k_raytracer<RT_WITNESS, true>), grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
}

}  // namespace sbx
