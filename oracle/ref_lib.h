/* oracle/ref_lib.h — CPU restatement of the reference's shared shader libraries.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Written from the reference's
 * algorithm, function by function; each block cites the reference lines it follows
 * (paths relative to /root/reference/).  Float literals are binary32, as the
 * reference's C++ build compiles with -fsingle-precision-constant (src/Makefile:12).
 */
#ifndef SBX_REF_LIB_H
#define SBX_REF_LIB_H
#include "ovec.h"

namespace sbxref {

/* ---- src/def.h:51-83 : constants and core structs ------------------------------- */
static const float PI = 3.14159265359f;      /* def.h:51 */
static const float BIAS = 1e-4f;             /* def.h:57 */
static const float MAX_DIST = 1e8f;          /* def.h:77 */

struct ray_t { vec3 origin, direction; };                         /* def.h:53-56 */
struct sphere_t { vec3 origin; float radius; int material; };     /* def.h:59-63 */
struct plane_t { vec3 direction; float distance; int material; }; /* def.h:65-69 */
struct hit_t { float t; int material_id; vec3 normal; vec3 origin; }; /* def.h:71-76 */
static inline hit_t no_hit() {                                    /* def.h:78-83 */
    hit_t h; h.t = (float)(MAX_DIST + 1e1f); h.material_id = -1;
    h.normal = vec3(0, 0, 0); h.origin = vec3(0, 0, 0); return h;
}

/* ---- uniforms: src/uniform_buffer.h:26-36 (b0) ---------------------------------- */
struct uniforms_t {
    vec2 u_res;
    vec2 u_mouse;
    float u_time;
};
/* src/uniform_buffer.h:39-55 : APP_CLOUDS aux block, with its defaults */
struct clouds_aux_t {
    vec3 wind_dir = vec3(0, 0, .2f);
    vec3 sun_dir = vec3(0, 0, -1);
    vec3 sun_color = vec3(1.f, .7f, .55f);
    float sun_power = 8.f;
    int cld_march_steps = 100;
    int illum_march_steps = 6;
    float sigma_scattering = .15f;
    float cld_coverage = .535f;
    float cld_thick = 125.f;
    float atm_radius = 5000.f;
    float atm_ground_y = 4750.f;
};
/* src/uniform_buffer.h:56-60 : APP_SDF_AO aux block */
struct sdf_ao_aux_t {
    float fog_density = .1f;
    float fog_falloff = .5f;
};

/* ---- src/util.h ----------------------------------------------------------------- */
/* util.h:5-20 */
static inline ray_t get_primary_ray(vec3 cam_local_point, vec3 cam_origin, vec3 cam_look_at) {
    vec3 fwd = normalize(cam_look_at - cam_origin);
    vec3 up = vec3(0, 1, 0);
    vec3 right = cross(up, fwd);
    up = cross(fwd, right);
    ray_t r;
    r.origin = cam_origin;
    r.direction = normalize(fwd + up * cam_local_point.y + right * cam_local_point.x);
    return r;
}
/* util.h:24-33 */
static inline mat3 transpose(const mat3& m) {
    return mat3(m.c[0].x, m.c[1].x, m.c[2].x,
                m.c[0].y, m.c[1].y, m.c[2].y,
                m.c[0].z, m.c[1].z, m.c[2].z);
}
/* util.h:44-51 */
static inline mat3 rotate_around_z(float angle_degrees) {
    float a = m_radians(angle_degrees);
    float s = m_sin(a), c = m_cos(a);
    return mat3(c, -s, 0, s, c, 0, 0, 0, 1);
}
/* util.h:53-60 */
static inline mat3 rotate_around_y(float angle_degrees) {
    float a = m_radians(angle_degrees);
    float s = m_sin(a), c = m_cos(a);
    return mat3(c, 0, s, 0, 1, 0, -s, 0, c);
}
/* util.h:62-69 */
static inline mat3 rotate_around_x(float angle_degrees) {
    float a = m_radians(angle_degrees);
    float s = m_sin(a), c = m_cos(a);
    return mat3(1, 0, 0, 0, c, -s, 0, s, c);
}
/* util.h:72-77 ; p = 1/2.2 evaluated in binary32 */
static inline vec3 linear_to_srgb(vec3 color) {
    const float p = 1.f / 2.2f;
    return vec3(m_pow(color.x, p), m_pow(color.y, p), m_pow(color.z, p));
}
/* util.h:85-93 (C++-only definition) */
static inline vec3 faceforward(vec3 N, vec3 I, vec3 Nref) { return dot(Nref, I) < 0 ? N : -N; }
/* util.h:95-101 */
static inline float checkboard_pattern(vec2 pos, float scale) {
    vec2 pattern = vec2(m_floor(pos.x * scale), m_floor(pos.y * scale));
    return m_mod(pattern.x + pattern.y, 2.0f);
}
/* util.h:103-112 */
static inline float band(float start, float peak, float end, float t) {
    return m_smoothstep(start, peak, t) * (1.f - m_smoothstep(peak, end, t));
}

/* ---- src/util_optics.h ---------------------------------------------------------- */
/* util_optics.h:5-14 : Schlick */
static inline float fresnel_factor(float n1, float n2, float VdotH) {
    float Rn = (n1 - n2) / (n1 + n2);
    float R0 = Rn * Rn;
    float F = 1.f - VdotH;
    return R0 + (1.f - R0) * (F * F * F * F * F);
}
/* util_optics.h:17-22 */
static inline vec3 reflect(vec3 incident, vec3 normal) {
    return incident - 2.f * dot(normal, incident) * normal;
}

/* ---- src/sdf.h ------------------------------------------------------------------ */
static inline vec2 op_add(vec2 d1, vec2 d2) { return d1.x < d2.x ? d1 : d2; }   /* sdf.h:5-11 */
static inline float op_add(float d1, float d2) { return m_min(d1, d2); }        /* sdf.h:13-18 */
static inline float op_sub(float d1, float d2) { return m_max(d1, -d2); }       /* sdf.h:20-28 */
/* sdf.h:38-47 */
static inline float op_blend(float a, float b, float k) {
    float h = m_clamp(0.5f + 0.5f * (b - a) / k, 0.0f, 1.0f);
    return m_mix(b, a, h) - k * h * (1.0f - h);
}
static inline float sd_plane(vec3 p, vec3 n, float d) { return dot(n, p) + d; }  /* sdf.h:49-57 */
static inline float sd_sphere(vec3 p, float r) { return length(p) - r; }         /* sdf.h:59-65 */
/* sdf.h:67-73 */
static inline float sd_box(vec3 p, vec3 b) {
    return m_max(m_abs(p.x) - b.x, m_max(m_abs(p.y) - b.y, m_abs(p.z) - b.z));
}
/* sdf.h:75-83 */
static inline float sd_torus(vec3 p, float R, float r) {
    return length(vec2(length(p.xy()) - R, p.z)) - r;
}
/* sdf.h:85-93 */
static inline float sd_y_cylinder(vec3 p, float r, float h) {
    return m_max(length(p.xz()) - r, m_abs(p.y) - h / 2.f);
}
/* sdf.h:95-109 */
static inline float sd_cylinder(vec3 P, vec3 P0, vec3 P1, float R) {
    vec3 dir = normalize(P1 - P0);
    float dist = length(cross(dir, P - P0));
    float plane_1 = sd_plane(P, dir, length(P1));
    float plane_2 = sd_plane(P, -dir, -length(P0));
    return op_sub(op_sub(dist, plane_1), plane_2) - R;
}
/* sdf.h:114-119 */
static inline float det2(vec2 a, vec2 b) { return a.x * b.y - b.x * a.y; }
/* sdf.h:120-139 */
static inline vec3 sd_bezier_get_closest(vec2 b0, vec2 b1, vec2 b2) {
    float a = det2(b0, b2);
    float b = 2.0f * det2(b1, b0);
    float d = 2.0f * det2(b2, b1);
    float f = b * d - a * a;
    vec2 d21 = b2 - b1;
    vec2 d10 = b1 - b0;
    vec2 d20 = b2 - b0;
    vec2 gf = 2.0f * (b * d21 + d * d10 + a * d20);
    gf = vec2(gf.y, -gf.x);
    vec2 pp = (-f * gf) / dot(gf, gf);
    vec2 d0p = b0 - pp;
    float ap = det2(d0p, d20);
    float bp = 2.0f * det2(d10, d0p);
    float t = m_clamp((ap + bp) / (2.0f * a + b + d), 0.0f, 1.0f);
    vec2 q = vmix(vmix(b0, b1, t), vmix(b1, b2, t), t);
    return vec3(q.x, q.y, t);
}
/* sdf.h:140-159 */
static inline vec2 sd_bezier(vec3 a, vec3 b, vec3 c, vec3 p, float thickness) {
    vec3 w = normalize(cross(c - b, a - b));
    vec3 u = normalize(c - b);
    vec3 v = normalize(cross(w, u));
    vec2 a2 = vec2(dot(a - b, u), dot(a - b, v));
    vec2 b2 = vec2(0.f, 0.f);
    vec2 c2 = vec2(dot(c - b, u), dot(c - b, v));
    vec3 p3 = vec3(dot(p - b, u), dot(p - b, v), dot(p - b, w));
    vec3 cp = sd_bezier_get_closest(a2 - p3.xy(), b2 - p3.xy(), c2 - p3.xy());
    return vec2(0.85f * (m_sqrt(dot(cp.xy(), cp.xy()) + p3.z * p3.z) - thickness), cp.z);
}

/* sdf.h:30-36 */
static inline float op_intersect(float d1, float d2) { return m_max(d1, d2); }
/* sdf.h:162-171 */
static inline float sd_capsule(vec3 p, vec3 a, vec3 b, float r) {
    vec3 ab = b - a;
    float t = m_clamp(dot(p - a, ab) / dot(ab, ab), 0.f, 1.f);
    return length((ab * t + a) - p) - r;
}

/* ---- src/IK.h ------------------------------------------------------------------- */
/* IK.h:5-42 (the law-of-cosines branch, the one compiled) */
static inline vec3 ik_2_bone_centered_solver(vec3 goal, float L1, float L2) {
    float G = length(goal);
    float cos_theta = (L1 * L1 + G * G - L2 * L2) / (2.f * L1 * G);
    float sin_theta = m_sqrt(1.f - cos_theta * cos_theta);
    mat3 rot = mat3(cos_theta, -sin_theta, 0,
                    sin_theta, cos_theta, 0,
                    0, 0, 1.f);
    return mul(rot, normalize(goal) * L1);
}
/* IK.h:44-52 */
static inline vec3 ik_solver(vec3 start, vec3 goal, float bone_length_1, float bone_length_2) {
    return start + ik_2_bone_centered_solver(goal - start, bone_length_1, bone_length_2);
}

/* ---- src/intersect.h ------------------------------------------------------------ */
/* intersect.h:7-33 */
static inline void intersect_sphere(const ray_t& ray, const sphere_t& sphere, hit_t& hit) {
    vec3 rc = sphere.origin - ray.origin;
    float radius2 = sphere.radius * sphere.radius;
    float tca = dot(rc, ray.direction);
    if (tca < 0.f) return;
    float d2 = dot(rc, rc) - tca * tca;
    if (d2 > radius2) return;
    float thc = m_sqrt(radius2 - d2);
    float t0 = tca - thc;
    float t1 = tca + thc;
    if (t0 < 0.f) t0 = t1;
    if (t0 > hit.t) return;
    vec3 impact = ray.origin + ray.direction * t0;
    hit.t = t0;
    hit.material_id = sphere.material;
    hit.origin = impact;
    hit.normal = (impact - sphere.origin) / sphere.radius;
}
/* intersect.h:35-53 (no early-outs: t0 = tca - thc is taken whatever its sign, sqrt of a negative number is NaN and flows on) */
static inline void intersect_sphere_from_inside(const ray_t& ray, const sphere_t& sphere, hit_t& hit) {
    vec3 rc = sphere.origin - ray.origin;
    float radius2 = sphere.radius * sphere.radius;
    float tca = dot(rc, ray.direction);
    float d2 = dot(rc, rc) - tca * tca;
    float thc = m_sqrt(radius2 - d2);
    float t0 = tca - thc;
    vec3 impact = ray.origin + ray.direction * t0;
    hit.t = t0;
    hit.material_id = sphere.material;
    hit.origin = impact;
    hit.normal = (impact - sphere.origin) / sphere.radius;
}
/* intersect.h:61-77 */
static inline void intersect_plane(const ray_t& ray, const plane_t& p, hit_t& hit) {
    float denom = dot(p.direction, ray.direction);
    if (denom < 1e-6f) return;
    vec3 P0 = vec3(p.distance, p.distance, p.distance);
    float t = dot(P0 - ray.origin, p.direction) / denom;
    if (t < 0.f || t > hit.t) return;
    hit.t = t;
    hit.material_id = p.material;
    hit.origin = ray.origin + ray.direction * t;
    hit.normal = faceforward(p.direction, ray.direction, p.direction);
}

/* ---- src/volumetric.h ----------------------------------------------------------- */
/* volumetric.h:13-19 */
static inline float rayleigh_phase_func(float mu) { return 3.f * (1.f + mu * mu) / (16.f * PI); }
/* volumetric.h:27-33 ; hg_g is a macro supplied by the including app; note (4 + PI) */
static inline float henyey_greenstein_phase_func(float mu, float hg_g) {
    return (1.f - hg_g * hg_g) / ((4.f + PI) * m_pow(1.f + hg_g * hg_g - 2.f * hg_g * mu, 1.5f));
}
/* volumetric.h:47-68 */
struct volume_sampler_t {
    vec3 origin, pos;
    float height, transmittance;
    vec3 radiance;
    float alpha;
};
static inline volume_sampler_t construct_volume(vec3 origin) {
    volume_sampler_t v;
    v.origin = origin; v.pos = origin; v.height = 0.f; v.transmittance = 1.f;
    v.radiance = vec3(0, 0, 0); v.alpha = 0.f;
    return v;
}

/* ---- src/noise_iq.h ------------------------------------------------------------- */
static inline float hash(float n) { return m_fract(m_sin(n) * 753.5453123f); } /* noise_iq.h:5-9 */
/* noise_iq.h:11-29 (the `#if 1` branch) */
static inline float noise_iq(vec3 x) {
    vec3 p = vfloor(x);
    vec3 f = vfract(x);
    f = f * f * (vec3(3.0f, 3.0f, 3.0f) - 2.0f * f);
    float n = p.x + p.y * 157.0f + 113.0f * p.z;
    return m_mix(m_mix(m_mix(hash(n + 0.0f), hash(n + 1.0f), f.x),
                       m_mix(hash(n + 157.0f), hash(n + 158.0f), f.x), f.y),
                 m_mix(m_mix(hash(n + 113.0f), hash(n + 114.0f), f.x),
                       m_mix(hash(n + 270.0f), hash(n + 271.0f), f.x), f.y), f.z);
}

/* ---- src/fbm.h:6 : DECL_FBM_FUNC(name, octaves, basis) -------------------------- */
template <int OCTAVES, class Basis>
static inline float fbm_generic(vec3 pos, float lacunarity, float init_gain, float gain, Basis basis) {
    vec3 p = pos;
    float H = init_gain;
    float t = 0.f;
    for (int i = 0; i < OCTAVES; i++) {
        t += basis(p) * H;
        p *= lacunarity;
        H *= gain;
    }
    return t;
}

/* ---- src/app_clouds_best.h:460-552 : Ashima simplex noise as flattened into the stand-alone cloud shader ---- */
/* app_clouds_best.h:460-466 : x - floor(x * (1.0 / 289.0)) * 289.0 */
static inline float sn_mod289(float x) { return x - m_floor(x * (1.0f / 289.0f)) * 289.0f; }
/* :469-471 */
static inline float sn_permute(float x) { return sn_mod289(((x * 34.0f) + 1.0f) * x); }
/* :478-551.  vec4 quantities are arrays of 4; every vector operation is done component-wise in the order written. */
static inline float snoise(vec3 v) {
    const float Cx = 1.0f / 6.0f, Cy = 1.0f / 3.0f;                       /* :480 */
    /* first corner :484-485 */
    const float s = dot(v, vec3(Cy, Cy, Cy));
    vec3 i = vfloor(v + s);
    const float t = dot(i, vec3(Cx, Cx, Cx));
    vec3 x0 = (v - i) + t;
    /* other corners :488-491 */
    vec3 g = vec3(m_step(x0.y, x0.x), m_step(x0.z, x0.y), m_step(x0.x, x0.z));   /* step(x0.yzx, x0.xyz) */
    vec3 l = vec3(1.0f - g.x, 1.0f - g.y, 1.0f - g.z);
    vec3 i1 = vec3(m_min(g.x, l.z), m_min(g.y, l.x), m_min(g.z, l.y));            /* min(g.xyz, l.zxy) */
    vec3 i2 = vec3(m_max(g.x, l.z), m_max(g.y, l.x), m_max(g.z, l.y));
    /* :497-499 */
    vec3 x1 = (x0 - i1) + Cx;
    vec3 x2 = (x0 - i2) + Cy;
    vec3 x3 = x0 - vec3(0.5f, 0.5f, 0.5f);
    /* permutations :502-506 */
    i = vec3(sn_mod289(i.x), sn_mod289(i.y), sn_mod289(i.z));
    const float oz[4] = {0.0f, i1.z, i2.z, 1.0f}, oy[4] = {0.0f, i1.y, i2.y, 1.0f}, ox[4] = {0.0f, i1.x, i2.x, 1.0f};
    float p[4];
    for (int k = 0; k < 4; k++)
        p[k] = sn_permute((sn_permute((sn_permute(i.z + oz[k]) + i.y) + oy[k]) + i.x) + ox[k]);
    /* gradients :510-531 */
    const float n_ = 0.142857142857f;
    const float nsx = n_ * 2.0f - 0.0f, nsy = n_ * 0.5f - 1.0f, nsz = n_ * 1.0f - 0.0f;   /* n_ * D.wyz - D.xzx */
    float gx[4], gy[4], h[4];
    for (int k = 0; k < 4; k++) {
        const float j = p[k] - 49.0f * m_floor(p[k] * nsz * nsz);
        const float x_ = m_floor(j * nsz);
        const float y_ = m_floor(j - 7.0f * x_);
        gx[k] = x_ * nsx + nsy;
        gy[k] = y_ * nsx + nsy;
        h[k] = 1.0f - m_abs(gx[k]) - m_abs(gy[k]);
    }
    const float b0[4] = {gx[0], gx[1], gy[0], gy[1]}, b1[4] = {gx[2], gx[3], gy[2], gy[3]};
    float s0[4], s1[4], sh[4];
    for (int k = 0; k < 4; k++) {
        s0[k] = m_floor(b0[k]) * 2.0f + 1.0f;
        s1[k] = m_floor(b1[k]) * 2.0f + 1.0f;
        sh[k] = -m_step(h[k], 0.0f);
    }
    /* a0 = b0.xzyw + s0.xzyw*sh.xxyy ; a1 = b1.xzyw + s1.xzyw*sh.zzww  :533-534 */
    const float a0[4] = {b0[0] + s0[0] * sh[0], b0[2] + s0[2] * sh[0], b0[1] + s0[1] * sh[1], b0[3] + s0[3] * sh[1]};
    const float a1[4] = {b1[0] + s1[0] * sh[2], b1[2] + s1[2] * sh[2], b1[1] + s1[1] * sh[3], b1[3] + s1[3] * sh[3]};
    vec3 p0 = vec3(a0[0], a0[1], h[0]), p1 = vec3(a0[2], a0[3], h[1]);
    vec3 p2 = vec3(a1[0], a1[1], h[2]), p3 = vec3(a1[2], a1[3], h[3]);
    /* normalise gradients :473-476, 542-546 */
    p0 *= 1.79284291400159f - 0.85373472095314f * dot(p0, p0);
    p1 *= 1.79284291400159f - 0.85373472095314f * dot(p1, p1);
    p2 *= 1.79284291400159f - 0.85373472095314f * dot(p2, p2);
    p3 *= 1.79284291400159f - 0.85373472095314f * dot(p3, p3);
    /* mix :549-551 */
    float m[4] = {m_max(0.6f - dot(x0, x0), 0.0f), m_max(0.6f - dot(x1, x1), 0.0f),
                  m_max(0.6f - dot(x2, x2), 0.0f), m_max(0.6f - dot(x3, x3), 0.0f)};
    for (int k = 0; k < 4; k++) m[k] = m[k] * m[k];
    const float d[4] = {dot(p0, x0), dot(p1, x1), dot(p2, x2), dot(p3, x3)};
    return 42.0f * ((((m[0] * m[0]) * d[0] + (m[1] * m[1]) * d[1]) + (m[2] * m[2]) * d[2]) + (m[3] * m[3]) * d[3]);
}

/* ---- src/noise_worley.h --------------------------------------------------------- */
/* noise_worley.h:5-17 */
static inline vec3 hash_w(vec3 x) {
    vec3 xx = vec3(dot(x, vec3(127.1f, 311.7f, 74.7f)),
                   dot(x, vec3(269.5f, 183.3f, 246.1f)),
                   dot(x, vec3(113.5f, 271.9f, 124.6f)));
    return vec3(m_fract(m_sin(xx.x) * 43758.5453123f),
                m_fract(m_sin(xx.y) * 43758.5453123f),
                m_fract(m_sin(xx.z) * 43758.5453123f));
}
/* noise_worley.h:20-51 : returns (sqrt F1, sqrt F2, |cell id|) */
static inline vec3 noise_w(vec3 pos, float domain_repeat) {
    vec3 x = pos * domain_repeat;
    vec3 p = vfloor(x);
    vec3 f = vfract(x);
    float id = 0.0f;
    vec2 res = vec2(100.0f, 100.0f);
    for (int k = -1; k <= 1; k++)
        for (int j = -1; j <= 1; j++)
            for (int i = -1; i <= 1; i++) {
                vec3 b = vec3((float)i, (float)j, (float)k);
                vec3 pb = p + b;
                vec3 cell = vec3(m_mod(pb.x, domain_repeat), m_mod(pb.y, domain_repeat), m_mod(pb.z, domain_repeat));
                vec3 r = b - f + hash_w(cell);
                float d = dot(r, r);
                if (d < res.x) {
                    id = dot(p + b, vec3(1.0f, 57.0f, 113.0f));
                    res = vec2(d, res.x);
                } else if (d < res.y) {
                    res.y = d;
                }
            }
    return vec3(m_sqrt(res.x), m_sqrt(res.y), m_abs(id));
}
/* src/fbm.h:8 DECL_FBM_FUNC_TILE instantiated as in util/ddsvolgen/src/ddsvolgen.cpp:52:
 * basis = (1 - (noise_w(p, L).r + .25)): p is never scaled, the octave scale is the repeat L */
static inline float fbm_worley_tile(vec3 pos, float lacunarity, float init_gain, float gain) {
    vec3 p = pos;
    float H = init_gain;
    float L = lacunarity;
    float t = 0.f;
    for (int i = 0; i < 4; i++) {
        t += (1.f - (noise_w(p, L).x + .25f)) * H;
        L *= lacunarity;
        H *= gain;
    }
    return t;
}

} /* namespace sbxref */
#endif
