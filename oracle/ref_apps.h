/* oracle/ref_apps.h — CPU restatement of the reference's apps and of mainImage().
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * PINNING: the reference has no tests, golden vectors or fixtures, and its own C++ program cannot be built
 * in this image (VML + SDL harness are external and absent), so this restatement is pinned against the known
 * answers of SURVEY.md Appendix C (tests/test_oracle_kat.py) — values the survey obtained by compiling the
 * reference headers verbatim over glibc.  Against the author's actual binary: PARITY UNPINNED.  APP_VINYL and
 * app_clouds_best.h have no Appendix C values at all (review + libm comparison only).
 *
 * One struct per APP_* project define (/root/reference/README.md:11-22).  An app object
 * is constructed afresh for every pixel, which is how this oracle implements the GLSL
 * per-invocation meaning of the reference's `_mutable` globals (src/def.h:18; SURVEY.md
 * App. B1): EGG's `depth` and ATMOSPHERE's `sun_dir` start from their initialisers in
 * every pixel.  Everything is recomputed per pixel/per call exactly where the reference
 * computes it (no hoisting) — the HIP kernels are the ones that restructure.
 */
#ifndef SBX_REF_APPS_H
#define SBX_REF_APPS_H
#include "ref_lib.h"

namespace sbxref {

static inline vec3 operator+(float s, vec3 a) { return vec3(s + a.x, s + a.y, s + a.z); }
static inline vec3 operator-(vec3 a, float s) { return vec3(a.x - s, a.y - s, a.z - s); }

/* ---- src/main.h:6-53 -------------------------------------------------------------- */
template <class App>
static inline vec4 main_image(App& app, vec2 fragCoord) {
    const uniforms_t& U = app.U;
    vec2 aspect_ratio = vec2(U.u_res.x / U.u_res.y, 1);                    /* main.h:33 */
    vec3 eye, look_at;
    app.setup_camera(eye, look_at);                                        /* main.h:36 */
    app.setup_scene();                                                     /* main.h:38 */
    vec2 point_ndc = fragCoord / U.u_res;                                  /* main.h:40 */
    vec3 point_cam = vec3((2.0f * point_ndc - 1.0f) * aspect_ratio * app.fov(), -1.0f); /* :44-46 */
    ray_t ray = get_primary_ray(point_cam, eye, look_at);                  /* main.h:48 */
    vec3 color = app.render(ray, point_cam);                               /* main.h:50 */
    return vec4(linear_to_srgb(color), 1);                                 /* main.h:52 */
}

/* =================================================================================== */
/* APP_EGG — src/app_egg.h                                                              */
/* =================================================================================== */
struct AppEgg {
    uniforms_t U;
    float depth = -MAX_DIST;                                  /* app_egg.h:188 */
    enum { mat_debug = 0, mat_egg = 1, mat_bike = 2, mat_ground = 3 }; /* app_egg.h:17-20 */

    float fov() const { return 1.f; }                          /* app_egg.h:253 */
    vec3 background(const ray_t&) const { return vec3(.1f, .1f, .7f); }  /* :9-12 */
    void setup_scene() {}
    void setup_camera(vec3& eye, vec3& look_at) const {         /* app_egg.h:23-27 */
        eye = vec3(.0f, .25f, 5.25f);
        look_at = vec3(.0f, .25f, .0f);
    }
    vec3 illuminate(const hit_t& hit) const {                   /* app_egg.h:29-35 */
        if (hit.material_id == mat_ground) return vec3(13.f / 255.f, 104.f / 255.f, 0.f / 255.f);
        if (hit.material_id == mat_egg) return vec3(0.9f, 0.95f, 0.95f);
        if (hit.material_id == mat_bike) return vec3(.2f, .2f, .2f);
        return vec3(1, 1, 1);
    }

    /* app_egg.h:38-144 (BEZIER defined, `#if 1` egg) */
    vec2 sdf(vec3 P) const {
        const float u_time = U.u_time;
        vec3 p = mul(rotate_around_y(u_time * -100.0f), P) - vec3(0, 0.5f, 3.5f);
        float material = (float)mat_egg;

        float egg_y = 0.65f;
        float egg_m = sd_sphere(p - vec3(0, egg_y, 0), 0.475f);
        float egg_b = sd_sphere(p - vec3(0, egg_y - 0.45f, 0), 0.25f);
        float egg_t = sd_sphere(p - vec3(0, egg_y + 0.45f, 0), 0.25f);
        float egg_1 = op_blend(egg_m, egg_b, .5f);
        float egg_2 = op_blend(egg_1, egg_t, .5f);
        vec2 egg = vec2(egg_2, material);

        vec3 wheel_pos = vec3(0, 1.2f, 0);
        float pedal_radius = 0.3f;
        float pedal_speed = 400.f;
        float pedal_off = 0.2f;

        mat3 rot_z = rotate_around_z(-u_time * pedal_speed);
        vec3 left_foot_pos = wheel_pos + mul(rot_z, vec3(0.f, pedal_radius, pedal_off));
        rot_z = rotate_around_z(-u_time * pedal_speed);
        vec3 right_foot_pos = wheel_pos + mul(rot_z, vec3(0.f, -pedal_radius, -pedal_off));

        vec3 side = vec3(0, 0, pedal_off);
        float femur = 0.8f;
        float tibia = 0.75f;
        float thick = .05f;

        vec3 pelvis = vec3(0, 0.f, 0) + side;
        vec3 knee_l = ik_solver(pelvis, left_foot_pos, femur, tibia);
        pelvis = vec3(0, 0.f, 0) - side;
        vec3 knee_r = ik_solver(pelvis, right_foot_pos, femur, tibia);

        vec2 legs = op_add(
            vec2(sd_bezier(-(vec3(0.f, 0.f, 0.f) + side), -knee_l, -left_foot_pos, p, thick).x, material),
            vec2(sd_bezier(-(vec3(0.f, 0.f, 0.f) - side), -knee_r, -right_foot_pos, p, thick).x, material));

        vec3 left_toe = normalize(vec3(left_foot_pos.y - knee_l.y, knee_l.x - left_foot_pos.x, 0));
        vec2 left_foot = vec2(sd_cylinder(p + left_foot_pos, vec3(0.f, 0.f, 0.f), left_toe / 8.f, thick), material);
        vec3 right_toe = normalize(vec3(right_foot_pos.y - knee_r.y, knee_r.x - right_foot_pos.x, 0));
        vec2 right_foot = vec2(sd_cylinder(p + right_foot_pos, vec3(0.f, 0.f, 0.f), right_toe / 8.f, thick), material);
        vec2 feet = op_add(left_foot, right_foot);

        vec2 bike = vec2(sd_torus(p + wheel_pos, 1.f, .03f), (float)mat_bike);
        vec2 ground = vec2(sd_plane(P, vec3(0.f, 1.f, 0.f), wheel_pos.y + 0.5f), (float)mat_ground);

        vec2 _1 = op_add(feet, bike);
        vec2 _2 = op_add(egg, _1);
        vec2 _3 = op_add(legs, _2);
        return op_add(ground, _3);
    }

    /* app_egg.h:161-186 */
    float shadowmarch(const ray_t& ray) const {
        const int steps = 20;
        const float end = 10.f;
        const float penumbra_factor = 15.f;
        const float darkest = 0.1f;
        float t = 0.f;
        float umbra = 1.f;
        for (int i = 0; i < steps; i++) {
            vec3 p = ray.origin + ray.direction * t;
            vec2 d = sdf(p);
            if (t > end) break;
            if (d.x < 0.001f) return darkest;
            t += d.x;
            umbra = m_min(umbra, penumbra_factor * d.x / t);
        }
        return umbra;
    }

    /* app_egg.h:190-231 */
    vec3 render_scene(const ray_t& ray) {
        const int steps = 80;
        const float end = 15.f;
        float t = 0.f;
        for (int i = 0; i < steps; i++) {
            vec3 p = ray.origin + ray.direction * t;
            vec2 d = sdf(p);
            if (t > end) break;
            if (d.x < 0.001f) {
                hit_t h;
                h.t = t; h.material_id = (int)d.y; h.normal = vec3(0, 0, 0); h.origin = p;
                if (h.material_id == mat_egg || h.material_id == mat_bike) {
                    depth = m_max(depth, p.z);
                }
                float s = 1.f;
                if ((int)d.y == mat_ground) {
                    vec3 sh_dir = vec3(0, 1, 1);
                    ray_t sh_ray;
                    sh_ray.origin = p + sh_dir * 0.05f;
                    sh_ray.direction = sh_dir;
                    s = shadowmarch(sh_ray);
                }
                return illuminate(h) * s;
            }
            t += d.x;
        }
        return background(ray);
    }

    /* app_egg.h:233-251 */
    vec3 render(const ray_t& eye, vec3 point_cam) {
        vec3 final_color = render_scene(eye);
        const float BAR_SEPARATION = 0.6f, BAR_WIDTH = 0.05f, BAR_DEPTH = 1.f;
        const vec3 BAR_COLOR = vec3(.6f, .6f, .6f);
        float bar_factor = 1.0f - m_smoothstep(0.0f, 0.01f, m_abs((m_abs(point_cam.x) - BAR_SEPARATION)) - BAR_WIDTH);
        float depth_factor = 1.f - m_step(BAR_DEPTH, depth);
        final_color = vmix(final_color, BAR_COLOR, bar_factor * depth_factor);
        return vabs(final_color);
    }
};

/* =================================================================================== */
/* APP_CLOUDS — src/app_clouds.h (SKY_SPHERE and USE_NOISE_TEX undefined, :8-9)         */
/* =================================================================================== */
/* The two 3-D noise textures of the USE_NOISE_TEX build (app_clouds.h:51-56: u_tex_noise = t1, u_tex_noise_2 = t2),
 * as util/ddsvolgen bakes them (ddsvolgen.cpp:101-117: RGBA32F, size^3, x fastest) and hlsltoy binds them with a
 * MIN_MAG_MIP_LINEAR / WRAP sampler (util/hlsltoy/src/hlsltoy.cpp:227-249, 437). */
struct noise_tex_t { const float* rgba = nullptr; int size = 0; };

/* SampleLevel(u_sampler0, pos, 0).r as the sbx spec fixes it (the reference leaves the filter to the GPU):
 * texel centres at (i + .5) / size, WRAP addressing, trilinear blend in the mix(x), mix(y), mix(z) order with full
 * binary32 weights.  PARITY UNPINNED against real D3D11 texture hardware (which quantises the weights to 8 bits). */
static inline void tex_axis(float c, int size, int& i0, int& i1, float& f) {
    const float fsize = (float)size;
    const float u = c * fsize - .5f;
    const float fl = m_floor(u);
    f = u - fl;
    float m = m_mod(fl, fsize);
    if (m < 0.f) m += fsize;
    if (m >= fsize) m -= fsize;
    const int i = (m >= 0.f && m < fsize) ? (int)m : 0;
    i0 = i;
    i1 = (i + 1 == size) ? 0 : i + 1;
}
static inline float tex3d_sample_r(const noise_tex_t& T, vec3 p) {
    int x0, x1, y0, y1, z0, z1;
    float fx, fy, fz;
    tex_axis(p.x, T.size, x0, x1, fx);
    tex_axis(p.y, T.size, y0, y1, fy);
    tex_axis(p.z, T.size, z0, z1, fz);
    auto at = [&](int x, int y, int z) { return T.rgba[(((size_t)z * T.size + y) * T.size + x) * 4]; };
    return m_mix(m_mix(m_mix(at(x0, y0, z0), at(x1, y0, z0), fx), m_mix(at(x0, y1, z0), at(x1, y1, z0), fx), fy),
                 m_mix(m_mix(at(x0, y0, z1), at(x1, y0, z1), fx), m_mix(at(x0, y1, z1), at(x1, y1, z1), fx), fy), fz);
}
/* util.h:127-138 */
static inline float remap(float original_value, float original_min, float original_max, float new_min, float new_max) {
    return new_min + (((original_value - original_min) / (original_max - original_min)) * (new_max - new_min));
}

struct AppClouds {
    uniforms_t U;
    clouds_aux_t A;
    /* USE_NOISE_TEX build (app_clouds.h:9): both set -> density_func samples the volumes instead of the fBm */
    noise_tex_t tex_noise, tex_noise_2;
    bool use_noise_tex() const { return tex_noise.rgba != nullptr && tex_noise_2.rgba != nullptr; }
    /* SKY_SPHERE build (app_clouds.h:8,14-19,154-162): the march starts on a sphere around the viewer and runs along the view ray */
    bool sky_sphere = false;
    sphere_t atmosphere() const {                               /* app_clouds.h:15-17 */
        sphere_t s; s.origin = vec3(0, A.atm_ground_y, 0); s.radius = A.atm_radius; s.material = 0;
        return s;
    }
    static constexpr float hg_g = .2f;                         /* app_clouds.h:5 */
    float cld_noise_factor() const {                            /* app_clouds.h:18 / :20 */
        return sky_sphere ? ((1.f / atmosphere().radius) * 10.f) : .001f;
    }

    float fov() const { return 1.f; }                          /* app_clouds.h:219 */
    void setup_scene() {}
    void setup_camera(vec3& eye, vec3& look_at) const {         /* app_clouds.h:23-30 */
        eye = vec3(0, -.5f, 0);
        float angle = U.u_mouse.x * .5f;
        look_at = mul(rotate_around_y(angle), vec3(0, 0, -1));
    }
    /* app_clouds.h:36-46 */
    vec3 render_sky_color(vec3 eye_dir) const {
        float sun_amount = m_max(dot(eye_dir, A.sun_dir), 0.f);
        vec3 sky = vmix(vec3(.0f, .1f, .4f), vec3(.3f, .6f, .8f), 1.0f - eye_dir.y);
        sky += A.sun_color * m_min(m_pow(sun_amount, 1500.0f) * 5.0f, 1.0f);
        sky += A.sun_color * m_min(m_pow(sun_amount, 10.0f) * .6f, 1.0f);
        return vabs(sky);
    }
    /* app_clouds.h:59 : DECL_FBM_FUNC(fbm, 4, noise_iq(p)) */
    static float fbm(vec3 pos, float lacunarity, float init_gain, float gain) {
        return fbm_generic<4>(pos, lacunarity, init_gain, gain, [](vec3 p) { return noise_iq(p); });
    }
    /* app_clouds.h:62-86 */
    float density_func(vec3 pos_in, float height) const {
        vec3 pos = pos_in * cld_noise_factor();
        float shape;
        if (use_noise_tex()) {                                  /* #ifdef USE_NOISE_TEX :69-70, :74-81 */
            shape = tex3d_sample_r(tex_noise, pos);
            float w = tex3d_sample_r(tex_noise_2, pos);
            float ww = m_mix(w, 1.f - w, height);
            shape = remap(shape, ww * .7f, 1.f, 0.f, 1.f);
        } else {
            shape = fbm(pos * 2.03f, 2.64f, .5f, .5f);          /* :72 */
        }
        const float cov = 1.f - A.cld_coverage;
        return shape * m_smoothstep(cov, cov + .0135f, shape);
    }
    /* app_clouds.h:91-123 */
    float illuminate_volume(vec3 origin, float /*height*/, vec3 V, vec3 L) const {
        const float dt = A.cld_thick / (float)A.cld_march_steps;
        volume_sampler_t vol = construct_volume(origin);
        vol.pos += L * dt;
        for (int i = 0; i < A.illum_march_steps; i++) {
            vol.height = (float)i / (float)A.illum_march_steps;
            float density = density_func(vol.pos, vol.height);
            vol.transmittance *= m_exp(-density * A.sigma_scattering * dt);
            vol.pos += L * dt;
        }
        float luminance = vol.transmittance;
        return luminance * A.sun_power * henyey_greenstein_phase_func(m_clamp(dot(L, V), 0.f, 1.f), hg_g);
    }
    /* app_clouds.h:125-148 */
    void integrate_volume(volume_sampler_t& vol, vec3 V, vec3 L, float density, float dt) const {
        if (density < .005f) return;
        float T_i = m_exp(-density * A.sigma_scattering * dt);
        vol.transmittance *= T_i;
        vol.radiance += (density * A.sigma_scattering) * illuminate_volume(vol.pos, vol.height, V, L) * vol.transmittance * dt;
        vol.alpha += (1.f - T_i) * (1.f - vol.alpha);
    }
    /* app_clouds.h:153-202 */
    vec4 render_clouds(const ray_t& eye) const {
        vec3 projection, origin;
        if (sky_sphere) {                                       /* #ifdef SKY_SPHERE :154-162 */
            hit_t hit = no_hit();
            const sphere_t atm = atmosphere();
            intersect_sphere_from_inside(eye, atm, hit);
            projection = eye.direction;
            origin = hit.origin;
            mat3 rot = rotate_around_x(U.u_time);
            origin = mul(rot, origin - atm.origin);
        } else {                                                /* #else :163-167 */
            projection = eye.direction / eye.direction.y;
            origin = eye.origin + projection * 150.f;
            origin += A.wind_dir * U.u_time * (1.f / cld_noise_factor());
        }
        volume_sampler_t cloud = construct_volume(origin);
        float t = 0.f;
        const float dt = A.cld_thick / (float)A.cld_march_steps;
        for (int i = 0; i < A.cld_march_steps; i++) {
            cloud.height = (float)i / (float)A.cld_march_steps;
            cloud.pos = cloud.origin + t * projection;
            t += dt;
            float density = density_func(cloud.pos, cloud.height);
            integrate_volume(cloud, eye.direction, A.sun_dir, density, dt);
            if (cloud.alpha > .999f) break;
        }
        float cutoff = dot(eye.direction, vec3(0, 1, 0));
        return vec4(cloud.radiance, cloud.alpha * m_smoothstep(.0f, .2f, cutoff));
    }
    /* app_clouds.h:204-218 */
    vec3 render(const ray_t& eye_ray, vec3 /*point_cam*/) const {
        vec3 sky = render_sky_color(eye_ray.direction);
        if (dot(eye_ray.direction, vec3(0, 1, 0)) < 0.05f) return sky;
        vec4 cld = render_clouds(eye_ray);
        vec3 col = vmix(sky, cld.rgb(), cld.w);
        return vabs(col);
    }
};

/* =================================================================================== */
/* APP_RAYTRACER — src/app_raytracer.h + material.h, light.h, cornell_box.h             */
/* =================================================================================== */
struct material_t {                                            /* material.h:5-12 */
    vec3 base_color;
    float metallic = 0, roughness = 0, ior = 0, reflectivity = 0, translucency = 0;
};
struct light_t { int type = 0; vec3 L; vec3 color; };           /* light.h:8-12 */

struct AppRaytracer {
    uniforms_t U;
    enum { num_materials = 8, mat_invalid = -1, mat_debug = 0 }; /* material.h:14-16 */
    enum { LIGHT_POINT = 1, LIGHT_DIR = 2 };                     /* light.h:5-6 */
    enum { cb_mat_white = 1, cb_mat_red, cb_mat_blue, cb_mat_reflect, cb_mat_refract, cb_mat_green };
    enum { cb_plane_ground = 0, cb_plane_behind, cb_plane_front, cb_plane_ceiling, cb_plane_left, cb_plane_right };
    enum { cb_sphere_light = 0, cb_sphere_left, cb_sphere_right };
    static constexpr float cb_plane_dist = 2.f;                 /* cornell_box.h:62 */

    material_t materials[num_materials];                        /* material.h:17 (zero-initialised, App. B5) */
    light_t lights[8];                                          /* light.h:14 */
    vec3 ambient_light = vec3(.01f, .01f, .01f);                /* light.h:16 */
    plane_t cb_planes[6];                                       /* cornell_box.h:9 */
    sphere_t cb_spheres[3];                                     /* cornell_box.h:12 */

    float fov() const { return m_tan(m_radians(30.f)); }        /* app_raytracer.h:138 */
    vec3 background(const ray_t&) const { return vec3(0, 0, 0); } /* :13-16 */

    /* material.h:19-36 (the non-HLSL loop form) */
    material_t get_material(int index) const {
        material_t mat;
        for (int i = 0; i < num_materials; ++i) {
            if (i == index) { mat = materials[i]; break; }
        }
        return mat;
    }
    static void setup_material(material_t& mat, vec3 diffuse, float metallic, float roughness) { /* cornell_box.h:14-26 */
        mat.base_color = diffuse; mat.metallic = metallic; mat.roughness = roughness;
        mat.ior = 1.f; mat.reflectivity = 0.f; mat.translucency = 0.f;
    }
    static void setup_plane(plane_t& p, vec3 n, float d, int mat_id) { /* cornell_box.h:28-37 */
        p.direction = n; p.distance = d; p.material = mat_id;
    }
    /* cornell_box.h:39-87 */
    void setup_cornell_box() {
        setup_material(materials[cb_mat_white], vec3(0.7913f, 0.7913f, 0.7913f), .0f, .5f);
        setup_material(materials[cb_mat_red], vec3(0.6795f, 0.0612f, 0.0529f), 0.f, .5f);
        setup_material(materials[cb_mat_blue], vec3(0.1878f, 0.1274f, 0.4287f), 0.f, .5f);
        setup_material(materials[cb_mat_reflect], vec3(0.95f, 0.64f, 0.54f), 1.f, .1f);
        materials[cb_mat_reflect].reflectivity = 1.f;
        setup_material(materials[cb_mat_refract], vec3(1.f, 0.77f, 0.345f), 1.f, .05f);
        materials[cb_mat_refract].reflectivity = 1.f;
        materials[cb_mat_refract].translucency = 0.f;
        materials[cb_mat_refract].ior = 1.333f;

        setup_plane(cb_planes[cb_plane_ground], vec3(0, -1, 0), 0.f, cb_mat_white);
        setup_plane(cb_planes[cb_plane_ceiling], vec3(0, 1, 0), 2.f * cb_plane_dist, cb_mat_white);
        setup_plane(cb_planes[cb_plane_behind], vec3(0, 0, -1), -cb_plane_dist, cb_mat_white);
        setup_plane(cb_planes[cb_plane_front], vec3(0, 0, 1), cb_plane_dist, cb_mat_white);
        setup_plane(cb_planes[cb_plane_left], vec3(1, 0, 0), cb_plane_dist, cb_mat_red);
        setup_plane(cb_planes[cb_plane_right], vec3(-1, 0, 0), -cb_plane_dist, cb_mat_blue);

        cb_spheres[cb_sphere_light].origin = vec3(0, 2.5f * cb_plane_dist + 0.4f, 0);
        cb_spheres[cb_sphere_light].radius = 1.5f;
        cb_spheres[cb_sphere_light].material = mat_debug;
        cb_spheres[cb_sphere_left].origin = vec3(0.75f, 1, -0.75f);
        cb_spheres[cb_sphere_left].radius = 0.75f;
        cb_spheres[cb_sphere_left].material = cb_mat_reflect;
        cb_spheres[cb_sphere_right].origin = vec3(-0.75f, 0.75f, 0.75f);
        cb_spheres[cb_sphere_right].radius = 0.75f;
        cb_spheres[cb_sphere_right].material = cb_mat_refract;

        lights[0].type = LIGHT_POINT;
        lights[0].L = vec3(0, 2.f * cb_plane_dist - 0.2f, 0);
        lights[0].color = vec3(1.f, 1.f, 1.f);
    }
    /* app_raytracer.h:18-36 */
    void setup_scene() {
        materials[mat_debug].base_color = vec3(1.f, 1.f, 1.f);
        materials[mat_debug].metallic = 0.f;
        materials[mat_debug].roughness = 0.f;
        materials[mat_debug].ior = 1.f;
        materials[mat_debug].reflectivity = 0.f;
        materials[mat_debug].translucency = 0.f;
        setup_cornell_box();
        float _sin = m_sin(U.u_time);
        float _cos = m_cos(U.u_time);
        cb_spheres[cb_sphere_left].origin += vec3(0, m_abs(_sin), _cos + 1.f);
        cb_spheres[cb_sphere_right].origin.z = 0.f;
        lights[0].L.z = 1.5f;
    }
    /* app_raytracer.h:38-44 */
    void setup_camera(vec3& eye, vec3& look_at) const {
        vec2 mouse = U.u_mouse.x < BIAS ? vec2(0, 0) : 2.f * (U.u_res / U.u_mouse) - 1.f;
        mat3 rot_y = rotate_around_y(mouse.x * 30.f);
        eye = mul(rot_y, vec3(0, cb_plane_dist, 2.333f * cb_plane_dist));
        look_at = vec3(0, cb_plane_dist, 0);
    }
    /* light.h:18-27 */
    static vec3 get_light_direction(const light_t& light, const hit_t& P) {
        if (light.type == LIGHT_DIR) return light.L;
        return normalize(light.L - P.origin);
    }
    /* light.h:64-92 */
    static vec3 illum_cook_torrance(vec3 V, vec3 L, const hit_t& hit, const material_t& mat) {
        vec3 H = normalize(L + V);
        float NdotL = dot(hit.normal, L);
        float NdotH = dot(hit.normal, H);
        float NdotV = dot(hit.normal, V);
        float VdotH = dot(V, H);
        float geo_a = (2.f * NdotH * NdotV) / VdotH;
        float geo_b = (2.f * NdotH * NdotL) / VdotH;
        float geo_term = m_min(1.f, m_min(geo_a, geo_b));
        float rough_sq = mat.roughness * mat.roughness;
        float rough_a = 1.f / (rough_sq * NdotH * NdotH * NdotH * NdotH);
        float rough_exp = (NdotH * NdotH - 1.f) / (rough_sq * NdotH * NdotH);
        float rough_term = rough_a * m_exp(rough_exp);
        float fresnel_term = fresnel_factor(1.f, mat.ior, VdotH);
        float specular = (geo_term * rough_term * fresnel_term) / (PI * NdotV * NdotL);
        return m_max(0.f, NdotL) * (specular + mat.base_color);
    }
    /* app_raytracer.h:46-68 */
    vec3 illuminate(vec3 eye, const hit_t& hit) const {
        material_t mat = get_material(hit.material_id);
        if (hit.material_id == mat_debug) return materials[mat_debug].base_color;
        vec3 accum = ambient_light;
        vec3 V = normalize(eye - hit.origin);
        vec3 L = get_light_direction(lights[0], hit);
        accum += illum_cook_torrance(V, L, hit, mat);
        return accum;
    }
    /* app_raytracer.h:70-86 */
    hit_t raytrace_iteration(const ray_t& ray, int mat_to_ignore) const {
        hit_t hit = no_hit();
        for (int i = 0; i < 6; ++i) intersect_plane(ray, cb_planes[i], hit);
        for (int i = 0; i < 3; ++i) {
            if (cb_spheres[i].material != mat_to_ignore) intersect_sphere(ray, cb_spheres[i], hit);
        }
        return hit;
    }
    /* app_raytracer.h:88-136 */
    vec3 render(const ray_t& primary_ray, vec3 /*point_cam*/) const {
        vec3 color = vec3(0, 0, 0);
        vec3 accum = vec3(1, 1, 1);
        ray_t ray = primary_ray;
        for (int i = 0; i < 2; i++) {
            hit_t hit = raytrace_iteration(ray, mat_invalid);
            if (hit.t >= MAX_DIST) {
                color += accum * background(ray);
                break;
            }
            float f = fresnel_factor(1.f, 1.f, dot(hit.normal, -ray.direction));
            color += (1.f - f) * accum * illuminate(primary_ray.origin, hit);
            if (i == 0) {
                vec3 shadow_line = lights[0].L - hit.origin;
                vec3 shadow_dir = normalize(shadow_line);
                ray_t shadow_trace;
                shadow_trace.origin = hit.origin + shadow_dir * BIAS;
                shadow_trace.direction = shadow_dir;
                hit_t shadow_hit = raytrace_iteration(shadow_trace, mat_debug);
                if (shadow_hit.t < length(shadow_line)) color *= 0.1f;
            }
            material_t mat = get_material(hit.material_id);
            if (mat.reflectivity > 0.f) {
                accum *= f;
                vec3 reflect_dir = normalize(reflect(hit.normal, ray.direction)); /* sic: swapped args, :127 */
                ray.origin = hit.origin + reflect_dir * BIAS;
                ray.direction = reflect_dir;
            } else {
                break;
            }
        }
        return color;
    }
};

/* =================================================================================== */
/* APP_ATMOSPHERE — src/app_atmosphere.h (FROM_SPACE defined, :162)                     */
/* =================================================================================== */
struct AppAtmosphere {
    uniforms_t U;
    static constexpr float hg_g = .76f;                        /* app_atmosphere.h:12 */
    vec3 betaR = vec3(5.5e-6f, 13.0e-6f, 22.4e-6f);            /* :29 */
    vec3 betaM = vec3(21e-6f, 21e-6f, 21e-6f);                 /* :30 */
    static constexpr float hR = 7994.0f, hM = 1200.0f;         /* :34-35 */
    static constexpr float earth_radius = 6360e3f;             /* :37 */
    static constexpr float atmosphere_radius = 6420e3f;        /* :38 */
    vec3 sun_dir = vec3(0, 1, 0);                              /* :40  (_mutable: fresh per pixel) */
    static constexpr float sun_power = 20.0f;                  /* :41 */
    static constexpr int num_samples = 16, num_samples_light = 8; /* :47-48 */

    float fov() const { return 1.f; }                          /* :230 */
    sphere_t atmosphere() const { sphere_t s; s.origin = vec3(0, 0, 0); s.radius = atmosphere_radius; s.material = 0; return s; }

    /* app_atmosphere.h:15-26 */
    static bool isect_sphere(const ray_t& ray, const sphere_t& sphere, float& t0, float& t1) {
        vec3 rc = sphere.origin - ray.origin;
        float radius2 = sphere.radius * sphere.radius;
        float tca = dot(rc, ray.direction);
        float d2 = dot(rc, rc) - tca * tca;
        float thc = m_sqrt(radius2 - d2);
        t0 = tca - thc;
        t1 = tca + thc;
        return d2 < radius2;
    }
    /* app_atmosphere.h:50-76 */
    bool get_sun_light(const ray_t& ray, float& optical_depthR, float& optical_depthM) const {
        float t0 = 0, t1 = 0;
        isect_sphere(ray, atmosphere(), t0, t1);
        float march_pos = 0.f;
        float march_step = t1 / (float)num_samples_light;
        for (int i = 0; i < num_samples_light; i++) {
            vec3 sample = ray.origin + ray.direction * (march_pos + 0.5f * march_step);
            float height = length(sample) - earth_radius;
            if (height < 0.f) return false;
            optical_depthR += m_exp(-height / hR) * march_step;
            optical_depthM += m_exp(-height / hM) * march_step;
            march_pos += march_step;
        }
        return true;
    }
    /* app_atmosphere.h:78-160 */
    vec3 get_incident_light(const ray_t& ray) const {
        float t0 = 0, t1 = 0;
        if (!isect_sphere(ray, atmosphere(), t0, t1)) return vec3(0.f, 0.f, 0.f);
        float march_step = t1 / (float)num_samples;
        float mu = dot(ray.direction, sun_dir);
        float phaseR = rayleigh_phase_func(mu);
        float phaseM = henyey_greenstein_phase_func(mu, hg_g);
        float optical_depthR = 0.f, optical_depthM = 0.f;
        vec3 sumR = vec3(0, 0, 0), sumM = vec3(0, 0, 0);
        float march_pos = 0.f;
        for (int i = 0; i < num_samples; i++) {
            vec3 sample = ray.origin + ray.direction * (march_pos + 0.5f * march_step);
            float height = length(sample) - earth_radius;
            float hr = m_exp(-height / hR) * march_step;
            float hm = m_exp(-height / hM) * march_step;
            optical_depthR += hr;
            optical_depthM += hm;
            ray_t light_ray; light_ray.origin = sample; light_ray.direction = sun_dir;
            float optical_depth_lightR = 0.f, optical_depth_lightM = 0.f;
            bool overground = get_sun_light(light_ray, optical_depth_lightR, optical_depth_lightM);
            if (overground) {
                vec3 tau = betaR * (optical_depthR + optical_depth_lightR) +
                           betaM * 1.1f * (optical_depthM + optical_depth_lightM);
                vec3 attenuation = vexp(-tau);
                sumR += hr * attenuation;
                sumM += hm * attenuation;
            }
            march_pos += march_step;
        }
        return sun_power * (sumR * phaseR * betaR + sumM * phaseM * betaM);
    }
    void setup_camera(vec3& eye, vec3& look_at) const {         /* :164-175 (FROM_SPACE) */
        eye = vec3(0, 0, 0);
        look_at = vec3(0, 1, 0);
    }
    void setup_scene() {                                        /* :177-181 */
        mat3 rot = rotate_around_x(-m_abs(m_sin(U.u_time / 2.f)) * 90.f);
        sun_dir = mul(sun_dir, rot);
    }
    /* app_atmosphere.h:183-228 (FROM_SPACE branch :190-209) */
    vec3 render(const ray_t& /*eye*/, vec3 point_cam) const {
        vec3 p = point_cam;
        float z2 = p.x * p.x + p.y * p.y;
        float phi = m_atan2(p.y, p.x);
        float theta = m_acos(1.0f - z2);
        vec3 dir = vec3(m_sin(theta) * m_cos(phi), m_cos(theta), m_sin(theta) * m_sin(phi));
        ray_t ray; ray.origin = vec3(0, earth_radius + 1.f, 0); ray.direction = dir;
        return get_incident_light(ray);
    }
};

/* =================================================================================== */
/* APP_SDF_AO — src/app_sdf_ao.h                                                        */
/* =================================================================================== */
struct AppSdfAo {
    uniforms_t U;
    sdf_ao_aux_t A;
    enum { mat_debug = 0, mat_ground, mat_pipe, mat_bottom, mat_deck, mat_coping, mat_count }; /* :14-20 */
    vec3 materials[mat_count];                                  /* :21 (zero-initialised) */
    vec3 size = vec3(1.3f, 1.f, 1.25f);                         /* :52 */

    float fov() const { return 1.f; }                           /* :312 */
    vec3 background(const ray_t&) const { return vec3(.1f, .1f, .7f); } /* :9-12 */
    vec3 get_material(int index) const {                        /* :23-33 */
        vec3 mat;
        for (int i = 0; i < mat_count; ++i) {
            if (i == index) { mat = materials[i]; break; }
        }
        return mat;
    }
    void setup_scene() {                                        /* :35-43 */
        materials[mat_debug] = vec3(1, 1, 1);
        materials[mat_ground] = vec3(0, .2f, 0);
        materials[mat_pipe] = vec3(.1f, .1f, .1f);
        materials[mat_bottom] = materials[mat_pipe];
        materials[mat_deck] = materials[mat_pipe];
        materials[mat_coping] = vec3(.4f, .4f, .4f);
    }
    void setup_camera(vec3& eye, vec3& look_at) const {          /* :45-50 */
        mat3 rot = rotate_around_y(U.u_time * 50.f);
        eye = mul(rot, vec3(0, 3, 5));
        look_at = vec3(0, 0, 0);
    }
    /* app_sdf_ao.h:54-113 */
    vec2 sdf_pipe(vec3 pos) const {
        vec3 p = pos - vec3(0, size.y, 0);
        float b = sd_box(p, size);
        p -= vec3(.7f, .5f, 0);
        p = mul(p, rotate_around_x(-90.f));
        float c = sd_y_cylinder(p, size.y + .55f, 2.f * size.z + .1f);
        vec2 pipe = vec2(op_sub(b, c), (float)mat_pipe);

        p = pos - vec3(0, size.y, 0);
        p -= vec3(-size.x + .525f, size.y, 0);
        p = mul(p, rotate_around_x(-90.f));
        vec2 coping = vec2(sd_y_cylinder(p, .025f, 2.f * size.z), (float)mat_coping);

        p = pos - vec3(0, size.y * 2.f, 0);
        float rail = sd_box(p + vec3(size.x, -.25f, 0), vec3(.025f, .05f, size.z));
        const vec3 B = vec3(.025f, .125f, .025f);
        const float H = -.125f;
        float bar_1 = sd_box(p + vec3(size.x, H, 0), B);
        float bar_2 = sd_box(p + vec3(size.x, H, size.z / 2.f), B);
        float bar_3 = sd_box(p + vec3(size.x, H, size.z), B);
        float bar_4 = sd_box(p + vec3(size.x, H, -size.z / 2.f), B);
        float bar_5 = sd_box(p + vec3(size.x, H, -size.z), B);
        float b_a = op_add(bar_1, bar_2);
        float b_b = op_add(b_a, bar_3);
        float b_c = op_add(bar_4, bar_5);
        float b_d = op_add(b_b, b_c);
        float bars = b_d;
        vec2 railing = vec2(op_add(rail, bars), (float)mat_deck);
        vec2 deck = op_add(railing, coping);
        return op_add(pipe, deck);
    }
    /* app_sdf_ao.h:115-150 */
    vec2 sdf(vec3 pos) const {
        const float B = .15f;
        vec3 p = pos - vec3(0, B, 0);
        vec2 bottom = vec2(sd_box(p, vec3(2.25f * size.x, B, size.z)), (float)mat_bottom);
        vec2 pipe1 = sdf_pipe(p + vec3(1.25f * size.x, 0, 0));
        p -= vec3(1.25f * size.x, 0, 0);
        p = mul(p, rotate_around_y(180.f));
        vec2 pipe2 = sdf_pipe(p);
        vec2 pipe = op_add(pipe1, pipe2);
        vec2 ref = vec2(sd_box(pos, vec3(.025f, 15, .025f)), (float)mat_debug);
        vec2 ground = vec2(sd_plane(pos, vec3(0, 1, 0), 0.f), (float)mat_ground);
        vec2 g = op_add(ground, ref);
        vec2 b = op_add(pipe, bottom);
        return op_add(b, g);
    }
    /* app_sdf_ao.h:152-163 */
    vec3 sdf_normal(vec3 p) const {
        float dt = 0.001f;
        vec3 x = vec3(dt, 0, 0), y = vec3(0, dt, 0), z = vec3(0, 0, dt);
        return normalize(vec3(sdf(p + x).x - sdf(p - x).x,
                              sdf(p + y).x - sdf(p - y).x,
                              sdf(p + z).x - sdf(p - z).x));
    }
    /* app_sdf_ao.h:165-181 */
    vec3 sdf_ao(const hit_t& hit) const {
        const float dt = .5f;
        const int steps = 5;
        float d = 0.f;
        float occlusion = 0.f;
        for (float i = 1.f; i <= (float)steps; i += 1.f) {
            vec3 p = hit.origin + dt * i * hit.normal;
            d = sdf(p).x;
            occlusion += 1.f / m_pow(2.f, i) * (dt * i - d);
        }
        float c = 1.f - m_clamp(occlusion, 0.f, 1.f);
        return vec3(c, c, c);
    }
    /* app_sdf_ao.h:209-243 */
    vec3 illuminate(vec3 eye, const hit_t& hit, float ao, float sh) const {
        const vec3 sun_dir = normalize(vec3(1, 2, 1));          /* :209 */
        vec3 V = normalize(eye - hit.origin);
        (void)V;
        vec3 accum = vec3(0, 0, 0);
        float sun_ray = m_max(0.f, dot(sun_dir, hit.normal));
        accum += sh * sun_ray * vec3(1.2f, 1.3f, 1.f);
        float h = hit.normal.y;
        accum += ao * h * vec3(.15f, .15f, .4f);
        float ind = m_max(0.f, dot(sun_dir * vec3(-1, 0, -1), hit.normal));
        accum += ao * ind * vec3(.4f, .28f, .2f);
        vec3 mat_c = get_material(hit.material_id);
        if (hit.material_id == mat_ground) {
            float cb = checkboard_pattern(hit.origin.xz(), .5f);
            mat_c = vmix(mat_c - .15f * mat_c, mat_c + .15f * mat_c, cb);
        }
        return accum * mat_c;
    }
    /* app_sdf_ao.h:245-285 */
    vec4 render_impl(const ray_t& ray, vec3 /*point_cam*/) const {
        const int steps = 70;
        const float end = 20.f;
        float t = 0.f;
        for (int i = 0; i < steps; i++) {
            vec3 p = ray.origin + ray.direction * t;
            vec2 d = sdf(p);
            if (t > end) break;
            if (d.x < .005f) {
                hit_t h;
                h.t = t; h.material_id = (int)d.y; h.normal = sdf_normal(p); h.origin = p;
                float ao = sdf_ao(h).x;
                float sh = 1.f;
                return vec4(illuminate(ray.origin, h, ao, sh), t);
            }
            t += d.x;
        }
        return vec4(background(ray), t);
    }
    /* app_sdf_ao.h:287-311 */
    vec3 render(const ray_t& ray, vec3 point_cam) const {
        vec4 orig = render_impl(ray, point_cam);
        const float t = orig.w;
        const vec3 fog_color = vec3(1, 1, 1);
        const float density = A.fog_density;
        const float falloff = A.fog_falloff;
        float fog_factor = density * m_exp(-ray.origin.y * falloff)
                         * (1.f - m_exp(-t * ray.direction.y * falloff))
                         / (ray.direction.y * falloff);
        return vabs(vmix(orig.rgb(), fog_color, fog_factor));
    }
};

/* =================================================================================== */
/* APP_PLANET — src/app_planet.h (CLOUDS and LIGHT defined, :63,249)                    */
/* =================================================================================== */
struct AppPlanet {
    uniforms_t U;
    static constexpr float hg_g = .76f;                        /* :5 (unused by the path) */
    static constexpr float max_height = .4f;                   /* :20 */
    static constexpr float max_ray_dist = max_height * 4.f;    /* :21 */
    static constexpr float vol_coeff_absorb = 30.034f;         /* :68 */
    static constexpr float TERR_EPS = .005f;                   /* :166 */
    enum { TERR_STEPS = 120 };                                 /* :165 */
    volume_sampler_t cloud;                                    /* :69 (_mutable, assigned before use) */
    /* APP_PLANET_ATMOSPHERE (BASELINE config 5 read literally; SURVEY.md 8a note: "planet background() replaced by
       get_incident_light"; NOT in the reference, parity unpinned): the sky of APP_ATMOSPHERE stands in for background() */
    bool atm_sky = false;
    AppAtmosphere sky;

    float fov() const { return m_tan(m_radians(30.f)); }       /* :368 */
    static sphere_t planet() { sphere_t s; s.origin = vec3(0, 0, 0); s.radius = 1.f; s.material = 0; return s; } /* :15-17 */

    /* app_planet.h:23-41 */
    static vec3 background(const ray_t& eye) {
        const vec3 sun_color = vec3(1.f, .9f, .55f);
        float sun_amount = m_clamp(dot(eye.direction, vec3(0, 0, 1)), 0.f, 1.f);
        vec3 sky = vmix(vec3(.0f, .05f, .2f), vec3(.15f, .3f, .4f), 1.0f - eye.direction.y);
        sky += sun_color * m_clamp(m_pow(sun_amount, 30.0f) * 5.0f, 0.f, 1.f);
        sky += sun_color * m_clamp(m_pow(sun_amount, 10.0f) * .6f, 0.f, 1.f);
        return vabs(sky);
    }
    void setup_scene() {
        if (atm_sky) { sky.U = U; sky.sun_dir = vec3(0, 1, 0); sky.setup_scene(); }   /* app_atmosphere.h:177-181, fresh per pixel */
    }
    /* background() of the path: app_planet.h:23-41, or — composite — get_incident_light (app_atmosphere.h:78-160) for the ray
       from 1 m above the ground (:204-207) along the view direction */
    vec3 bg(const ray_t& eye) const {
        if (!atm_sky) return background(eye);
        ray_t r; r.origin = vec3(0, AppAtmosphere::earth_radius + 1.f, 0); r.direction = eye.direction;
        return sky.get_incident_light(r);
    }
    void setup_camera(vec3& eye, vec3& look_at) const {         /* :47-58 */
        eye = vec3(0, 0, -2.5f);
        look_at = vec3(0, 0, 2);
    }
    /* noise bases: :9, :65, :167 */
    static float noise(vec3 p) { return noise_iq(p); }
    static float anoise(vec3 p) { return m_abs(noise(p) * 2.f - 1.f); }
    static float rnoise(vec3 p) { return 1.f - m_abs(noise(p) * 2.f - 1.f); }
    static float fbm_clouds(vec3 pos, float l, float ig, float g) { return fbm_generic<4>(pos, l, ig, g, anoise); }         /* :66 */
    static float fbm_terr(vec3 pos, float l, float ig, float g) { return fbm_generic<3>(pos, l, ig, g, noise); }            /* :169 */
    static float fbm_terr_r(vec3 pos, float l, float ig, float g) { return fbm_generic<3>(pos, l, ig, g, rnoise); }         /* :170 */
    static float fbm_terr_normals(vec3 pos, float l, float ig, float g) { return fbm_generic<7>(pos, l, ig, g, noise); }    /* :172 */
    static float fbm_terr_r_normals(vec3 pos, float l, float ig, float g) { return fbm_generic<7>(pos, l, ig, g, rnoise); } /* :173 */

    /* app_planet.h:71-77 */
    static float illuminate_volume(const volume_sampler_t& c) { return m_exp(c.height) / .055f; }
    /* app_planet.h:79-100 */
    static void integrate_volume(volume_sampler_t& vol, float density, float dt) {
        float T_i = m_exp(-vol_coeff_absorb * density * dt);
        vol.transmittance *= T_i;
        vol.radiance += density * illuminate_volume(vol) * vol.transmittance * dt;
        vol.alpha += (1.f - T_i) * (1.f - vol.alpha);
    }
    /* app_planet.h:102-119 */
    static void clouds_map(volume_sampler_t& c, float t_step) {
        float dens = fbm_clouds(c.pos * 3.2343f + vec3(.35f, 13.35f, 2.67f), 2.0276f, .5f, .5f);
        const float cld_coverage = .29475675f, cld_fuzzy = .0335f;
        dens *= m_smoothstep(cld_coverage, cld_coverage + cld_fuzzy, dens);
        dens *= band(.2f, .35f, .65f, c.height);
        integrate_volume(c, dens, t_step);
    }
    /* app_planet.h:121-141 */
    static void clouds_march(const ray_t& eye, volume_sampler_t& c, float max_travel, const mat3& rot) {
        const int steps = 75;
        const float t_step = max_ray_dist / (float)steps;
        float t = 0.f;
        for (int i = 0; i < steps; i++) {
            if (t > max_travel || c.alpha >= 1.f) return;
            vec3 o = c.origin + t * eye.direction;
            c.pos = mul(rot, o - planet().origin);
            c.height = (length(c.pos) - planet().radius) / max_height;
            t += t_step;
            clouds_map(c, t_step);
        }
    }
    /* app_planet.h:143-160 */
    static void clouds_shadow_march(vec3 dir, volume_sampler_t& c, const mat3& rot) {
        const int steps = 5;
        const float t_step = max_height / (float)steps;
        float t = 0.f;
        for (int i = 0; i < steps; i++) {
            vec3 o = c.origin + t * dir;
            c.pos = mul(rot, o - planet().origin);
            c.height = (length(c.pos) - planet().radius) / max_height;
            t += t_step;
            clouds_map(c, t_step);
        }
    }
    /* app_planet.h:175-186 */
    static vec2 sdf_terrain_map(vec3 pos) {
        float h0 = fbm_terr(pos * 2.0987f, 2.0244f, .454f, .454f);
        float n0 = m_smoothstep(.35f, 1.f, h0);
        float h1 = fbm_terr_r(pos * 1.50987f + vec3(1.9489f, 2.435f, .5483f), 2.0244f, .454f, .454f);
        float n1 = m_smoothstep(.6f, 1.f, h1);
        float n = n0 + n1;
        return vec2(length(pos) - planet().radius - n * max_height, n / max_height);
    }
    /* app_planet.h:188-199 */
    static vec2 sdf_terrain_map_detail(vec3 pos) {
        float h0 = fbm_terr_normals(pos * 2.0987f, 2.0244f, .454f, .454f);
        float n0 = m_smoothstep(.35f, 1.f, h0);
        float h1 = fbm_terr_r_normals(pos * 1.50987f + vec3(1.9489f, 2.435f, .5483f), 2.0244f, .454f, .454f);
        float n1 = m_smoothstep(.6f, 1.f, h1);
        float n = n0 + n1;
        return vec2(length(pos) - planet().radius - n * max_height, n / max_height);
    }
    /* app_planet.h:201-212 */
    static vec3 sdf_terrain_normal(vec3 p) {
        const float e = 0.001f;
        vec3 dx = vec3(e, 0, 0), dy = vec3(0, e, 0), dz = vec3(0, 0, e);
        return normalize(vec3(
            sdf_terrain_map_detail(p + dx).x - sdf_terrain_map_detail(p - dx).x,
            sdf_terrain_map_detail(p + dy).x - sdf_terrain_map_detail(p - dy).x,
            sdf_terrain_map_detail(p + dz).x - sdf_terrain_map_detail(p - dz).x));
    }
    /* app_planet.h:217-236 */
    static vec3 setup_lights(vec3 L, vec3 normal) {
        vec3 diffuse = vec3(0, 0, 0);
        vec3 c_L = vec3(7, 5, 3);
        diffuse += m_max(0.f, dot(L, normal)) * c_L;
        float hemi = m_clamp(.25f + .5f * normal.y, .0f, 1.f);
        diffuse += hemi * vec3(.4f, .6f, .8f) * .2f;
        float amb = m_clamp(.12f + .8f * m_max(0.f, dot(-L, normal)), 0.f, 1.f);
        diffuse += amb * vec3(.4f, .5f, .6f);
        return diffuse;
    }
    /* app_planet.h:238-298 (LIGHT defined) */
    static vec3 illuminate(vec3 pos, vec3 /*eye*/, const mat3& local_xform, vec2 df) {
        float h = df.y;
        vec3 w_normal = normalize(pos);
        vec3 normal = sdf_terrain_normal(pos);
        float N = dot(normal, w_normal);
        const vec3 c_water = vec3(.015f, .110f, .455f), c_grass = vec3(.086f, .132f, .018f),
                   c_beach = vec3(.153f, .172f, .121f), c_rock = vec3(.080f, .050f, .030f),
                   c_snow = vec3(.600f, .600f, .600f);
        const float l_water = .05f, l_shore = .17f, l_grass = .211f, l_rock = .351f;
        float s = m_smoothstep(.4f, 1.f, h);
        vec3 rock = vmix(c_rock, c_snow, m_smoothstep(1.f - .3f * s, 1.f - .2f * s, N));
        vec3 grass = vmix(c_grass, rock, m_smoothstep(l_grass, l_rock, h));
        vec3 shoreline = vmix(c_beach, grass, m_smoothstep(l_shore, l_grass, h));
        vec3 water = vmix(c_water / 2.f, c_water, m_smoothstep(0.f, l_water, h));
        vec3 L = mul(local_xform, normalize(vec3(1, 1, 0)));
        shoreline *= setup_lights(L, normal);
        vec3 ocean = setup_lights(L, w_normal) * water;
        return vmix(ocean, shoreline, m_smoothstep(l_water, l_shore, h));
    }
    /* app_planet.h:303-367 */
    vec3 render(const ray_t& eye, vec3 /*point_cam*/) {
        mat3 rot_y = rotate_around_y(27.f);
        mat3 rot = mul(rotate_around_x(U.u_time * -12.f), rot_y);
        mat3 rot_cloud = mul(rotate_around_x(U.u_time * 8.f), rot_y);
        sphere_t atmosphere = planet();
        atmosphere.radius += max_height;
        hit_t hit = no_hit();
        intersect_sphere(eye, atmosphere, hit);
        if (hit.material_id < 0) return bg(eye);

        float t = 0.f;
        vec2 df = vec2(1, max_height);
        vec3 pos;
        float max_cld_ray_dist = max_ray_dist;
        for (int i = 0; i < TERR_STEPS; i++) {
            if (t > max_ray_dist) break;
            vec3 o = hit.origin + t * eye.direction;
            pos = mul(rot, o - planet().origin);
            df = sdf_terrain_map(pos);
            if (df.x < TERR_EPS) { max_cld_ray_dist = t; break; }
            t += df.x * .4567f;
        }
        cloud = construct_volume(hit.origin);
        clouds_march(eye, cloud, max_cld_ray_dist, rot_cloud);

        if (df.x < TERR_EPS) {
            vec3 c_terr = illuminate(pos, eye.direction, rot, df);
            vec3 c_cld = cloud.radiance;
            float alpha = cloud.alpha;
            float shadow = 1.f;
            pos = mul(transpose(rot), pos);
            cloud = construct_volume(pos);
            vec3 local_up = normalize(pos);
            clouds_shadow_march(local_up, cloud, rot_cloud);
            shadow = m_mix(.7f, 1.f, m_step(cloud.alpha, 0.33f));
            return vabs(vmix(c_terr * shadow, c_cld, alpha));
        } else {
            return vabs(vmix(bg(eye), cloud.radiance, cloud.alpha));
        }
    }
};

/* =================================================================================== */
/* APP_VINYL — src/app_vinyl.h (C++ build: 60 march steps, :411-416; SHADERTOY undefined)  */
/* SURVEY.md §8f row 4 ("remaining apps").  The reference gives no known answers for it.   */
/* =================================================================================== */
struct AppVinyl {
    uniforms_t U;
    int march_steps = 60;                                    /* :411-416: the __cplusplus value; 180 = the GLSL / HLSL builds */
    enum { num_materials = 8, mat_debug = 0, mat_groove = 1, mat_dead_wax, mat_label, mat_logo, mat_shiny }; /* :20-24 */
    material_t materials[num_materials];                     /* material.h:17 */
    mat3 platter_rot;                                        /* :66 (_mutable, assigned in render before use) */
    vec3 sun_dir = normalize(vec3(-1, 4, -3));               /* :285-286 */

    float fov() const { return 1.f; }                        /* :459 */
    vec3 background(const ray_t&) const { return vec3(1, 1, 1); }   /* :15-18 */
    material_t get_material(int index) const {               /* material.h:19-36 */
        material_t mat;
        for (int i = 0; i < num_materials; ++i) if (i == index) { mat = materials[i]; break; }
        return mat;
    }
    static void setup_mat(material_t& mat, vec3 diffuse, float metallic, float roughness) {   /* :26-38 */
        mat.base_color = diffuse; mat.metallic = metallic; mat.roughness = roughness;
        mat.ior = 1.f; mat.reflectivity = 0.f; mat.translucency = 0.f;
    }
    void setup_scene() {                                     /* :40-54 */
        setup_mat(materials[mat_debug], vec3(1, 1, 1), .0f, .0f);
        setup_mat(materials[mat_groove], vec3(.01f, .01f, .01f), .0f, .013f);
        setup_mat(materials[mat_dead_wax], vec3(.05f, .05f, .05f), .0f, .005f);
        setup_mat(materials[mat_label], vec3(.5f, .5f, .0f), .0f, .5f);
        setup_mat(materials[mat_logo], vec3(0, 0, .7f), .0f, .5f);
        setup_mat(materials[mat_shiny], vec3(.7f, .7f, .7f), 1.f, .01f);
    }
    void setup_camera(vec3& eye, vec3& look_at) const {       /* :56-64 */
        eye = vec3(0, 5.75f, 6.75f);
        look_at = vec3(0, -2.5f, 0);
    }
    /* :68-85 */
    static float sdf_logo(vec3 pos, float thick) {
        vec3 b = vec3(.25f, thick, 1.2f);
        vec3 d = vec3(.7f, 0, 0);
        vec3 p = mul(pos, rotate_around_y(30.f));
        float v1 = sd_box(p - d, b);
        p = mul(pos, rotate_around_y(-30.f));
        float v2 = sd_box(p + d, b);
        float x = sd_box(pos, vec3(1.5f, thick, 1.35f));
        float v = op_add(v1, v2);
        return op_intersect(v, x);
    }
    /* :87-125 */
    static vec2 sdf_platter(vec3 p) {
        const float thick = .1f;
        vec2 lead_in = vec2(sd_y_cylinder(p, 6.f, thick - .05f), (float)mat_dead_wax);
        vec2 groove = vec2(sd_y_cylinder(p, 5.9f, thick), (float)mat_groove);
        vec2 dead_wax = vec2(sd_y_cylinder(p, 3.f, thick), (float)mat_dead_wax);
        vec2 label = vec2(sd_y_cylinder(p, 2.f, thick), (float)mat_label);
        vec2 logo = vec2(sdf_logo(p, thick - .0175f), (float)mat_logo);
        float spc = sd_y_cylinder(p, .10f, .6f);
        float sps = sd_sphere(p - vec3(0, .3f, 0), .10f);
        vec2 spindle = vec2(op_add(spc, sps), (float)mat_shiny);
        vec2 d0 = op_add(groove, lead_in);
        vec2 d1 = op_add(d0, dead_wax);
        vec2 d2 = op_add(label, logo);
        vec2 d3 = op_add(d1, d2);
        vec2 d4 = op_add(d3, spindle);
        float defect1 = sd_sphere(p + vec3(6.05f, 0, 0), .1f);
        float defect2 = sd_sphere(p + vec3(-6.05f, 0, 0), .1f);
        float defect = op_add(defect1, defect2);
        return vec2(op_sub(d4.x, defect), d4.y);
    }
    /* :127-255 */
    vec2 sdf_tonearm(vec3 pos) const {
        vec3 base_p = vec3(-7, 0, -5);
        float platter = sd_y_cylinder(pos, 6.25f, 1.f);
        float base_0 = sd_y_cylinder(pos - base_p, 3.f, .25f);
        float base_1 = op_sub(base_0, platter);
        float base_2 = sd_y_cylinder(pos - base_p, 1.25f, 1.f);
        float base_12 = op_add(base_1, base_2);
        vec2 base_a = vec2(base_12, (float)mat_shiny);
        vec2 base_b = vec2(sd_y_cylinder(pos - base_p, 0.5f, 2.5f), (float)mat_shiny);
        vec2 base = op_add(base_a, base_b);

        vec3 p = mul(pos, rotate_around_x(m_sin(U.u_time * 3.6758f) * .1f));

        const float R = .1f;
        const float H = .8f;
        vec3 a1 = vec3(-6, H, -3);
        vec3 a11 = vec3(-4.25f, H, 2);
        vec3 a2 = vec3(-4.1f, H, 2.45f);
        vec3 a33 = vec3(-3.5f, H, 3);
        vec3 a3 = vec3(-2, H, 4);
        float arm1 = sd_capsule(p, base_p + vec3(-1, H, -2), a1, R);
        float arm2 = sd_capsule(p, a1, a11, R);
        float arm3 = sd_capsule(p, a33, a3, R);
        vec2 armb = sd_bezier(a11, a2, a33, p, R);
        float arm_link1 = op_add(arm1, arm2);
        float arm_link2 = op_add(arm_link1, arm3);
        vec2 arm = vec2(op_add(arm_link2, armb.x), (float)mat_shiny);

        vec3 arm_fwd = normalize(a3 - a33);
        vec3 arm_up = vec3(0, 1, 0);
        vec3 arm_right = cross(arm_fwd, arm_up);
        mat3 arm_xform;
        arm_xform.c[0] = arm_fwd; arm_xform.c[1] = arm_up; arm_xform.c[2] = arm_right;   /* mat3(col, col, col) */

        vec3 clr_p = p - a3;
        float clr_r = R * 1.5f;
        float collar = sd_cylinder(clr_p, vec3(0, 0, 0), vec3(0, 0, 0) + arm_fwd * .05f, clr_r);

        const float fl_w = .045f;
        const float fl_h = .020f;
        float fl_len1 = clr_r * 1.f;
        float fl_len2 = fl_len1 * 1.2f;
        mat3 fl_rot = mul(arm_xform, rotate_around_x(45.f));
        vec3 fl_p = mul(clr_p - arm_right * clr_r - arm_up * clr_r, fl_rot);
        float fl1 = sd_box(fl_p, vec3(fl_w, fl_h, fl_len1));
        mat3 fl_rot2 = rotate_around_x(-45.f);
        float fl2 = sd_box(mul(fl_p - vec3(0, 0, fl_len1), fl_rot2) - vec3(0, 0, fl_len2), vec3(fl_w, fl_h, fl_len2));
        float finger_lift = op_add(fl1, fl2);
        vec2 headshell = vec2(op_add(collar, finger_lift), (float)mat_shiny);

        const float ctg_w = .05f;
        const float ctg_h = .05f;
        float ctg_len1 = .3f;
        float ctg_len2 = .5f;
        vec3 ctg_p = mul(clr_p, arm_xform);
        float ctg1 = sd_box(ctg_p, vec3(ctg_len1, ctg_h, ctg_w));
        mat3 ctg_rot = rotate_around_z(44.f);
        vec3 ctg2_p = mul(ctg_p - vec3(ctg_len1, 0, 0), ctg_rot) - vec3(ctg_len2 - 0.03f, -.01f, 0);
        float ctg2 = sd_box(ctg2_p, vec3(ctg_len2, ctg_h, ctg_w));
        float cut = sd_box(mul(mul(ctg2_p, rotate_around_x(10.f)) - vec3(0, .05f, .175f), rotate_around_y(-5.f)),
                           vec3(ctg_len2 * 2.f, ctg_h * 3.f, ctg_w * 3.2f));
        float cut2 = sd_box(mul(ctg2_p - vec3(.3f, .2f, 0), rotate_around_z(10.f)), vec3(.4f, .2f, .3f));
        float ctg12 = op_add(ctg1, ctg2);
        float ctg12c = op_sub(ctg12, cut);
        vec2 cartridge = vec2(op_sub(ctg12c, cut2), (float)mat_shiny);

        vec2 tone1 = op_add(base, arm);
        vec2 tone2 = op_add(headshell, cartridge);
        return op_add(tone1, tone2);
    }
    /* :257-265 */
    vec2 sdf(vec3 pos) const {
        vec3 p = mul(pos, platter_rot);
        vec2 plat = sdf_platter(p);
        vec2 arm = sdf_tonearm(pos);
        return op_add(plat, arm);
    }
    /* :267-278 */
    vec3 sdf_normal(vec3 p) const {
        float dt = 0.001f;
        vec3 x = vec3(dt, 0, 0), y = vec3(0, dt, 0), z = vec3(0, 0, dt);
        return normalize(vec3(sdf(p + x).x - sdf(p - x).x, sdf(p + y).x - sdf(p - y).x, sdf(p + z).x - sdf(p - z).x));
    }
    static float saw(float x) { return x - m_floor(x); }           /* :280-283 */
    static float pulse(float x) { return saw(x + .5f) - saw(x); }  /* :285-288 */
    /* :293-377 */
    vec3 illuminate(vec3 eye, hit_t& hit) const {
        vec3 L = sun_dir;
        vec3 V = normalize(eye - hit.origin);
        material_t mat = get_material(hit.material_id);
        if (hit.material_id == mat_groove || hit.material_id == mat_dead_wax) {
            hit.origin = mul(hit.origin, platter_rot);
            L = mul(L, platter_rot);
            V = mul(V, platter_rot);
            float r = length(hit.origin);
            vec3 B = hit.origin / r;
            vec3 N = vec3(0, 1, 0);
            if (hit.material_id == mat_groove) {
                float rr = r + .07575f * noise_iq(hit.origin * 2.456f);
                float s = pulse(rr * 24.f);
                if (s > 0.f) {
                    N = normalize(N + B);
                    N = reflect(N, vec3(0, 1, 0));
                }
            }
            if (hit.material_id == mat_dead_wax) {
                float s = saw(r * 4.f);
                N = normalize(N + B * (float)(s > .9f));
            }
            vec3 T = cross(B, N);
            const float ro_diff = 1.f;
            const float ro_spec = .0725f;
            const float a_x = .025f;
            const float a_y = .5f;
            vec3 H = normalize(V + L);
            float dotLN = dot(L, N);
            vec3 diffuse = mat.base_color * (ro_diff / PI) * m_max(0.f, dotLN);
            float spec_a = ro_spec / m_sqrt(dotLN * dot(V, N));
            float spec_b = 1.f / (4.f * PI * a_x * a_y);
            float ht = dot(H, T) / a_x;
            float hb = dot(H, B) / a_y;
            float spec_c = -2.f * (ht * ht + hb * hb) / (1.f + dot(H, N));
            vec3 specular = vec3(1, 1, 1) * spec_a * spec_b * m_exp(spec_c);
            return diffuse + specular;
        } else {
            hit.normal = sdf_normal(hit.origin);
            vec3 diffuse = mat.base_color * m_max(0.f, dot(L, hit.normal));
            vec3 H = normalize(V + L);
            vec3 specular = m_pow(m_max(0.f, dot(H, hit.normal)), 50.f) * vec3(1, 1, 1);
            return diffuse + specular;
        }
    }
    /* :379-404 */
    float sdf_shadow(const ray_t& ray) const {
        const int steps = 20;
        const float end = 5.f;
        const float penumbra_factor = 16.f;
        const float darkest = .05f;
        float t = 0.f;
        float umbra = 1.f;
        for (int i = 0; i < steps; i++) {
            vec3 p = ray.origin + ray.direction * t;
            vec2 d = sdf(p);
            if (t > end) break;
            if (d.x < .005f) return darkest;
            t += d.x;
            umbra = m_min(umbra, penumbra_factor * d.x / t);
        }
        return umbra;
    }
    /* :406-457 */
    vec3 render(const ray_t& ray, vec3 /*point_cam*/) {
        const int steps = march_steps;                        /* 60 under __cplusplus, 180 otherwise :411-416 */
        const float end = 40.f;
        float rot = U.u_time * 200.f;
        platter_rot = mul(rotate_around_y(rot), rotate_around_x(m_sin(U.u_time) * .1f));
        float t = 0.f;
        for (int i = 0; i < steps; i++) {
            vec3 p = ray.origin + ray.direction * t;
            vec2 d = sdf(p);
            if (t > end) break;
            if (d.x < .005f) {
                hit_t h;
                h.t = t; h.material_id = (int)d.y; h.normal = vec3(0, 1, 0); h.origin = p;
                float sh = 1.f;
                ray_t sh_ray; sh_ray.origin = p + sun_dir * 0.05f; sh_ray.direction = sun_dir;
                sh = sdf_shadow(sh_ray);
                return illuminate(ray.origin, h) * sh;
            }
            t += d.x;
        }
        return background(ray);
    }
};

/* =================================================================================== */
/* The UE4 cloud variant — ue4/volumetric_clouds/Shaders/app_clouds.usf.             */
/* An HLSL library for an Unreal material: its entry is ue4_render_clouds(cam_dir, time, coverage, ...)     */
/* (:234-265), called per pixel by a material graph that is a binary asset.  THE BUILD'S HOST MAPPING       */
/* (there is nothing in the tree to pin it): cam_dir = the primary-ray direction of APP_CLOUDS' mainImage   */
/* camera (src/app_clouds.h:23-30, FOV 1), time = u_time, the parameters = the TWEAK block's defaults       */
/* (:4-17) unless an aux block overrides them, and the result goes through main.h's sRGB epilogue like      */
/* every other app.  No known answers exist: PARITY UNPINNED (review only).                                 */
/* =================================================================================== */
struct clouds_ue4_aux_t {                                       /* the material's scalar / vector parameters */
    float coverage = .50f, thickness = 15.f, absorbtion = 1.030725f, fuzziness = 0.035f;   /* :4-7 */
    vec3 sun_dir, wind_dir;
    bool has_dirs = false;                                      /* false: SUN_DIR / WIND_DIR macros (:13-14) */
};
struct AppCloudsUe4 {
    uniforms_t U;
    clouds_ue4_aux_t A;
    static constexpr int STEPS = 25;                            /* :16 */
    float fov() const { return 1.f; }
    void setup_scene() {}
    void setup_camera(vec3& eye, vec3& look_at) const {          /* host mapping: APP_CLOUDS' camera */
        eye = vec3(0, -.5f, 0);
        float angle = U.u_mouse.x * .5f;
        look_at = mul(rotate_around_y(angle), vec3(0, 0, -1));
    }
    /* :123-135 — unrolled 4-octave fBm with its own weights; FBM_FREQ 2.76434 (:9) */
    static float fbm(vec3 pos, float lacunarity) {
        vec3 p = pos;
        float t = 0.51749673f * noise_iq(p); p = p * lacunarity;
        t += 0.25584929f * noise_iq(p); p = p * lacunarity;
        t += 0.12527603f * noise_iq(p); p = p * lacunarity;
        t += 0.06255931f * noise_iq(p);
        return t;
    }
    static float noise_func(vec3 x) { return fbm(x, 2.76434f); }  /* :142-149 */
    /* :151-162 (returns sky WITHOUT abs, unlike app_clouds.h) */
    static vec3 render_sky_color(vec3 eye_dir, vec3 sun_dir) {
        const vec3 sun_color = vec3(1.f, .7f, .55f);
        float sun_amount = m_max(dot(eye_dir, sun_dir), 0.f);
        vec3 sky = vmix(vec3(.0f, .1f, .4f), vec3(.3f, .6f, .8f), 1.0f - eye_dir.y);
        sky += sun_color * m_min(m_pow(sun_amount, 1500.0f) * 5.0f, 1.0f);
        sky += sun_color * m_min(m_pow(sun_amount, 10.0f) * .6f, 1.0f);
        return sky;
    }
    /* :164-179 */
    static float density_func(vec3 pos, vec3 offset, float coverage, float fuziness) {
        vec3 p = pos * .0212242f + offset;
        float dens = noise_func(p);
        dens *= m_smoothstep(coverage, coverage + fuziness, dens);
        return m_clamp(dens, 0.f, 1.f);
    }
    /* :181-231, plane projection branch */
    static vec4 render_clouds(const ray_t& eye, vec3 /*sun_dir*/, vec3 wind_dir, float coverage, float thickness,
                              float absorbtion, float fuzziness) {
        const int steps = STEPS;
        float march_step = thickness / float(steps);
        vec3 dir_step = eye.direction / eye.direction.y * march_step;
        vec3 pos = eye.origin + eye.direction * 100.f;
        float T = 1.f;
        vec3 C = vec3(0, 0, 0);
        float alpha = 0.f;
        for (int i = 0; i < steps; i++) {
            float h = float(i) / float(steps);
            float dens = density_func(pos, wind_dir, coverage, fuzziness);
            float T_i = m_exp(-absorbtion * dens * march_step);
            T *= T_i;
            C += T * (m_exp(h) / 1.75f) * dens * march_step;
            alpha += (1.f - T_i) * (1.f - alpha);
            pos += dir_step;
        }
        return vec4(C, alpha);
    }
    /* :234-265 */
    static vec3 ue4_render_clouds(vec3 cam_dir, float /*time*/, float coverage, float thickness, float absorbtion,
                                  float fuzziness, vec3 sun_dir, vec3 wind_dir) {
        ray_t eye_ray;
        eye_ray.origin = vec3(0, 0, 0);
        eye_ray.direction = cam_dir;
        vec3 sky = render_sky_color(eye_ray.direction, sun_dir);
        vec4 cld = render_clouds(eye_ray, sun_dir, wind_dir, 1.f - coverage, thickness, absorbtion, fuzziness);
        return vmix(sky, cld.rgb(), cld.w);
    }
    vec3 render(const ray_t& eye_ray, vec3 /*point_cam*/) const {
        vec3 sun = A.has_dirs ? A.sun_dir : normalize(vec3(0, m_abs(m_sin(U.u_time * .3f)), -1));     /* SUN_DIR :14 */
        vec3 wind = A.has_dirs ? A.wind_dir : vec3(0, 0, -U.u_time * .2f);                           /* WIND_DIR :13 */
        return ue4_render_clouds(eye_ray.direction, U.u_time, A.coverage, A.thickness, A.absorbtion, A.fuzziness, sun, wind);
    }
};

/* =================================================================================== */
/* "clouds_best" — src/app_clouds_best.h, the stand-alone (flattened) cloud shader; SURVEY.md §8f row 4. */
/* Not selected by an APP_* define in the reference: it is a complete shader with its own mainImage      */
/* (:669-696, identical to src/main.h) and FOV 1 (:663).  C++ branch of its macros (:31-40).             */
/* =================================================================================== */
struct AppCloudsBest {
    uniforms_t U;
    static constexpr int cld_march_steps = 50;                 /* :410 */
    static constexpr float cld_coverage = .3125f;              /* :411 */
    static constexpr float cld_thick = 90.f;                   /* :412 */
    static constexpr float cld_absorb_coeff = 1.f;             /* :413 */
    vec3 cld_wind_dir() const { return vec3(0, 0, -U.u_time * .2f); }         /* :414 */
    static vec3 cld_sun_dir() { return normalize(vec3(0, 0, -1)); }           /* :415 */

    struct volume_t {                                          /* :362-372 */
        vec3 origin, pos;
        float height = 0.f, coeff_absorb = 0.f, T = 1.f;
        vec3 C;
        float alpha = 0.f;
    };
    float fov() const { return 1.f; }                          /* :663 */
    void setup_scene() {}                                      /* :643-645 */
    void setup_camera(vec3& eye, vec3& look_at) const {        /* :635-641 */
        eye = vec3(0, 1.f, 0);
        look_at = vec3(0, 1.6f, -1);
    }
    /* :562 DECL_FBM_FUNC(fbm_clouds, 5, abs(noise(p))) with noise = snoise (:552) */
    static float fbm_clouds(vec3 pos, float lacunarity, float init_gain, float gain) {
        return fbm_generic<5>(pos, lacunarity, init_gain, gain, [](vec3 p) { return m_abs(snoise(p)); });
    }
    /* :564-575 */
    vec3 render_sky_color(vec3 eye_dir) const {
        const vec3 sun_color = vec3(1.f, .7f, .55f);
        float sun_amount = m_max(dot(eye_dir, cld_sun_dir()), 0.f);
        vec3 sky = vmix(vec3(.0f, .1f, .4f), vec3(.3f, .6f, .8f), 1.0f - eye_dir.y);
        sky += sun_color * m_min(m_pow(sun_amount, 1500.0f) * 5.0f, 1.0f);
        sky += sun_color * m_min(m_pow(sun_amount, 10.0f) * .6f, 1.0f);
        return sky;
    }
    /* :577-589 */
    float density_func(vec3 pos, float /*h*/) const {
        vec3 p = pos * .001f + cld_wind_dir();
        float dens = fbm_clouds(p * 2.032f, 2.6434f, .5f, .5f);
        dens *= m_smoothstep(cld_coverage, cld_coverage + .035f, dens);
        return dens;
    }
    /* :591-597 */
    static float illuminate_volume(const volume_t& cloud) { return m_exp(cloud.height) / 1.95f; }
    /* :392-407 */
    static void integrate_volume(volume_t& vol, float density, float dt) {
        float T_i = m_exp(-vol.coeff_absorb * density * dt);
        vol.T *= T_i;
        vol.C += vol.T * illuminate_volume(vol) * density * dt;
        vol.alpha += (1.f - T_i) * (1.f - vol.alpha);
    }
    /* :599-633 */
    vec4 render_clouds(const ray_t& eye) const {
        const int steps = cld_march_steps;
        const float march_step = cld_thick / float(steps);
        vec3 projection = eye.direction / eye.direction.y;
        vec3 iter = projection * march_step;
        float cutoff = dot(eye.direction, vec3(0, 1, 0));
        volume_t cloud;                                        /* begin_volume :374-384 */
        cloud.origin = eye.origin + projection * 100.f;
        cloud.pos = cloud.origin;
        cloud.coeff_absorb = cld_absorb_coeff;
        for (int i = 0; i < steps; i++) {
            cloud.height = (cloud.pos.y - cloud.origin.y) / cld_thick;
            float dens = density_func(cloud.pos, cloud.height);
            integrate_volume(cloud, dens, march_step);
            cloud.pos += iter;
            if (cloud.alpha > .999f) break;
        }
        return vec4(cloud.C, cloud.alpha * m_smoothstep(.0f, .2f, cutoff));
    }
    /* :647-661 */
    vec3 render(const ray_t& eye_ray, vec3 /*point_cam*/) const {
        vec3 sky = render_sky_color(eye_ray.direction);
        if (dot(eye_ray.direction, vec3(0, 1, 0)) < 0.05f) return sky;
        vec4 cld = render_clouds(eye_ray);
        vec3 col = vmix(sky, cld.rgb(), cld.w);
        return col;
    }
};

} /* namespace sbxref */
#endif
