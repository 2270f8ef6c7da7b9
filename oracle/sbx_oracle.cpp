/* oracle/sbx_oracle.cpp — C entry points of the CPU oracle (libsbx_oracle.so).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md): loaded by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the product.
 *
 * PINNING: the reference has no tests, golden vectors or fixtures, and its own C++ program cannot be built
 * in this image (VML + SDL harness are external and absent), so this restatement is pinned against the known
 * answers of SURVEY.md Appendix C (tests/test_oracle_kat.py) — values the survey obtained by compiling the
 * reference headers verbatim over glibc.  Against the author's actual binary: PARITY UNPINNED.  APP_VINYL and
 * app_clouds_best.h have no Appendix C values at all (review + libm comparison only).
 *
 * The host loop here plays the role of the reference's external per-pixel harness
 * (vml/test/SDL_app/SDL_app.cpp, named at /root/reference/src/Makefile:21, absent from
 * the tree): for every pixel it constructs a fresh app (GLSL per-invocation semantics)
 * and calls mainImage with fragCoord = (x + .5, y + .5); row 0 is the bottom row
 * (/root/reference/src/main.h:40-43, the y flip is HLSL-only).
 */
#include "ref_apps.h"
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace sbxref;

/* app ids: same order as the README table (/root/reference/README.md:15-22) + SDF_AO */
enum { APP_PLANET = 0, APP_CLOUDS = 1, APP_VINYL = 2, APP_EGG = 3, APP_RAYTRACER = 4, APP_ATMOSPHERE = 5, APP_SDF_AO = 6,
       APP_CLOUDS_BEST = 7 /* src/app_clouds_best.h: not an APP_* define of the reference, numbered after them */,
       APP_CLOUDS_TEX = 8  /* APP_CLOUDS compiled with USE_NOISE_TEX (src/app_clouds.h:9,51-56,69-81) */,
       APP_CLOUDS_UE4 = 9  /* ue4/volumetric_clouds/Shaders/app_clouds.usf under the build's host mapping (ref_apps.h) */,
       APP_CLOUDS_SKY = 10 /* APP_CLOUDS compiled with SKY_SPHERE (src/app_clouds.h:8,14-19,154-162) */,
       APP_VINYL_GPU = 11  /* APP_VINYL with the march length of its non-C++ builds: 180 steps (src/app_vinyl.h:411-416) */,
       APP_PLANET_ATMOSPHERE = 12 /* the config-5 composite: APP_PLANET with background() = APP_ATMOSPHERE's get_incident_light
                                     (ref_apps.h AppPlanet::bg; definition in include/sbx.h).  PARITY UNPINNED. */ };

/* the two bound 3-D textures of the USE_NOISE_TEX build (t1, t2): set by sbxo_set_noise_volumes, owned by the caller */
static noise_tex_t g_tex_noise, g_tex_noise_2;

/* aux blocks arrive as the 16-byte-register images of src/uniform_buffer.h:39-60 */
static clouds_aux_t parse_clouds_aux(const void* aux) {
    clouds_aux_t a;
    if (!aux) return a;
    const float* f = (const float*)aux;
    const int* i = (const int*)aux;
    a.wind_dir = vec3(f[0], f[1], f[2]);        /* c0 */
    a.sun_dir = vec3(f[4], f[5], f[6]);         /* c1 */
    a.sun_color = vec3(f[8], f[9], f[10]);      /* c2 */
    a.sun_power = f[12];                        /* c3.x */
    a.cld_march_steps = i[13];                  /* c3.y */
    a.illum_march_steps = i[14];                /* c3.z */
    a.sigma_scattering = f[15];                 /* c3.w */
    a.cld_coverage = f[16];                     /* c4.x */
    a.cld_thick = f[17];                        /* c4.y */
    a.atm_radius = f[18];                       /* c4.z */
    a.atm_ground_y = f[19];                     /* c4.w */
    return a;
}
static sdf_ao_aux_t parse_sdf_ao_aux(const void* aux) {
    sdf_ao_aux_t a;
    if (!aux) return a;
    const float* f = (const float*)aux;
    a.fog_density = f[0];
    a.fog_falloff = f[1];
    return a;
}

static bool pixel(int app, const uniforms_t& U, const void* aux, float fx, float fy, float* out) {
    vec4 c;
    vec2 fc(fx, fy);
    switch (app) {
    case APP_EGG: { AppEgg a; a.U = U; c = main_image(a, fc); break; }
    case APP_CLOUDS: { AppClouds a; a.U = U; a.A = parse_clouds_aux(aux); c = main_image(a, fc); break; }
    case APP_CLOUDS_SKY: { AppClouds a; a.U = U; a.A = parse_clouds_aux(aux); a.sky_sphere = true; c = main_image(a, fc); break; }
    case APP_CLOUDS_TEX: {
        if (!g_tex_noise.rgba || !g_tex_noise_2.rgba) return false;
        AppClouds a; a.U = U; a.A = parse_clouds_aux(aux); a.tex_noise = g_tex_noise; a.tex_noise_2 = g_tex_noise_2;
        c = main_image(a, fc); break; }
    case APP_RAYTRACER: { AppRaytracer a; a.U = U; c = main_image(a, fc); break; }
    case APP_ATMOSPHERE: { AppAtmosphere a; a.U = U; c = main_image(a, fc); break; }
    case APP_SDF_AO: { AppSdfAo a; a.U = U; a.A = parse_sdf_ao_aux(aux); c = main_image(a, fc); break; }
    case APP_PLANET: { AppPlanet a; a.U = U; c = main_image(a, fc); break; }
    case APP_PLANET_ATMOSPHERE: { AppPlanet a; a.U = U; a.atm_sky = true; c = main_image(a, fc); break; }
    case APP_VINYL: { AppVinyl a; a.U = U; c = main_image(a, fc); break; }
    case APP_VINYL_GPU: { AppVinyl a; a.U = U; a.march_steps = 180; c = main_image(a, fc); break; }
    case APP_CLOUDS_BEST: { AppCloudsBest a; a.U = U; c = main_image(a, fc); break; }
    case APP_CLOUDS_UE4: {
        AppCloudsUe4 a; a.U = U;
        if (aux) {                                  /* sbx_aux_clouds_ue4: coverage, thickness, absorbtion, fuzziness, sun_dir@c1, wind_dir@c2, use_dirs@c2.w */
            const float* f = (const float*)aux;
            a.A.coverage = f[0]; a.A.thickness = f[1]; a.A.absorbtion = f[2]; a.A.fuzziness = f[3];
            a.A.sun_dir = vec3(f[4], f[5], f[6]); a.A.wind_dir = vec3(f[8], f[9], f[10]);
            a.A.has_dirs = ((const int*)aux)[11] != 0;
        }
        c = main_image(a, fc); break; }
    default: return false;
    }
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
    return true;
}

extern "C" {

/* Bind the RGBA32F size^3 volumes (x fastest, as util/ddsvolgen writes them) to t1 / t2 for APP_CLOUDS_TEX.
 * The pointers are kept, not copied. */
int sbxo_set_noise_volumes(int size1, const float* rgba1, int size2, const float* rgba2) {
    g_tex_noise.rgba = rgba1; g_tex_noise.size = size1;
    g_tex_noise_2.rgba = rgba2; g_tex_noise_2.size = size2;
    return 0;
}
/* SampleLevel(linear, wrap, 0).r of a volume at n points (xyz interleaved): the texture-filter spec on its own */
int sbxo_tex3d(int size, const float* rgba, const float* xyz, float* out, long n) {
    noise_tex_t T; T.rgba = rgba; T.size = size;
    for (long i = 0; i < n; ++i) out[i] = tex3d_sample_r(T, vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
    return 0;
}

/* mainImage for one pixel. uniforms = {u_res.x, u_res.y, u_mouse.x, u_mouse.y, u_time} */
int sbxo_main_image(int app, const float* uniforms, const void* aux, float fx, float fy, float* rgba) {
    uniforms_t U; U.u_res = vec2(uniforms[0], uniforms[1]); U.u_mouse = vec2(uniforms[2], uniforms[3]); U.u_time = uniforms[4];
    return pixel(app, U, aux, fx, fy, rgba) ? 0 : -1;
}

/* Render the listed rows (global row indices, 0 = bottom) of a W x H frame into
 * out[nrows][W][4]; the work is dealt to `nthreads` std::threads through an atomic counter in tiles of 64 pixels of
 * one row (dealing whole rows leaves most of a many-core host idle behind the last few expensive rows). */
int sbxo_render_rows(int app, const float* uniforms, const void* aux, const int* rows, int nrows,
                     float* out, int nthreads) {
    uniforms_t U; U.u_res = vec2(uniforms[0], uniforms[1]); U.u_mouse = vec2(uniforms[2], uniforms[3]); U.u_time = uniforms[4];
    const int W = (int)U.u_res.x;
    float probe[4];
    if (!pixel(app, U, aux, .5f, .5f, probe)) return -1;
    if (nthreads < 1) nthreads = 1;
    const int TILE = 64;
    const int tiles_x = (W + TILE - 1) / TILE;
    const long ntiles = (long)nrows * tiles_x;
    std::atomic<long> next(0);
    auto work = [&]() {
        for (;;) {
            const long i = next.fetch_add(1);
            if (i >= ntiles) break;
            const int r = (int)(i / tiles_x), x0 = (int)(i % tiles_x) * TILE;
            const int x1 = x0 + TILE < W ? x0 + TILE : W;
            const int y = rows[r];
            float* dst = out + (size_t)r * W * 4;
            for (int x = x0; x < x1; ++x) pixel(app, U, aux, (float)x + .5f, (float)y + .5f, dst + 4 * x);
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < nthreads; ++i) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    return 0;
}

/* scalar math spec, elementwise: fn in {sin,cos,tan,exp,pow,acos,atan2} */
int sbxo_math(const char* fn, const float* a, const float* b, float* out, long n) {
    std::string f(fn);
    for (long i = 0; i < n; ++i) {
        if (f == "sin") out[i] = m_sin(a[i]);
        else if (f == "cos") out[i] = m_cos(a[i]);
        else if (f == "tan") out[i] = m_tan(a[i]);
        else if (f == "exp") out[i] = m_exp(a[i]);
        else if (f == "pow") out[i] = m_pow(a[i], b[i]);
        else if (f == "acos") out[i] = m_acos(a[i]);
        else if (f == "atan2") out[i] = m_atan2(a[i], b[i]);
        else return -1;
    }
    return 0;
}

/* elementwise library noise functions over n points (xyz interleaved); out has 3 floats per point.
 * fn: "noise_iq" (out[0]), "hash_w", "noise_w" (par[0] = domain_repeat), "fbm_worley_tile"
 * (par = lacunarity, init_gain, gain; out[0]) */
int sbxo_noise(const char* fn, const float* xyz, const float* par, float* out, long n) {
    std::string f(fn);
    for (long i = 0; i < n; ++i) {
        vec3 p(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        vec3 r(0, 0, 0);
        if (f == "noise_iq") r.x = noise_iq(p);
        else if (f == "hash_w") r = hash_w(p);
        else if (f == "noise_w") r = noise_w(p, par[0]);
        else if (f == "fbm_worley_tile") r.x = fbm_worley_tile(p, par[0], par[1], par[2]);
        else return -1;
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
    return 0;
}

/* The noise volume of util/ddsvolgen (/root/reference/util/ddsvolgen/src/ddsvolgen.cpp:101-117):
 * size^3 voxels, RGBA32F, R = fbm_worley_tile((xyz + .5)/size, 2, 1, .5) (:52-61), G = B = A = 0,
 * x fastest.  Slices [z0, z1). */
int sbxo_worley_volume(int size, int z0, int z1, float* out) {
    for (int z = z0; z < z1; ++z)
        for (int y = 0; y < size; ++y)
            for (int x = 0; x < size; ++x) {
                vec3 pos = (vec3((float)x, (float)y, (float)z) + .5f) / (float)size;
                float* o = out + (((size_t)(z - z0) * size + y) * size + x) * 4;
                o[0] = fbm_worley_tile(pos, 2.f, 1.f, .5f);
                o[1] = o[2] = o[3] = 0.f;
            }
    return 0;
}

/* Known-answer hooks: evaluate one library/app function on explicit arguments.
 * `in` / `out` are flat float arrays; meaning per name is listed in tests/test_oracle_kat.py.
 * Uniform-dependent functions take {W, H, mouse.x, mouse.y, time} as the first 5 inputs. */
int sbxo_kat(const char* name, const float* in, float* out) {
    std::string f(name);
    auto V3 = [&](int o) { return vec3(in[o], in[o + 1], in[o + 2]); };
    auto put3 = [&](vec3 v, int o = 0) { out[o] = v.x; out[o + 1] = v.y; out[o + 2] = v.z; };
    auto uni = [&]() { uniforms_t U; U.u_res = vec2(in[0], in[1]); U.u_mouse = vec2(in[2], in[3]); U.u_time = in[4]; return U; };

    if (f == "hash") { out[0] = hash(in[0]); return 0; }
    if (f == "snoise") { out[0] = snoise(V3(0)); return 0; }
    if (f == "clouds_best.fbm_clouds") { out[0] = AppCloudsBest::fbm_clouds(V3(0), in[3], in[4], in[5]); return 0; }
    if (f == "noise_iq") { out[0] = noise_iq(V3(0)); return 0; }
    if (f == "clouds.fbm") { out[0] = AppClouds::fbm(V3(0), in[3], in[4], in[5]); return 0; }
    if (f == "clouds.density_func") { AppClouds a; out[0] = a.density_func(V3(0), in[3]); return 0; }
    if (f == "hg") { out[0] = henyey_greenstein_phase_func(in[0], in[1]); return 0; }
    if (f == "rayleigh") { out[0] = rayleigh_phase_func(in[0]); return 0; }
    if (f == "clouds.illuminate_volume") { AppClouds a; out[0] = a.illuminate_volume(V3(0), in[3], V3(4), V3(7)); return 0; }
    if (f == "clouds.render_sky_color") { AppClouds a; put3(a.render_sky_color(V3(0))); return 0; }
    if (f == "primary_ray") { /* in: app, W, H, mx, my, t, pc.x, pc.y, pc.z -> dir, origin */
        int app = (int)in[0];
        uniforms_t U; U.u_res = vec2(in[1], in[2]); U.u_mouse = vec2(in[3], in[4]); U.u_time = in[5];
        vec3 eye, look_at;
        switch (app) {
        case APP_EGG: { AppEgg a; a.U = U; a.setup_camera(eye, look_at); break; }
        case APP_CLOUDS: { AppClouds a; a.U = U; a.setup_camera(eye, look_at); break; }
        case APP_RAYTRACER: { AppRaytracer a; a.U = U; a.setup_camera(eye, look_at); break; }
        case APP_ATMOSPHERE: { AppAtmosphere a; a.U = U; a.setup_camera(eye, look_at); break; }
        case APP_SDF_AO: { AppSdfAo a; a.U = U; a.setup_camera(eye, look_at); break; }
        case APP_PLANET: { AppPlanet a; a.U = U; a.setup_camera(eye, look_at); break; }
        default: return -1;
        }
        ray_t r = get_primary_ray(V3(6), eye, look_at);
        put3(r.direction, 0); put3(r.origin, 3);
        return 0;
    }
    if (f == "egg.sdf") { AppEgg a; a.U = uni(); vec2 d = a.sdf(V3(5)); out[0] = d.x; out[1] = d.y; return 0; }
    if (f == "ik_solver") { put3(ik_solver(V3(0), V3(3), in[6], in[7])); return 0; }
    if (f == "sd_bezier") { vec2 d = sd_bezier(V3(0), V3(3), V3(6), V3(9), in[12]); out[0] = d.x; out[1] = d.y; return 0; }
    if (f == "sd_cylinder") { out[0] = sd_cylinder(V3(0), V3(3), V3(6), in[9]); return 0; }
    if (f == "sd_torus") { out[0] = sd_torus(V3(0), in[3], in[4]); return 0; }
    if (f == "op_blend") { out[0] = op_blend(in[0], in[1], in[2]); return 0; }
    if (f == "sdf_ao.sdf") { AppSdfAo a; a.U = uni(); a.setup_scene(); vec2 d = a.sdf(V3(5)); out[0] = d.x; out[1] = d.y; return 0; }
    if (f == "sdf_ao.sdf_normal") { AppSdfAo a; a.U = uni(); a.setup_scene(); put3(a.sdf_normal(V3(5))); return 0; }
    if (f == "sdf_ao.sdf_ao") { /* in: uniforms5, normal3, origin3 */
        AppSdfAo a; a.U = uni(); a.setup_scene();
        hit_t h; h.t = 1; h.material_id = 1; h.normal = V3(5); h.origin = V3(8);
        put3(a.sdf_ao(h)); return 0; }
    if (f == "sdf_ao.illuminate") { /* in: uniforms5, normal3, origin3, ao, sh ; h = {1, 1, n, o} */
        AppSdfAo a; a.U = uni(); a.setup_scene();
        vec3 eye, la; a.setup_camera(eye, la);
        hit_t h; h.t = 1; h.material_id = 1; h.normal = V3(5); h.origin = V3(8);
        put3(a.illuminate(eye, h, in[11], in[12])); return 0; }
    if (f == "sdf_ao.render_impl") { /* in: uniforms5, point_cam3 */
        AppSdfAo a; a.U = uni(); a.setup_scene();
        vec3 eye, la; a.setup_camera(eye, la);
        ray_t r = get_primary_ray(V3(5), eye, la);
        vec4 c = a.render_impl(r, V3(5)); out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w; return 0; }
    if (f == "raytracer.left_sphere") { AppRaytracer a; a.U = uni(); a.setup_scene(); put3(a.cb_spheres[1].origin); return 0; }
    if (f == "raytracer.raytrace_iteration") { /* in: uniforms5, point_cam3 -> t, mat, n3, o3 */
        AppRaytracer a; a.U = uni(); a.setup_scene();
        vec3 eye, la; a.setup_camera(eye, la);
        ray_t r = get_primary_ray(V3(5), eye, la);
        hit_t h = a.raytrace_iteration(r, -1);
        out[0] = h.t; out[1] = (float)h.material_id; put3(h.normal, 2); put3(h.origin, 5); return 0; }
    if (f == "raytracer.illuminate") { /* same ray: illuminate(eye, hit) */
        AppRaytracer a; a.U = uni(); a.setup_scene();
        vec3 eye, la; a.setup_camera(eye, la);
        ray_t r = get_primary_ray(V3(5), eye, la);
        hit_t h = a.raytrace_iteration(r, -1);
        put3(a.illuminate(eye, h)); return 0; }
    if (f == "raytracer.render") {
        AppRaytracer a; a.U = uni(); a.setup_scene();
        vec3 eye, la; a.setup_camera(eye, la);
        ray_t r = get_primary_ray(V3(5), eye, la);
        put3(a.render(r, V3(5))); return 0; }
    if (f == "fresnel_factor") { out[0] = fresnel_factor(in[0], in[1], in[2]); return 0; }
    /* "@2" variants apply setup_scene twice: SURVEY.md App. C's function-level ATMOSPHERE numbers
     * were taken in that state (its printed sun_dir is the doubled rotation). */
    if (f == "atmosphere.get_incident_light@2") {
        AppAtmosphere a; a.U = uni(); a.setup_scene(); a.setup_scene();
        ray_t r; r.origin = V3(5); r.direction = V3(8);
        put3(a.get_incident_light(r)); return 0; }
    if (f == "atmosphere.render@2") {
        AppAtmosphere a; a.U = uni(); a.setup_scene(); a.setup_scene();
        ray_t r; put3(a.render(r, V3(5))); return 0; }
    if (f == "atmosphere.sun_dir") { AppAtmosphere a; a.U = uni(); a.setup_scene(); put3(a.sun_dir); return 0; }
    if (f == "atmosphere.get_sun_light") { /* in: uniforms5, origin3, dir3 -> ok, odR, odM */
        AppAtmosphere a; a.U = uni(); a.setup_scene();
        ray_t r; r.origin = V3(5); r.direction = V3(8);
        float odr = 0, odm = 0; bool ok = a.get_sun_light(r, odr, odm);
        out[0] = ok ? 1.f : 0.f; out[1] = odr; out[2] = odm; return 0; }
    if (f == "atmosphere.get_incident_light") {
        AppAtmosphere a; a.U = uni(); a.setup_scene();
        ray_t r; r.origin = V3(5); r.direction = V3(8);
        put3(a.get_incident_light(r)); return 0; }
    if (f == "atmosphere.render") { /* in: uniforms5, point_cam3 */
        AppAtmosphere a; a.U = uni(); a.setup_scene();
        ray_t r; put3(a.render(r, V3(5))); return 0; }
    if (f == "hash_w") { put3(hash_w(V3(0))); return 0; }
    if (f == "noise_w") { put3(noise_w(V3(0), in[3])); return 0; }
    if (f == "fbm_worley_tile") { out[0] = fbm_worley_tile(V3(0), in[3], in[4], in[5]); return 0; }
    if (f == "planet.sdf_terrain_map") { vec2 d = AppPlanet::sdf_terrain_map(V3(0)); out[0] = d.x; out[1] = d.y; return 0; }
    if (f == "planet.sdf_terrain_normal") { put3(AppPlanet::sdf_terrain_normal(V3(0))); return 0; }
    return -1;
}

} /* extern "C" */
