// host/sbx_render.cpp — C++ host over the C ABI: renders a frame or an animation on one or several MI355X and writes
// every frame as binary PPM (sRGB 8-bit, top row first) and/or raw float32 RGBA (row 0 = bottom).
//
//   sbx_render --app clouds --res 3840x2160 [--time 0.37] [--frames N --dt S] [--mouse X,Y] [--gpus N [--exchange spans|slabs|blocks]]
//              [--ppm out_%04d.ppm] [--f32 out_%04d.f32] [--rgba8]
//     --rgba8: the kernels write the 8-bit display format themselves (SBX_FORMAT_RGBA8, include/sbx.h): 4 bytes per pixel in
//              HBM, in the multi-GPU exchange and over PCIe; the same .ppm bytes as the float frame packed afterwards; no --f32
//     APP_CLOUDS aux block (the ImGui panel of hlsltoy, util/hlsltoy/src/hlsltoy.cpp:466-483):
//              [--wind x,y,z] [--sun x,y,z] [--sun-color r,g,b] [--sun-power P] [--sky-radius R] [--sky-height Y]
//              [--sigma S] [--coverage C] [--thick T] [--steps N] [--light-steps N]
//     APP_SDF_AO aux block (:484-487):  [--fog-density D] [--fog-falloff F]
//     APP_CLOUDS with USE_NOISE_TEX (app "clouds_tex"; hlsltoy's argv[2], argv[3], hlsltoy.cpp:227-238):
//              --noise-tex shape.dds,detail.dds   (DX10 RGBA32F volume .dds as util/ddsvolgen / sbx_ddsvolgen write)
//              --noise-tex bake:128,64            (bake the two volumes here with sbx_worley_volume)
//
// Plays the role of the reference's frame-granular hosts (hlsltoy.cpp:494-516: draw, advance u_time, re-upload the
// uniform blocks): frame f is rendered at u_time = time + f * dt with the aux values given on the command line, and
// every frame of --frames N is written (a printf pattern in the file name numbers the frames; without one "_%04d" is put
// before the extension when N > 1).  --gpus N shards every frame over N GPUs inside the library (sbx_multi_*).
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/sbx.h"

static int app_from_name(const std::string& s) {
    const char* names[] = {"planet", "clouds", "vinyl", "egg", "raytracer", "atmosphere", "sdf_ao", "clouds_best", "clouds_tex", "clouds_ue4",
                           "clouds_sky", "vinyl_gpu", "planet_atmosphere"};
    const int n = 13;
    std::string low;
    for (char c : s) low += (char)tolower(c);
    for (int i = 0; i < n; ++i)
        if (low == names[i] || low == std::string("app_") + names[i]) return i;
    return -1;
}

// A file-name pattern may hold ONE integer conversion (%d, %4d, %04d ...) and any number of "%%"; anything else (%s, %n, a
// second %d) would make the user's string an snprintf format with undefined behaviour, so it is rejected up front.
static bool pattern_ok(const std::string& p) {
    int conversions = 0;
    for (size_t i = 0; i < p.size(); ++i) {
        if (p[i] != '%') continue;
        if (i + 1 < p.size() && p[i + 1] == '%') { ++i; continue; }
        size_t j = i + 1;
        while (j < p.size() && p[j] >= '0' && p[j] <= '9') ++j;      // flag 0 and a width
        if (j >= p.size() || p[j] != 'd' || j - i > 4) return false;
        ++conversions;
        i = j;
    }
    return conversions <= 1;
}
static std::string frame_name(const std::string& pattern, int f, int frames) {
    char buf[4096];
    if (pattern.find('%') != std::string::npos) { snprintf(buf, sizeof(buf), pattern.c_str(), f); return buf; }   // checked in main
    if (frames <= 1) return pattern;
    const size_t dot = pattern.rfind('.');
    snprintf(buf, sizeof(buf), "%s_%04d%s", pattern.substr(0, dot).c_str(), f, dot == std::string::npos ? "" : pattern.substr(dot).c_str());
    return buf;
}

// a DX10 volume .dds with RGBA32F texels (what ddsvolgen writes): magic, 124-byte header, 20-byte DX10 header, texels
static bool read_dds_volume(const std::string& path, int& size, std::vector<float>& rgba) {
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) { perror(path.c_str()); return false; }
    uint32_t head[37];                                        // 4 + 124 + 20 bytes
    bool ok = fread(head, 4, 37, fp) == 37 && head[0] == 0x20534444u && head[1] == 124 && head[21] == 0x30315844u /* DX10 */ &&
              head[32] == 2 /* R32G32B32A32_FLOAT */ && head[33] == 4 /* TEXTURE3D */;
    if (ok) {
        const uint32_t h = head[3], w = head[4], d = head[6];
        ok = h == w && w == d && w > 0 && w <= 1024;
        if (ok) {
            size = (int)w;
            rgba.resize((size_t)w * w * w * 4);
            ok = fread(rgba.data(), 16, (size_t)w * w * w, fp) == (size_t)w * w * w;
        }
    }
    fclose(fp);
    if (!ok) fprintf(stderr, "%s: not a cubic RGBA32F DX10 volume .dds\n", path.c_str());
    return ok;
}

int main(int argc, char** argv) {
    std::string app = "clouds", ppm, f32, noise_tex;
    int W = 1280, H = 720, frames = 1, gpus = 1, exchange = SBX_MULTI_EXCHANGE_SPANS;
    float t = 0.37f, dt = 1.f / 30.f, mx = 0, my = 0;
    sbx_aux_clouds ac;
    sbx_aux_sdf_ao as;
    sbx_aux_clouds_defaults(&ac);
    sbx_aux_sdf_ao_defaults(&as);
    bool have_ac = false, have_as = false, rgba8_out = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        auto vec3 = [&](float* v) { if (sscanf(next(), "%f,%f,%f", v, v + 1, v + 2) != 3) { fprintf(stderr, "%s x,y,z\n", a.c_str()); exit(2); } };
        if (a == "--app") app = next();
        else if (a == "--res") { if (sscanf(next(), "%dx%d", &W, &H) != 2) { fprintf(stderr, "--res WxH\n"); return 2; } }
        else if (a == "--time") t = (float)atof(next());
        else if (a == "--dt") dt = (float)atof(next());
        else if (a == "--frames") frames = atoi(next());
        else if (a == "--gpus") gpus = atoi(next());
        else if (a == "--exchange") {
            const std::string v = next();
            exchange = v == "slabs" ? SBX_MULTI_EXCHANGE_SLABS : v == "blocks" ? SBX_MULTI_EXCHANGE_BLOCKS : v == "spans" ? SBX_MULTI_EXCHANGE_SPANS : -1;
            if (exchange < 0) { fprintf(stderr, "--exchange slabs|blocks|spans\n"); return 2; }
        }
        else if (a == "--mouse") { if (sscanf(next(), "%f,%f", &mx, &my) != 2) { fprintf(stderr, "--mouse X,Y\n"); return 2; } }
        else if (a == "--ppm") ppm = next();
        else if (a == "--f32") f32 = next();
        else if (a == "--rgba8") rgba8_out = true;
        else if (a == "--noise-tex") noise_tex = next();
        else if (a == "--wind") { vec3(ac.wind_dir); have_ac = true; }
        else if (a == "--sun") { vec3(ac.sun_dir); have_ac = true; }
        else if (a == "--sun-color") { vec3(ac.sun_color); have_ac = true; }
        else if (a == "--sun-power") { ac.sun_power = (float)atof(next()); have_ac = true; }
        else if (a == "--sky-radius") { ac.atm_radius = (float)atof(next()); have_ac = true; }
        else if (a == "--sky-height") { ac.atm_ground_y = (float)atof(next()); have_ac = true; }
        else if (a == "--sigma") { ac.sigma_scattering = (float)atof(next()); have_ac = true; }
        else if (a == "--coverage") { ac.cld_coverage = (float)atof(next()); have_ac = true; }
        else if (a == "--thick") { ac.cld_thick = (float)atof(next()); have_ac = true; }
        else if (a == "--steps") { ac.cld_march_steps = atoi(next()); have_ac = true; }
        else if (a == "--light-steps") { ac.illum_march_steps = atoi(next()); have_ac = true; }
        else if (a == "--fog-density") { as.fog_density = (float)atof(next()); have_as = true; }
        else if (a == "--fog-falloff") { as.fog_falloff = (float)atof(next()); have_as = true; }
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    const int id = app_from_name(app);
    if (id < 0) { fprintf(stderr, "unknown app %s\n", app.c_str()); return 2; }
    if (frames < 1 || gpus < 1) { fprintf(stderr, "--frames and --gpus must be >= 1\n"); return 2; }
    if (rgba8_out && !f32.empty()) { fprintf(stderr, "--rgba8 renders 8-bit pixels: there is no float frame for --f32\n"); return 2; }
    for (const std::string* p : {&ppm, &f32})
        if (!pattern_ok(*p)) { fprintf(stderr, "bad file pattern %s: one %%d / %%0Nd conversion at most, a literal percent as %%%%\n", p->c_str()); return 2; }
    const void* aux = nullptr;
    if ((id == SBX_APP_CLOUDS || id == SBX_APP_CLOUDS_TEX || id == SBX_APP_CLOUDS_SKY) && have_ac) aux = &ac;
    if (id == SBX_APP_SDF_AO && have_as) aux = &as;

    sbx_ctx* ctx = nullptr;
    int rc = sbx_create(0, &ctx);
    if (rc != SBX_OK) { fprintf(stderr, "sbx_create failed (%d): a gfx950 GPU is required, there is no CPU path\n", rc); return 1; }
    sbx_multi* multi = nullptr;
    if (gpus > 1) {
        int ndev = 0;
        (void)hipGetDeviceCount(&ndev);
        std::vector<int> devs(gpus);
        for (int i = 0; i < gpus; ++i) devs[i] = ndev >= gpus ? i : i % (ndev > 0 ? ndev : 1);
        if (ndev < gpus) fprintf(stderr, "note: %d GPU(s) visible, running the %d-rank schedule with ranks sharing devices\n", ndev, gpus);
        rc = sbx_multi_create(gpus, devs.data(), &multi);
        if (rc != SBX_OK) { fprintf(stderr, "sbx_multi_create failed (%d): %s\n", rc, sbx_multi_create_error()); return 1; }
        if ((rc = sbx_multi_set_exchange(multi, exchange)) != SBX_OK) { fprintf(stderr, "exchange: %s\n", sbx_multi_last_error(multi)); return 1; }
        printf("%d ranks, transfers by %s, exchange %s\n", gpus, sbx_multi_uses_rccl(multi) ? "RCCL send/recv" : "device copies",
               exchange == SBX_MULTI_EXCHANGE_SPANS ? "spans" : exchange == SBX_MULTI_EXCHANGE_BLOCKS ? "blocks" : "slabs");
    }
    (void)hipSetDevice(0);
    if (id == SBX_APP_CLOUDS_TEX) {
        if (noise_tex.empty()) { fprintf(stderr, "app clouds_tex needs --noise-tex shape.dds,detail.dds or --noise-tex bake:S1,S2\n"); return 2; }
        int s1 = 0, s2 = 0;
        float *d1 = nullptr, *d2 = nullptr;
        if (noise_tex.rfind("bake:", 0) == 0) {
            if (sscanf(noise_tex.c_str() + 5, "%d,%d", &s1, &s2) != 2 || s1 <= 0 || s2 <= 0) { fprintf(stderr, "--noise-tex bake:S1,S2\n"); return 2; }
            if (hipMalloc((void**)&d1, (size_t)s1 * s1 * s1 * 16) != hipSuccess || hipMalloc((void**)&d2, (size_t)s2 * s2 * s2 * 16) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
            if (sbx_worley_volume(ctx, s1, d1, nullptr) != SBX_OK || sbx_worley_volume(ctx, s2, d2, nullptr) != SBX_OK) { fprintf(stderr, "sbx_worley_volume: %s\n", sbx_last_error(ctx)); return 1; }
        } else {
            const size_t comma = noise_tex.find(',');
            if (comma == std::string::npos) { fprintf(stderr, "--noise-tex shape.dds,detail.dds\n"); return 2; }
            std::vector<float> v1, v2;
            if (!read_dds_volume(noise_tex.substr(0, comma), s1, v1) || !read_dds_volume(noise_tex.substr(comma + 1), s2, v2)) return 1;
            if (hipMalloc((void**)&d1, v1.size() * 4) != hipSuccess || hipMalloc((void**)&d2, v2.size() * 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
            (void)hipMemcpy(d1, v1.data(), v1.size() * 4, hipMemcpyHostToDevice);
            (void)hipMemcpy(d2, v2.data(), v2.size() * 4, hipMemcpyHostToDevice);
        }
        (void)hipDeviceSynchronize();
        rc = multi ? sbx_multi_set_noise_volumes(multi, s1, d1, s2, d2) : sbx_set_noise_volumes(ctx, s1, d1, s2, d2, nullptr);
        if (rc != SBX_OK) { fprintf(stderr, "noise volumes: %s\n", multi ? sbx_multi_last_error(multi) : sbx_last_error(ctx)); return 1; }
        (void)hipDeviceSynchronize();
        (void)hipFree(d1); (void)hipFree(d2);
    }
    if (rgba8_out) {
        rc = multi ? sbx_multi_set_output_format(multi, SBX_FORMAT_RGBA8) : sbx_set_output_format(ctx, SBX_FORMAT_RGBA8);
        if (rc != SBX_OK) { fprintf(stderr, "output format: %s\n", multi ? sbx_multi_last_error(multi) : sbx_last_error(ctx)); return 1; }
    }
    // the frame, the 8-bit copy and the timing events belong to rank 0's device (the sbx_multi_* calls leave it current;
    // set again here so that nothing below depends on that)
    (void)hipSetDevice(0);
    const size_t n = (size_t)W * H * 4, npx = (size_t)W * H;
    float* dev = nullptr;
    unsigned char* dev8 = nullptr;
    if (hipMalloc((void**)&dev, (rgba8_out ? npx : n) * sizeof(float)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }   // (--rgba8: one word per pixel)
    if (!ppm.empty() && !rgba8_out && hipMalloc((void**)&dev8, npx * 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    std::vector<float> host(f32.empty() ? 0 : n);
    std::vector<unsigned char> rgba8(ppm.empty() ? 0 : npx * 4), rgb(ppm.empty() ? 0 : npx * 3);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int f = 0; f < frames; ++f) {
        sbx_uniforms u{};
        u.u_res[0] = (float)W; u.u_res[1] = (float)H; u.u_mouse[0] = mx; u.u_mouse[1] = my; u.u_time = t + f * dt;   // hlsltoy.cpp:502-504
        (void)hipEventRecord(e0, nullptr);
        rc = multi ? sbx_multi_render(multi, id, &u, aux, dev, nullptr) : sbx_render_rows(ctx, id, &u, aux, 0, H, dev, nullptr);
        if (rc != SBX_OK) { fprintf(stderr, "render: %s\n", multi ? sbx_multi_last_error(multi) : sbx_last_error(ctx)); return 1; }
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("frame %d  u_time %.4f  %s %dx%d  %.3f ms  %.1f Mpix/s\n", f, u.u_time, app.c_str(), W, H, ms, W * (double)H / ms / 1e3);
        if (!f32.empty()) {
            if (hipMemcpy(host.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "hipMemcpy failed\n"); return 1; }
            FILE* fp = fopen(frame_name(f32, f, frames).c_str(), "wb");
            if (!fp) { perror("fopen"); return 1; }
            fwrite(host.data(), sizeof(float), n, fp);
            fclose(fp);
        }
        if (!ppm.empty()) {
            // the back-buffer write of hlsltoy (R8G8B8A8_UNORM, top row first) on the device, then 4 B/pixel over PCIe
            if (rgba8_out) {
                // the kernels wrote this format already (row 0 = bottom): 4 B/pixel over PCIe, the rows turned here
                if (hipMemcpy(rgba8.data(), dev, npx * 4, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "hipMemcpy failed\n"); return 1; }
                for (int y = 0; y < H; ++y) {
                    const unsigned char* src = &rgba8[(size_t)(H - 1 - y) * W * 4];
                    unsigned char* dst = &rgb[(size_t)y * W * 3];
                    for (int x = 0; x < W; ++x) { dst[x * 3] = src[x * 4]; dst[x * 3 + 1] = src[x * 4 + 1]; dst[x * 3 + 2] = src[x * 4 + 2]; }
                }
            } else {
            rc = sbx_pack_unorm8(ctx, W, H, dev, dev8, /*flip_y=*/1, nullptr);
            if (rc != SBX_OK) { fprintf(stderr, "sbx_pack_unorm8: %s\n", sbx_last_error(ctx)); return 1; }
            if (hipMemcpy(rgba8.data(), dev8, npx * 4, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "hipMemcpy failed\n"); return 1; }
            for (size_t i = 0; i < npx; ++i) { rgb[i * 3] = rgba8[i * 4]; rgb[i * 3 + 1] = rgba8[i * 4 + 1]; rgb[i * 3 + 2] = rgba8[i * 4 + 2]; }
            }
            FILE* fp = fopen(frame_name(ppm, f, frames).c_str(), "wb");
            if (!fp) { perror("fopen"); return 1; }
            fprintf(fp, "P6\n%d %d\n255\n", W, H);
            fwrite(rgb.data(), 1, rgb.size(), fp);
            fclose(fp);
        }
    }
    if (dev8) (void)hipFree(dev8);
    (void)hipFree(dev);
    if (multi) sbx_multi_destroy(multi);
    sbx_destroy(ctx);
    return 0;
}
