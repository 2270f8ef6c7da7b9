// host/sbx_render.cpp — C++ host over the C ABI: renders one frame (or an animation) on an MI355X and
// writes it as binary PPM (sRGB 8-bit, top row first) and/or raw float32 RGBA (row 0 = bottom).
//
//   sbx_render --app clouds --res 3840x2160 --time 0.37 [--mouse X,Y] [--frames N --dt S] [--ppm out.ppm] [--f32 out.f32]
//
// Plays the role of the reference's frame-granular hosts (util/hlsltoy/src/hlsltoy.cpp:494-516: draw,
// advance u_time, re-upload uniforms) with the aux uniforms at their reference defaults.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/sbx.h"

static int app_from_name(const std::string& s) {
    const char* names[] = {"planet", "clouds", "vinyl", "egg", "raytracer", "atmosphere", "sdf_ao", "clouds_best"};
    const int n = 8;
    for (int i = 0; i < n; ++i)
        if (s == names[i] || s == std::string("APP_") + names[i]) return i;
    std::string up;
    for (char c : s) up += (char)tolower(c);
    for (int i = 0; i < n; ++i)
        if (up == names[i] || up == std::string("app_") + names[i]) return i;
    return -1;
}

int main(int argc, char** argv) {
    std::string app = "clouds", ppm, f32;
    int W = 1280, H = 720, frames = 1;
    float t = 0.37f, dt = 1.f / 30.f, mx = 0, my = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--app") app = next();
        else if (a == "--res") { if (sscanf(next(), "%dx%d", &W, &H) != 2) { fprintf(stderr, "--res WxH\n"); return 2; } }
        else if (a == "--time") t = (float)atof(next());
        else if (a == "--dt") dt = (float)atof(next());
        else if (a == "--frames") frames = atoi(next());
        else if (a == "--mouse") { if (sscanf(next(), "%f,%f", &mx, &my) != 2) { fprintf(stderr, "--mouse X,Y\n"); return 2; } }
        else if (a == "--ppm") ppm = next();
        else if (a == "--f32") f32 = next();
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    const int id = app_from_name(app);
    if (id < 0) { fprintf(stderr, "unknown app %s\n", app.c_str()); return 2; }
    sbx_ctx* ctx = nullptr;
    int rc = sbx_create(0, &ctx);
    if (rc != SBX_OK) { fprintf(stderr, "sbx_create failed (%d): a gfx950 GPU is required, there is no CPU path\n", rc); return 1; }
    const size_t n = (size_t)W * H * 4;
    float* dev = nullptr;
    if (hipMalloc((void**)&dev, n * sizeof(float)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    std::vector<float> host(n);
    sbx_set_timing(ctx, 1);
    for (int f = 0; f < frames; ++f) {
        sbx_uniforms u{};
        u.u_res[0] = (float)W; u.u_res[1] = (float)H; u.u_mouse[0] = mx; u.u_mouse[1] = my; u.u_time = t + f * dt;
        rc = sbx_render_rows(ctx, id, &u, nullptr, 0, H, dev, nullptr);
        if (rc != SBX_OK) { fprintf(stderr, "sbx_render_rows: %s\n", sbx_last_error(ctx)); return 1; }
        float ms = 0;
        sbx_last_kernel_ms(ctx, &ms);
        printf("frame %d  u_time %.4f  %s %dx%d  %.3f ms  %.1f Mpix/s\n", f, u.u_time, app.c_str(), W, H, ms, W * (double)H / ms / 1e3);
    }
    if (hipMemcpy(host.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "hipMemcpy failed\n"); return 1; }
    if (!f32.empty()) {
        FILE* fp = fopen(f32.c_str(), "wb");
        if (!fp) { perror("fopen"); return 1; }
        fwrite(host.data(), sizeof(float), n, fp);
        fclose(fp);
    }
    if (!ppm.empty()) {
        FILE* fp = fopen(ppm.c_str(), "wb");
        if (!fp) { perror("fopen"); return 1; }
        fprintf(fp, "P6\n%d %d\n255\n", W, H);
        // the back-buffer write of hlsltoy (R8G8B8A8_UNORM, top row first) on the device, then 4 B/pixel over PCIe
        unsigned char* dev8 = nullptr;
        const size_t npx = (size_t)W * H;
        if (hipMalloc((void**)&dev8, npx * 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
        rc = sbx_pack_unorm8(ctx, W, H, dev, dev8, /*flip_y=*/1, nullptr);
        if (rc != SBX_OK) { fprintf(stderr, "sbx_pack_unorm8: %s\n", sbx_last_error(ctx)); return 1; }
        std::vector<unsigned char> rgba8(npx * 4), rgb(npx * 3);
        if (hipMemcpy(rgba8.data(), dev8, npx * 4, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "hipMemcpy failed\n"); return 1; }
        for (size_t i = 0; i < npx; ++i) { rgb[i * 3] = rgba8[i * 4]; rgb[i * 3 + 1] = rgba8[i * 4 + 1]; rgb[i * 3 + 2] = rgba8[i * 4 + 2]; }
        fwrite(rgb.data(), 1, rgb.size(), fp);
        fclose(fp);
        (void)hipFree(dev8);
    }
    (void)hipFree(dev);
    sbx_destroy(ctx);
    return 0;
}
