// host/mainimage_stress.cpp — sbx_main_image under contention: T host threads ask ONE context for random pixel centres of THREE
// different frames (three u_time values) while the context caches two, so frames are evicted and re-rendered under the readers all
// the time, and one resolution change retires the pinned buffers in mid-run.  Every colour returned must be the right frame's pixel
// (reference frames rendered up front with sbx_render_rows).  Exit 0 = no wrong pixel, no error.   mainimage_stress [threads seconds]
#include "../include/sbx.h"

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

static std::vector<float> frame_of(sbx_ctx* ctx, int app, int W, int H, float t) {
    sbx_uniforms u{};
    u.u_res[0] = (float)W; u.u_res[1] = (float)H; u.u_time = t;
    float* dev = nullptr;
    std::vector<float> f((size_t)W * H * 4);
    if (hipMalloc((void**)&dev, f.size() * 4) != hipSuccess) exit(2);
    if (sbx_render_rows(ctx, app, &u, nullptr, 0, H, dev, nullptr) != SBX_OK) { fprintf(stderr, "%s\n", sbx_last_error(ctx)); exit(2); }
    if (hipMemcpy(f.data(), dev, f.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) exit(2);
    (void)hipFree(dev);
    return f;
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 8;
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    sbx_ctx *ctx = nullptr, *refctx = nullptr;
    if (sbx_create(0, &ctx) != SBX_OK || sbx_create(0, &refctx) != SBX_OK) { fprintf(stderr, "sbx_create failed\n"); return 2; }
    const int app = SBX_APP_EGG;
    const float times[3] = {0.37f, 1.25f, 2.5f};
    const int sizes[2][2] = {{256, 144}, {320, 200}};       // the second size makes every entry outgrow its pinned buffer once
    std::vector<float> ref[2][3];
    for (int s = 0; s < 2; ++s) for (int k = 0; k < 3; ++k) ref[s][k] = frame_of(refctx, app, sizes[s][0], sizes[s][1], times[k]);
    std::atomic<long long> calls{0}, wrong{0}, errors{0};
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int id = 0; id < T; ++id)
        th.emplace_back([&, id] {
            std::mt19937 rng(1234u + (unsigned)id);
            long long n = 0;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                const int s = el > seconds / 2 ? 1 : 0;                      // the resolution changes half way through
                const int W = sizes[s][0], H = sizes[s][1];
                // mostly the thread's "own" frame, sometimes another: hits and evictions mixed
                const int k = (rng() % 8 == 0) ? (int)(rng() % 3) : (id % 3);
                sbx_uniforms u{};
                u.u_res[0] = (float)W; u.u_res[1] = (float)H; u.u_time = times[k];
                for (int j = 0; j < 64; ++j) {
                    const int x = (int)(rng() % (unsigned)W), y = (int)(rng() % (unsigned)H);
                    const float fc[2] = {x + .5f, y + .5f};
                    float c[4];
                    if (sbx_main_image(ctx, app, &u, nullptr, fc, c) != SBX_OK) { errors++; continue; }
                    if (std::memcmp(c, &ref[s][k][((size_t)y * W + x) * 4], 16) != 0) wrong++;
                    ++n;
                }
            }
            calls += n;
        });
    for (auto& t : th) t.join();
    sbx_stats st{};
    sbx_get_stats(ctx, &st);
    printf("threads=%d seconds=%.1f calls=%lld hits=%llu frames_rendered=%llu wrong=%lld errors=%lld\n", T, seconds, (long long)calls,
           (unsigned long long)st.main_image_hits, (unsigned long long)st.main_image_frames, (long long)wrong, (long long)errors);
    sbx_destroy(ctx);
    sbx_destroy(refctx);
    return (wrong == 0 && errors == 0 && st.main_image_frames > 20) ? 0 : 1;
}
