// host/mainimage_threads.cpp — the reference-style per-pixel loop run by MANY host threads over the drop-in header
// (the reference's globals are thread_local for exactly this use, /root/reference/src/def.h:7-8): T threads take disjoint rows of
// one frame and call mainImage(fragColor, fragCoord) per pixel.  Checks that the whole job was ONE kernel launch (sbx_get_stats)
// and that every pixel equals sbx_render_rows' frame bit for bit.  Exit 0 = both hold; prints one line of facts.
//     mainimage_threads [W H threads time]
#include "../include/sbx_mainimage.hpp"

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

struct vec2 { float x, y; float operator[](int i) const { return i ? y : x; } };
struct vec4 { float x, y, z, w; float& operator[](int i) { return (&x)[i]; } };

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 1920, H = argc > 2 ? atoi(argv[2]) : 1080;
    const int T = argc > 3 ? atoi(argv[3]) : 16;
    const float time = argc > 4 ? (float)atof(argv[4]) : 0.37f;
    std::vector<float> got((size_t)W * H * 4, -1.f);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int k = 0; k < T; ++k)
        th.emplace_back([&, k] {
            // the uniforms are thread_local, as in the reference: every thread sets its own
            iResolution[0] = (float)W; iResolution[1] = (float)H;
            iGlobalTime = time;
            for (int y = k; y < H; y += T)
                for (int x = 0; x < W; ++x) {
                    vec4 c;
                    mainImage(c, vec2{x + .5f, y + .5f});
                    std::memcpy(&got[((size_t)y * W + x) * 4], &c, 16);
                }
        });
    for (auto& t : th) t.join();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // the same loop again: the frame is cached now, every call is a hit (the first pass's time holds sbx_create and the launch)
    const auto h0 = std::chrono::steady_clock::now();
    th.clear();
    for (int k = 0; k < T; ++k)
        th.emplace_back([&, k] {
            iResolution[0] = (float)W; iResolution[1] = (float)H;
            iGlobalTime = time;
            for (int y = k; y < H; y += T)
                for (int x = 0; x < W; ++x) {
                    vec4 c;
                    mainImage(c, vec2{x + .5f, y + .5f});
                    std::memcpy(&got[((size_t)y * W + x) * 4], &c, 16);
                }
        });
    for (auto& t : th) t.join();
    const double hit_sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count();
    sbx_ctx* ctx = sbx_host::context().get();
    sbx_stats st{};
    sbx_get_stats(ctx, &st);
    // the same frame through the frame-granular entry
    sbx_uniforms u{};
    u.u_res[0] = (float)W; u.u_res[1] = (float)H; u.u_time = time;
    float* dev = nullptr;
    std::vector<float> ref((size_t)W * H * 4);
    if (hipMalloc((void**)&dev, ref.size() * 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 2; }
    if (sbx_render_rows(ctx, SBX_SELECTED_APP, &u, nullptr, 0, H, dev, nullptr) != SBX_OK) { fprintf(stderr, "%s\n", sbx_last_error(ctx)); return 2; }
    if (hipMemcpy(ref.data(), dev, ref.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "hipMemcpy failed\n"); return 2; }
    (void)hipFree(dev);
    size_t differing = 0;
    for (size_t i = 0; i < ref.size(); i += 4) differing += std::memcmp(&got[i], &ref[i], 16) != 0;
    printf("%dx%d threads=%d pixels=%zu launches_by_main_image=%llu frames=%llu hits=%llu points=%llu differing=%zu seconds=%.3f Mpixels_per_s=%.1f | second pass (hits only) seconds=%.4f Mpixels_per_s=%.1f ns_per_call_per_thread=%.1f\n",
           W, H, T, (size_t)W * H, (unsigned long long)st.render_launches, (unsigned long long)st.main_image_frames,
           (unsigned long long)st.main_image_hits, (unsigned long long)st.main_image_points, differing, sec, W * (double)H / sec / 1e6,
           hit_sec, W * (double)H / hit_sec / 1e6, hit_sec * 1e9 * T / (W * (double)H));
    return (st.render_launches == 1 && st.main_image_frames == 1 && differing == 0 && st.main_image_hits + 1 == 2ull * W * H) ? 0 : 1;
}
