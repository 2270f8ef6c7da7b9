// include/sbx_mainimage.hpp — C++ drop-in for the reference's C++/VML build of an app.
//
// In the reference, a C++ host includes one app header chosen by a global APP_* define
// (/root/reference/README.md:11-22, src/Makefile:9), gets the globals iResolution / iGlobalTime /
// iMouse (src/uniform_buffer.h:32-36) and the entry point
//     void mainImage(out vec4 fragColor, in vec2 fragCoord)            (src/main.h:6-9)
// and loops over the pixels itself.  This header keeps exactly that surface — same define, same
// globals, same entry point — but evaluates the frame on an MI355X through the C ABI of libsbx:
// the first mainImage() call after the uniforms change renders the WHOLE frame with one HIP kernel
// launch and copies it to the host; every call then returns its pixel from that frame.  So an
// unmodified per-pixel host loop keeps working and costs one GPU frame per frame.
//
// Threads.  The reference's globals are thread_local (src/def.h:7-8) because its harness may call mainImage()
// from many threads at once, each with private scene state.  Here the UNIFORMS are thread_local too, but the
// library context is ONE per process: all host threads that ask for pixels of the same frame share one launch
// and one host copy (sbx_main_image serves hits without a lock, include/sbx.h), instead of one context, one
// launch and one 133 MB frame per thread.  Threads with different uniforms at the same time are served from
// the context's two cached frames; more than two distinct frames in flight re-render on every switch.
//
//   g++ -std=c++17 -DAPP_CLOUDS host.cpp -I include -L shaderbox_amd/lib -lsbx -L /opt/rocm/lib -lamdhip64
//                                                                       (see host/Makefile; no HIP headers needed)
//
// Vector types: anything indexable with operator[] (VML's vector<float,...> is); see host/sbx_render.cpp
// for a complete host.  Errors (no GPU, HIP failure) throw std::runtime_error — there is no CPU path.
#pragma once
#include <stdexcept>
#include <string>

#include "sbx.h"

#if defined(APP_PLANET_ATMOSPHERE)   /* config 5's composite (include/sbx.h): not an APP_* define of the reference */
#define SBX_SELECTED_APP SBX_APP_PLANET_ATMOSPHERE
#elif defined(APP_PLANET)
#define SBX_SELECTED_APP SBX_APP_PLANET
#elif defined(APP_CLOUDS)
#define SBX_SELECTED_APP SBX_APP_CLOUDS
#elif defined(APP_EGG)
#define SBX_SELECTED_APP SBX_APP_EGG
#elif defined(APP_RAYTRACER)
#define SBX_SELECTED_APP SBX_APP_RAYTRACER
#elif defined(APP_ATMOSPHERE)
#define SBX_SELECTED_APP SBX_APP_ATMOSPHERE
#elif defined(APP_SDF_AO)
#define SBX_SELECTED_APP SBX_APP_SDF_AO
#elif defined(APP_VINYL)
#define SBX_SELECTED_APP SBX_APP_VINYL
#elif defined(APP_CLOUDS_BEST)   /* src/app_clouds_best.h, the stand-alone shader (no APP_* define in the reference) */
#define SBX_SELECTED_APP SBX_APP_CLOUDS_BEST
#else
#error "define one of APP_PLANET APP_CLOUDS APP_VINYL APP_EGG APP_RAYTRACER APP_ATMOSPHERE APP_SDF_AO"
#endif

namespace sbx_host {

struct float2_t { float v[2]; float& operator[](int i) { return v[i]; } float operator[](int i) const { return v[i]; } };
struct float4_t { float v[4]; float& operator[](int i) { return v[i]; } float operator[](int i) const { return v[i]; } };

// the reference's C++ globals (src/uniform_buffer.h:32-36)
inline thread_local float2_t iResolution{{0, 0}};
inline thread_local float iGlobalTime = 0.f;
inline thread_local float4_t iMouse{{0, 0, 0, 0}};

// ONE library context per process, created by whichever thread asks first (C++11 guarantees the one-time, thread-safe
// construction of a function-local static); the library's ABI is checked against this header's.
struct Context {
    sbx_ctx* ctx = nullptr;
    Context() {
        if (sbx_abi_version() != SBX_ABI_VERSION)
            throw std::runtime_error("libsbx.so has ABI " + std::to_string(sbx_abi_version()) + ", this header was written for ABI " +
                                     std::to_string(SBX_ABI_VERSION));
        const int rc = sbx_create(0, &ctx);
        if (rc != SBX_OK) throw std::runtime_error("sbx_create failed (" + std::to_string(rc) + "): a gfx950 GPU is required");
    }
    ~Context() { if (ctx) sbx_destroy(ctx); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    sbx_ctx* get() { return ctx; }
};
inline Context& context() { static Context c; return c; }

}  // namespace sbx_host

using sbx_host::iGlobalTime;
using sbx_host::iMouse;
using sbx_host::iResolution;

// void mainImage(out vec4 fragColor, in vec2 fragCoord)   — src/main.h:6-9
template <class Vec4, class Vec2>
inline void mainImage(Vec4& fragColor, const Vec2& fragCoord) {
    sbx_uniforms u{};
    u.u_res[0] = iResolution[0]; u.u_res[1] = iResolution[1];
    u.u_mouse[0] = iMouse[0]; u.u_mouse[1] = iMouse[1];
    u.u_time = iGlobalTime;
    const float fc[2] = {fragCoord[0], fragCoord[1]};
    float c[4];
    sbx_ctx* ctx = sbx_host::context().get();
    const int rc = sbx_main_image(ctx, SBX_SELECTED_APP, &u, nullptr, fc, c);    // renders the frame on first use
    if (rc != SBX_OK) throw std::runtime_error(std::string("sbx_main_image: ") + sbx_last_error(ctx));
    fragColor[0] = c[0]; fragColor[1] = c[1]; fragColor[2] = c[2]; fragColor[3] = c[3];
}
