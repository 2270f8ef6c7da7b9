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
//   g++ -std=c++17 -DAPP_CLOUDS host.cpp -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__
//       -L shaderbox_amd/lib -lsbx -L /opt/rocm/lib -lamdhip64          (see host/Makefile)
//
// Vector types: anything indexable with operator[] (VML's vector<float,...> is); see host/sbx_render.cpp
// for a complete host.  Errors (no GPU, HIP failure) throw std::runtime_error — there is no CPU path.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "sbx.h"

#if defined(APP_PLANET)
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

class FrameCache {
public:
    ~FrameCache() {
        if (dev_) (void)hipFree(dev_);
        if (ctx_) sbx_destroy(ctx_);
    }
    // returns the host copy of the frame for the current uniforms, rendering it if they changed
    const std::vector<float>& frame(int app, const void* aux = nullptr) {
        const float w = iResolution[0], h = iResolution[1];
        if (!ctx_) {
            int rc = sbx_create(0, &ctx_);
            if (rc != SBX_OK) throw std::runtime_error("sbx_create failed (" + std::to_string(rc) + "): a gfx950 GPU is required");
        }
        if (valid_ && w == w_ && h == h_ && iGlobalTime == t_ && iMouse[0] == mx_ && iMouse[1] == my_ && app == app_)
            return host_;
        const size_t n = (size_t)w * (size_t)h * 4;
        if (n != host_.size()) {
            if (dev_) (void)hipFree(dev_);
            dev_ = nullptr;
            if (hipMalloc((void**)&dev_, n * sizeof(float)) != hipSuccess) throw std::runtime_error("hipMalloc failed");
            host_.assign(n, 0.f);
        }
        sbx_uniforms u{};
        u.u_res[0] = w; u.u_res[1] = h; u.u_mouse[0] = iMouse[0]; u.u_mouse[1] = iMouse[1]; u.u_time = iGlobalTime;
        int rc = sbx_render_rows(ctx_, app, &u, aux, 0, (int)h, dev_, nullptr);
        if (rc != SBX_OK) throw std::runtime_error(std::string("sbx_render_rows: ") + sbx_last_error(ctx_));
        if (hipMemcpy(host_.data(), dev_, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
            throw std::runtime_error("hipMemcpy failed");
        w_ = w; h_ = h; t_ = iGlobalTime; mx_ = iMouse[0]; my_ = iMouse[1]; app_ = app; valid_ = true;
        return host_;
    }
private:
    sbx_ctx* ctx_ = nullptr;
    float* dev_ = nullptr;
    std::vector<float> host_;
    float w_ = 0, h_ = 0, t_ = 0, mx_ = 0, my_ = 0;
    int app_ = -1;
    bool valid_ = false;
};
inline FrameCache& frame_cache() { static thread_local FrameCache c; return c; }

}  // namespace sbx_host

using sbx_host::iGlobalTime;
using sbx_host::iMouse;
using sbx_host::iResolution;

// void mainImage(out vec4 fragColor, in vec2 fragCoord)   — src/main.h:6-9
template <class Vec4, class Vec2>
inline void mainImage(Vec4& fragColor, const Vec2& fragCoord) {
    const std::vector<float>& f = sbx_host::frame_cache().frame(SBX_SELECTED_APP);
    const int W = (int)iResolution[0], H = (int)iResolution[1];
    int x = (int)std::floor(fragCoord[0]), y = (int)std::floor(fragCoord[1]);   // pixel centre (x+.5, y+.5) -> (x, y)
    x = x < 0 ? 0 : (x >= W ? W - 1 : x);
    y = y < 0 ? 0 : (y >= H ? H - 1 : y);
    const float* p = &f[((size_t)y * W + x) * 4];
    fragColor[0] = p[0]; fragColor[1] = p[1]; fragColor[2] = p[2]; fragColor[3] = p[3];
}
