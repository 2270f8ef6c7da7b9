// shaderbox_amd/csrc/kern_clouds.hip — APP_CLOUDS: volumetric fBm cloud integrator.
//
// Follows /root/reference/src/app_clouds.h (SKY_SPHERE / USE_NOISE_TEX undefined, :8-9):
// render :204-218, render_sky_color :36-46, render_clouds :153-202, integrate_volume :125-148,
// illuminate_volume :91-123, density_func :62-86 over fbm = 4 octaves of noise_iq (:59),
// henyey_greenstein_phase_func src/volumetric.h:27-33 with hg_g = .2 (:5).
#include "sbx_device.h"
#include "sbx_noise.h"

namespace sbx {

__device__ __forceinline__ float clouds_density(const FrameClouds& F, v3 pos_in) {
    v3 pos = pos_in * .001f;                                   // cld_noise_factor, :20,66
    float shape = fbm<4>(pos * 2.03f, 2.64f, .5f, .5f, [](v3 p) { return noise_iq(p); });   // :72
    return shape * smoothstep_(F.cov, F.cov_hi, shape);        // :83-84
}

__device__ __forceinline__ float hg_phase(float mu, float g) {  // volumetric.h:27-33, note (4 + PI)
    return (1.f - g * g) / ((4.f + 3.14159265359f) * pow_(1.f + g * g - 2.f * g * mu, 1.5f));
}

// illuminate_volume :91-123.  `phase` = henyey_greenstein(clamp(dot(L,V),0,1)) depends only on
// the pixel's ray, so the caller evaluates it once per pixel (same inputs, same bits).
__device__ __forceinline__ float clouds_illuminate(const FrameClouds& F, v3 origin, float phase) {
    const v3 step = F.sun_dir * F.dt;
    v3 pos = origin + step;
    float transmittance = 1.f;
    for (int i = 0; i < F.lsteps; ++i) {
        float density = clouds_density(F, pos);
        transmittance *= exp_(-density * F.sigma * F.dt);
        pos = pos + step;
    }
    return transmittance * F.sun_power * phase;
}

__global__ void __launch_bounds__(WG_THREADS) k_clouds(FrameClouds F, RowMap M, float* __restrict__ out) {
    const Pixel px = pixel_of_thread(M);
    if (!px.valid) return;
    const v2 pc = point_cam(F.cam, (float)px.x + .5f, (float)px.y + .5f);
    const v3 dir = primary_dir(F.cam, pc);

    // render_sky_color :36-46
    float sun_amount = fmax_(dot(dir, F.sun_dir), 0.f);
    v3 sky = mix3(V3(.0f, .1f, .4f), V3(.3f, .6f, .8f), 1.0f - dir.y);
    sky = sky + F.sun_color * fmin_(pow_(sun_amount, 1500.0f) * 5.0f, 1.0f);
    sky = sky + F.sun_color * fmin_(pow_(sun_amount, 10.0f) * .6f, 1.0f);
    sky = abs3(sky);

    const float cutoff = dot(dir, V3(0, 1, 0));
    v3 col = sky;
    if (!(cutoff < 0.05f)) {                                   // :212
        // render_clouds :153-202
        const v3 projection = dir / dir.y;
        v3 origin = F.cam.eye + projection * 150.f;
        origin = origin + F.wind_off;
        const float phase = hg_phase(clamp_(dot(F.sun_dir, dir), 0.f, 1.f), .2f);
        float transmittance = 1.f, radiance = 0.f, alpha = 0.f, t = 0.f;
        for (int i = 0; i < F.steps; ++i) {
            const v3 pos = origin + t * projection;
            t += F.dt;
            const float density = clouds_density(F, pos);
            if (!(density < .005f)) {                          // integrate_volume :132
                const float T_i = exp_(-density * F.sigma * F.dt);
                transmittance *= T_i;
                radiance += (density * F.sigma) * clouds_illuminate(F, pos, phase) * transmittance * F.dt;
                alpha += (1.f - T_i) * (1.f - alpha);
            }
            if (alpha > .999f) break;
        }
        const float a = alpha * smoothstep_(.0f, .2f, cutoff);
        col = abs3(mix3(sky, V3s(radiance), a));               // :215-217 (radiance is r=g=b)
    }
    store_rgba(out, px.idx, to_srgb(col));
}

void launch_clouds(const FrameClouds& F, const RowMap& M, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_clouds, grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
}

}  // namespace sbx
