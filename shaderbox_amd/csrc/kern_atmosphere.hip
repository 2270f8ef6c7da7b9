// shaderbox_amd/csrc/kern_atmosphere.hip — APP_ATMOSPHERE: Rayleigh/Mie single scattering.
//
// Follows /root/reference/src/app_atmosphere.h (FROM_SPACE defined, :162): render :183-228 (sky
// dome branch :190-209), get_incident_light :78-160 (16 view samples), get_sun_light :50-76
// (8 light samples), isect_sphere :15-26, phase functions src/volumetric.h:13-33 (hg_g = .76).
// sun_dir (a _mutable global rotated by setup_scene, :177-181) is a frame constant built on the
// host from (0,1,0), i.e. it restarts from its initialiser for every pixel (GLSL semantics).
// Magnitudes are ~6.4e6 in binary32, so the evaluation order below is part of the result.
#include "sbx_device.h"

namespace sbx {

constexpr float ATM_EARTH_R = 6360e3f, ATM_ATMOS_R = 6420e3f, ATM_HR = 7994.0f, ATM_HM = 1200.0f;   // :34-38
constexpr double ATM_HR_RD = 1.0 / (double)ATM_HR, ATM_HM_RD = 1.0 / (double)ATM_HM;   // exact division by constants (sbx_math.h div_by)

// isect_sphere with the atmosphere sphere (origin 0)                        :15-26
__device__ __forceinline__ bool isect_atmosphere(v3 ro, v3 rd, float& t1) {
    const v3 rc = V3(0, 0, 0) - ro;
    const float radius2 = ATM_ATMOS_R * ATM_ATMOS_R;
    const float tca = dot(rc, rd);
    const float d2 = dot(rc, rc) - tca * tca;
    const float thc = sqrt_(radius2 - d2);
    t1 = tca + thc;
    return d2 < radius2;
}

__device__ __forceinline__ bool sun_light(v3 ro, v3 rd, float& odR, float& odM) {   // :50-76
    float t1;
    isect_atmosphere(ro, rd, t1);
    float march_pos = 0.f;
    const float march_step = t1 / 8.f;
    for (int i = 0; i < 8; ++i) {
        const v3 s = ro + rd * (march_pos + 0.5f * march_step);
        const float height = length(s) - ATM_EARTH_R;
        if (height < 0.f) return false;
        odR += exp_(div_by(-height, ATM_HR_RD)) * march_step;
        odM += exp_(div_by(-height, ATM_HM_RD)) * march_step;
        march_pos += march_step;
    }
    return true;
}

__global__ void __launch_bounds__(WG_THREADS) k_atmosphere(FrameAtmosphere F, RowMap M, float* __restrict__ out) {
    const Pixel px = pixel_of_thread(M);
    if (!px.valid) return;
    const v2 pc = point_cam(F.cam, (float)px.x + .5f, (float)px.y + .5f);

    // sky dome mapping :195-207
    const float z2 = pc.x * pc.x + pc.y * pc.y;
    const float phi = atan2_(pc.y, pc.x);
    const float theta = acos_(1.0f - z2);
    const v3 rd = V3(sin_(theta) * cos_(phi), cos_(theta), sin_(theta) * sin_(phi));
    const v3 ro = V3(0, ATM_EARTH_R + 1.f, 0);

    v3 col = V3(0.f, 0.f, 0.f);
    float t1;
    if (isect_atmosphere(ro, rd, t1)) {                             // get_incident_light :78-160
        const v3 betaR = V3(5.5e-6f, 13.0e-6f, 22.4e-6f), betaM = V3(21e-6f, 21e-6f, 21e-6f);   // :29-30
        const float march_step = t1 / 16.f;
        const float mu = dot(rd, F.sun_dir);
        const float phaseR = 3.f * (1.f + mu * mu) / (16.f * 3.14159265359f);            // volumetric.h:13-19
        const float g = .76f;
        const float phaseM = (1.f - g * g) / ((4.f + 3.14159265359f) * pow_(1.f + g * g - 2.f * g * mu, 1.5f));
        float odR = 0.f, odM = 0.f, march_pos = 0.f;
        v3 sumR = V3(0, 0, 0), sumM = V3(0, 0, 0);
        for (int i = 0; i < 16; ++i) {
            const v3 s = ro + rd * (march_pos + 0.5f * march_step);
            const float height = length(s) - ATM_EARTH_R;
            const float hr = exp_(div_by(-height, ATM_HR_RD)) * march_step;
            const float hm = exp_(div_by(-height, ATM_HM_RD)) * march_step;
            odR += hr;
            odM += hm;
            float lR = 0.f, lM = 0.f;
            if (sun_light(s, F.sun_dir, lR, lM)) {
                const v3 tau = betaR * (odR + lR) + betaM * 1.1f * (odM + lM);
                const v3 att = V3(exp_(-tau.x), exp_(-tau.y), exp_(-tau.z));
                sumR = sumR + hr * att;
                sumM = sumM + hm * att;
            }
            march_pos += march_step;
        }
        col = 20.0f * (sumR * phaseR * betaR + sumM * phaseM * betaM);   // sun_power :41
    }
    store_rgba(out, px.idx, to_srgb(col));
}

void launch_atmosphere(const FrameAtmosphere& F, const RowMap& M, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_atmosphere, grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
}

}  // namespace sbx
