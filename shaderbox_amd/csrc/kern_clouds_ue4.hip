// shaderbox_amd/csrc/kern_clouds_ue4.hip — the UE4 cloud variant (SBX_APP_CLOUDS_UE4).
//
// Follows /root/reference/ue4/volumetric_clouds/Shaders/app_clouds.usf: ue4_render_clouds :234-265, render_clouds :181-231
// (plane-projection branch, STEPS 25 :16), density_func :164-179, fbm :123-135 (four octaves of noise_iq with its own
// weights, FBM_FREQ 2.76434 :9), render_sky_color :151-162 (no abs), TWEAK defaults :4-17.  The shader is a library for an
// Unreal material whose graph (a binary asset) supplies cam_dir and the parameters per pixel; the HOST MAPPING here is the
// build's own and is stated in include/sbx.h: cam_dir = the primary-ray direction of APP_CLOUDS' mainImage camera, the
// parameters = the TWEAK defaults or an sbx_aux_clouds_ue4 block, output through main.h's sRGB epilogue.  No known
// answers exist for it: parity is against the oracle's restatement only (bit-identical).
//
// Every lane marches all 25 steps (the shader has no early exit), so the wave is uniform by construction and the four
// noise_iq of a density sample go through the per-wave LDS hash cache as one cooperative batch (sbx_hashcache.h).
#ifndef SBX_HC_SLOTS
#define SBX_HC_SLOTS 32
#endif
#ifndef SBX_WG_WAVES
#define SBX_WG_WAVES 1        // single-wave workgroups (one hash cache per wave, waves never talk): 1.78 -> 1.76 ms at 4K
#endif
#include "sbx_device.h"
#include "sbx_noise.h"
#include "sbx_hashcache.h"
#include "sbx_exp4k_table.h"
#include <cmath>
#ifndef UE4_FAST_MATH
#define UE4_FAST_MATH 1
#endif

namespace sbx {

// waves per SIMD (tools/ab_time.py, 3840x2160, same bits): 4 (100 VGPRs) 2.00 ms; 5 (96 VGPRs, no scratch, HBM traffic = the frame)
// 1.80; 6 (80 VGPRs, 60 B of scratch per lane, 0.5 GB of traffic for a 0.13 GB frame) 1.76; 6 with 16-slot tables 1.81; 7 1.89; 8 2.78
#ifndef UE4_MIN_WAVES
#define UE4_MIN_WAVES 5
#endif
// FAST (decided on the host, launch_clouds_ue4): the coverage smoothstep's division through div3_ and the exp through exp_reg4k_ of
// sbx_math.h — equal to smoothstep_rd / exp_ on their domains: dens is a sum of noise values in [0, 1] with gains < 1 (or NaN), so
// dens - cov is zero or >= 2^-45 in magnitude for 2^-20 <= |cov| <= 2^20, and the exp argument is -absorbtion * clamp(dens, 0, 1) *
// march_step with |absorbtion * march_step| <= 80.
template <bool FAST>
__global__ void __launch_bounds__(WG_THREADS, UE4_MIN_WAVES) k_clouds_ue4(FrameCloudsUe4 F, RowMap M, float* __restrict__ out) {
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();      // (the dispatch order's cost table, RowMap.cost)
    __shared__ WaveCache cache[WG_THREADS / 64];
    const int lane = threadIdx.x & 63;
    WaveCache& S = cache[threadIdx.x >> 6];
    hc_init(S, lane);
    const Pixel px = pixel_of_thread(M);
    const v2 pc = point_cam(F.cam, px.fx, px.fy);
    const v3 dir = primary_dir(F.cam, pc);                        // cam_dir

    // render_clouds :181-231
    const v3 dir_step = (dir / dir.y) * F.march_step;
    v3 pos = V3(0, 0, 0) + dir * 100.f;
    float T = 1.f, C = 0.f, alpha = 0.f;
    const int tab[4] = {0, 1, 2, 3};
    float cd = F.cov_d, cr = F.cov_r;
    if (FAST) asm volatile("" : "+v"(cd), "+v"(cr));          // VGPR operands: full rate
    for (int i = 0; i < UE4_STEPS; ++i) {
        // density_func :164-179
        v3 p[4];
        p[0] = pos * .0212242f + F.wind_dir;
        p[1] = p[0] * 2.76434f; p[2] = p[1] * 2.76434f; p[3] = p[2] * 2.76434f;
        float nz[4];
        coop_noise_n<4>(S, p, tab, px.valid, lane, nz);
        float dens = 0.51749673f * nz[0];                          // fbm :123-135
        dens += 0.25584929f * nz[1];
        dens += 0.12527603f * nz[2];
        dens += 0.06255931f * nz[3];
        if (FAST) dens = x_smoothstep_d3_med3(F.cov, cd, cr, dens);      // dens * smoothstep(..): one v_med3 for the clamp, a NaN dens stays NaN
        else dens *= smoothstep_rd(F.cov, F.cov_rd, dens);
        dens = clamp_(dens, 0.f, 1.f);
        const float T_i = FAST ? exp_reg4k_(-F.absorbtion * dens * F.march_step, kExp2Tab4096) : exp_(-F.absorbtion * dens * F.march_step);
        T *= T_i;
        C += T * F.eh[i] * dens * F.march_step;                    // exp(h) / 1.75, h = i / steps: a frame constant
        alpha += (1.f - T_i) * (1.f - alpha);
        pos = pos + dir_step;
    }
    if (!px.valid) return;
    // render_sky_color :151-162
    const float sun_amount = fmax_(dot(dir, F.sun_dir), 0.f);
    v3 sky = mix3(V3(.0f, .1f, .4f), V3(.3f, .6f, .8f), 1.0f - dir.y);
    sky = sky + V3(1.f, .7f, .55f) * fmin_(pow_(sun_amount, 1500.0f) * 5.0f, 1.0f);
    sky = sky + V3(1.f, .7f, .55f) * fmin_(pow_(sun_amount, 10.0f) * .6f, 1.0f);
    tile_cost_store(M, tl_t0);
    store_rgba(M, out, px.idx, to_srgb(mix3(sky, V3s(C), alpha)));
}

dim3 clouds_ue4_grid(const RowMap& M) { return grid_for(M); }

void launch_clouds_ue4(const FrameCloudsUe4& F, const RowMap& M, float* out, hipStream_t s) {
    const double c = std::fabs((double)F.cov), d = std::fabs((double)F.cov_d);
    const bool fast = UE4_FAST_MATH && std::isfinite(F.cov) && std::isfinite(F.cov_d) && std::isfinite(F.cov_r) && c >= 0x1p-20 && c <= 0x1p20 &&
                      d >= 0x1p-60 && d <= 0x1p60 && std::fabs((double)F.absorbtion) * std::fabs((double)F.march_step) <= 80.0;   // NaN: false
    if (fast) hipLaunchKernelGGL(k_clouds_ue4<true>, grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
    else hipLaunchKernelGGL(k_clouds_ue4<false>, grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
}

hipError_t bind_fault_clouds_ue4(unsigned* word) { return hc_bind_fault_word(word); }

}  // namespace sbx
