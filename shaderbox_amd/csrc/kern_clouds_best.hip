// shaderbox_amd/csrc/kern_clouds_best.hip — the reference's stand-alone cloud shader
// (/root/reference/src/app_clouds_best.h): 50-step march through a 5-octave |simplex|
// fBm with analytic "height" lighting.  No lattice hashing and no transcendental inside the noise — the
// Ashima simplex noise is ~250 plain fp32 operations — so this is a straight per-lane kernel: one thread
// per pixel, early exit on alpha, the per-step frame constants (FrameCloudsBest::row) read as scalars.
// Bit-identical to oracle/ref_apps.h AppCloudsBest (same operations in the same order, no contraction).
#include <hip/hip_runtime.h>
#include <cmath>
#include "sbx_device.h"
#include "sbx_exp4k_table.h"

#ifndef CB_FAST_MATH
#define CB_FAST_MATH 1
#endif
namespace sbx {

// XI ("exact integers", decided on the host per launch: every lattice coordinate of the frame is below 2^22 in magnitude): the
// simplex noise's index arithmetic works on integer-valued floats, and wherever a product m * c of such values is EXACT
// (an integer below 2^24) the reference's RN(x -+ RN(m * c)) equals RN(x -+ m * c), which is one fma.  That holds for
//   mod289(x) = x - floor(x / 289) * 289          (floor(x / 289) * 289 <= |x| + 289 < 2^24)
//   permute(x) = mod289((x * 34 + 1) * x)         (x < 600: x * 34 < 2^15)
//   j = p - 49 * floor(..), y_ = floor(j - 7 * x_)  (p < 600)
// and, for ANY finite value, for f * 2 + 1 (a power-of-two scale is exact) and b + s * sh with sh in {-0, -1}.
// 51 of the ~250 instructions of a snoise; same bits (the oracle evaluates the plain forms).  Without XI (huge or non-finite
// coordinates) the plain forms run.
template <bool XI> __device__ __forceinline__ float sn_mod289(float x) {                                  // :460-466
    const float q = floor_(x * (1.0f / 289.0f));
    return XI ? __builtin_fmaf(-289.0f, q, x) : x - q * 289.0f;
}
template <bool XI> __device__ __forceinline__ float sn_permute(float x) {                                 // :469-471
    return sn_mod289<XI>((XI ? __builtin_fmaf(x, 34.0f, 1.0f) : ((x * 34.0f) + 1.0f)) * x);
}

// snoise :478-551; vec4 quantities are 4 scalars, operations component-wise in source order
template <bool XI>
__device__ __forceinline__ float snoise(float vx, float vy, float vz) {
    const float Cx = 1.0f / 6.0f, Cy = 1.0f / 3.0f;
    const float s = (vx * Cy + vy * Cy) + vz * Cy;                         // dot(v, C.yyy)
    float ix = floor_(vx + s), iy = floor_(vy + s), iz = floor_(vz + s);
    const float t = (ix * Cx + iy * Cx) + iz * Cx;                         // dot(i, C.xxx)
    const float x0x = (vx - ix) + t, x0y = (vy - iy) + t, x0z = (vz - iz) + t;
    const float gx = step_(x0y, x0x), gy = step_(x0z, x0y), gz = step_(x0x, x0z);   // step(x0.yzx, x0.xyz)
    const float lx = 1.0f - gx, ly = 1.0f - gy, lz = 1.0f - gz;
    const float i1x = fmin_(gx, lz), i1y = fmin_(gy, lx), i1z = fmin_(gz, ly);      // min(g.xyz, l.zxy)
    const float i2x = fmax_(gx, lz), i2y = fmax_(gy, lx), i2z = fmax_(gz, ly);
    const float x1x = (x0x - i1x) + Cx, x1y = (x0y - i1y) + Cx, x1z = (x0z - i1z) + Cx;
    const float x2x = (x0x - i2x) + Cy, x2y = (x0y - i2y) + Cy, x2z = (x0z - i2z) + Cy;
    const float x3x = x0x - 0.5f, x3y = x0y - 0.5f, x3z = x0z - 0.5f;
    ix = sn_mod289<XI>(ix); iy = sn_mod289<XI>(iy); iz = sn_mod289<XI>(iz);
    const float oz[4] = {0.0f, i1z, i2z, 1.0f}, oy[4] = {0.0f, i1y, i2y, 1.0f}, ox[4] = {0.0f, i1x, i2x, 1.0f};
    const float n_ = 0.142857142857f;
    const float nsx = n_ * 2.0f - 0.0f, nsy = n_ * 0.5f - 1.0f, nsz = n_ * 1.0f - 0.0f;
    float ax[4], ay[4], h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float p = sn_permute<XI>((sn_permute<XI>((sn_permute<XI>(iz + oz[k]) + iy) + oy[k]) + ix) + ox[k]);
        const float m49 = floor_(p * nsz * nsz);
        const float j = XI ? __builtin_fmaf(-49.0f, m49, p) : p - 49.0f * m49;
        const float x_ = floor_(j * nsz);
        const float y_ = floor_(XI ? __builtin_fmaf(-7.0f, x_, j) : j - 7.0f * x_);
        ax[k] = x_ * nsx + nsy;
        ay[k] = y_ * nsx + nsy;
        h[k] = 1.0f - abs_(ax[k]) - abs_(ay[k]);
    }
    const float b0[4] = {ax[0], ax[1], ay[0], ay[1]}, b1[4] = {ax[2], ax[3], ay[2], ay[3]};
    float s0[4], s1[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s0[k] = __builtin_fmaf(floor_(b0[k]), 2.0f, 1.0f);       // f * 2 is exact: one rounding either way
        s1[k] = __builtin_fmaf(floor_(b1[k]), 2.0f, 1.0f);
        sh[k] = -step_(h[k], 0.0f);
    }
    // a0 = b0.xzyw + s0.xzyw * sh.xxyy ; a1 = b1.xzyw + s1.xzyw * sh.zzww
    // (sh is -0 or -1: s * sh is exact, so b + s * sh is one rounding either way)
    const float a0x = __builtin_fmaf(s0[0], sh[0], b0[0]), a0y = __builtin_fmaf(s0[2], sh[0], b0[2]);
    const float a0z = __builtin_fmaf(s0[1], sh[1], b0[1]), a0w = __builtin_fmaf(s0[3], sh[1], b0[3]);
    const float a1x = __builtin_fmaf(s1[0], sh[2], b1[0]), a1y = __builtin_fmaf(s1[2], sh[2], b1[2]);
    const float a1z = __builtin_fmaf(s1[1], sh[3], b1[1]), a1w = __builtin_fmaf(s1[3], sh[3], b1[3]);
    v3 p0 = V3(a0x, a0y, h[0]), p1 = V3(a0z, a0w, h[1]), p2 = V3(a1x, a1y, h[2]), p3 = V3(a1z, a1w, h[3]);
    p0 = p0 * (1.79284291400159f - 0.85373472095314f * dot(p0, p0));       // taylorInvSqrt :473-476
    p1 = p1 * (1.79284291400159f - 0.85373472095314f * dot(p1, p1));
    p2 = p2 * (1.79284291400159f - 0.85373472095314f * dot(p2, p2));
    p3 = p3 * (1.79284291400159f - 0.85373472095314f * dot(p3, p3));
    const v3 x0 = V3(x0x, x0y, x0z), x1 = V3(x1x, x1y, x1z), x2 = V3(x2x, x2y, x2z), x3 = V3(x3x, x3y, x3z);
    float m0 = fmax_(0.6f - dot(x0, x0), 0.0f), m1 = fmax_(0.6f - dot(x1, x1), 0.0f);
    float m2 = fmax_(0.6f - dot(x2, x2), 0.0f), m3 = fmax_(0.6f - dot(x3, x3), 0.0f);
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2; m3 = m3 * m3;
    const float d0 = dot(p0, x0), d1 = dot(p1, x1), d2 = dot(p2, x2), d3 = dot(p3, x3);
    return 42.0f * ((((m0 * m0) * d0 + (m1 * m1) * d1) + (m2 * m2) * d2) + (m3 * m3) * d3);
}

template <bool XI>
__global__ void __launch_bounds__(WG_THREADS) k_clouds_best(FrameCloudsBest F, RowMap M, float* __restrict__ out) {
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();      // (the dispatch order's cost table, RowMap.cost)
    const Pixel px = pixel_of_thread(M);
    if (!px.valid) return;
    const v2 pc = point_cam(F.cam, px.fx, px.fy);
    const v3 dir = primary_dir(F.cam, pc);

    // render_sky_color :564-575
    const v3 sun_color = V3(1.f, .7f, .55f);
    const float sun_amount = fmax_(dot(dir, F.sun_dir), 0.f);
    v3 sky = mix3(V3(.0f, .1f, .4f), V3(.3f, .6f, .8f), 1.0f - dir.y);
    sky = sky + sun_color * fmin_(pow_(sun_amount, 1500.0f) * 5.0f, 1.0f);
    sky = sky + sun_color * fmin_(pow_(sun_amount, 10.0f) * .6f, 1.0f);

    v3 col = sky;
    const float cutoff = dot(dir, V3(0, 1, 0));
    if (!(cutoff < 0.05f)) {                                               // render :652-655
        // render_clouds :599-633
        const v3 projection = dir / dir.y;
        const v3 iter = projection * F.march_step;
        const v3 origin = F.cam.eye + projection * 100.f;
        float pos_x = origin.x, pos_z = origin.z;
        float T = 1.f, C = 0.f, alpha = 0.f;                               // C: the three channels are equal
        float cd = (F.cov + .035f) - F.cov, cr = 1.0f / ((F.cov + .035f) - F.cov);        // div3_'s divisor and reciprocal, in VGPRs
        asm volatile("" : "+v"(cd), "+v"(cr));
        for (int i = 0; i < CB_STEPS; ++i) {
            const CBRow& row = F.row[i];
            // density_func :577-589 : p = pos * .001 + wind ; fbm_clouds(p * 2.032, 2.6434, .5, .5)
            float qx = (pos_x * .001f + 0.f) * 2.032f;
            float qz = (pos_z * .001f + F.wind_z) * 2.032f;
            float dens = 0.f, H = .5f;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                dens = __builtin_fmaf(abs_(snoise<XI>(qx, row.qy[k], qz)), H, dens);   // |n| * 2^-(k+1) is exact: `dens += |n| * H` in one rounding
                qx = qx * 2.6434f; qz = qz * 2.6434f;
                H *= .5f;
            }
            // XI frames (finite, bounded: launch_clouds_best): the smoothstep's division through div3_ and the exp through exp_reg4k_ of
            // sbx_math.h, both equal to the forms of the other branch on their domains — dens is a sum of |simplex noise| <= 1 with gains
            // .5 ... .03125, so dens - cov is zero or >= 2^-26 in magnitude and |dens * march_step| < 80 (checked on the host).
            if (XI && CB_FAST_MATH) dens = x_smoothstep_d3_med3(F.cov, cd, cr, dens);      // (dens * smoothstep: a NaN dens stays NaN)
            else dens = dens * smoothstep_rd(F.cov, F.cov_rd, dens);
            // integrate_volume :392-407
            const float T_i = (XI && CB_FAST_MATH) ? exp_reg4k_((-1.f * dens) * F.march_step, kExp2Tab4096) : exp_((-1.f * dens) * F.march_step);
            T *= T_i;
            C += ((T * row.illum) * dens) * F.march_step;
            alpha += (1.f - T_i) * (1.f - alpha);
            pos_x += iter.x; pos_z += iter.z;
            if (alpha > .999f) break;
        }
        const float a = alpha * smoothstep_(.0f, .2f, cutoff);
        col = mix3(sky, V3s(C), a);                                        // :658
    }
    tile_cost_store(M, tl_t0);
    store_rgba(M, out, px.idx, to_srgb(col));
}

dim3 clouds_best_grid(const RowMap& M) { return grid_for(M); }

void launch_clouds_best(const FrameCloudsBest& F, const RowMap& M, float* out, hipStream_t s) {
    // XI: lattice coordinates below 2^22.  x, z: |pos| * .001 <= (|eye| + 20 * 100 + 20 * 50 * |march_step|) * .001, plus the
    // wind offset, times 2.032 * 2.6434^4 (the finest octave), times 2 (the skew v + dot(v, 1/3)); y: the table's rows.
    double far = (std::fabs((double)F.cam.eye.x) + std::fabs((double)F.cam.eye.y) + std::fabs((double)F.cam.eye.z) + 2100.0 +
                  21.0 * CB_STEPS * std::fabs((double)F.march_step)) * .001 + std::fabs((double)F.wind_z);
    for (int i = 0; i < CB_STEPS; ++i) far = std::fmax(far, std::fabs((double)F.row[i].qy[0]) / 2.032);
    const bool xi = far * (2.032 * 48.83 * 2.0) < 4194304.0 &&       // NaN compares false
                    std::fabs((double)F.march_step) <= 40.0 && F.cov >= 0x1p-20f && F.cov <= 0x1p20f;   // exp_reg4k_'s / div3_'s domains
    if (xi) hipLaunchKernelGGL(k_clouds_best<true>, grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
    else hipLaunchKernelGGL(k_clouds_best<false>, grid_for(M), dim3(WG_THREADS), 0, s, F, M, out);
}

}  // namespace sbx
