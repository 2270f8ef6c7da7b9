// shaderbox_amd/csrc/sbx_noise.h — value noise + fBm for the kernels
// (/root/reference/src/noise_iq.h:5-29, src/fbm.h:6).
#pragma once
#include "sbx_vec.h"

namespace sbx {

// hash(n) = fract(sin(n) * 753.5453123)                                   noise_iq.h:5-9
__device__ __forceinline__ float hash1(float n) { return fract_(sin_(n) * 753.5453123f); }

// trilinear value noise over the 8 lattice corners                        noise_iq.h:11-29
// (1 - f) is written once per axis instead of once per mix(): same value, same bits.
__device__ __forceinline__ float noise_iq(v3 x) {
    const float px = floor_(x.x), py = floor_(x.y), pz = floor_(x.z);
    float fx = x.x - px, fy = x.y - py, fz = x.z - pz;        // fract = x - floor(x)
    fx = fx * fx * tm2_(fx);
    fy = fy * fy * tm2_(fy);
    fz = fz * fz * tm2_(fz);
    const float n = px + py * 157.0f + 113.0f * pz;
    const float gx = 1.0f - fx, gy = 1.0f - fy, gz = 1.0f - fz;
    const float h000 = hash1(n + 0.0f), h100 = hash1(n + 1.0f);
    const float h010 = hash1(n + 157.0f), h110 = hash1(n + 158.0f);
    const float h001 = hash1(n + 113.0f), h101 = hash1(n + 114.0f);
    const float h011 = hash1(n + 270.0f), h111 = hash1(n + 271.0f);
    const float a = h000 * gx + h100 * fx;
    const float b = h010 * gx + h110 * fx;
    const float c = h001 * gx + h101 * fx;
    const float d = h011 * gx + h111 * fx;
    const float ab = a * gy + b * fy;
    const float cd = c * gy + d * fy;
    return ab * gz + cd * fz;
}

// DECL_FBM_FUNC(name, OCT, basis)                                          fbm.h:6
// t += basis(p) * H; p *= lacunarity; H *= gain
template <int OCT, class Basis>
__device__ __forceinline__ float fbm(v3 pos, float lacunarity, float init_gain, float gain, Basis basis) {
    v3 p = pos;
    float H = init_gain, t = 0.f;
#pragma unroll
    for (int i = 0; i < OCT; ++i) {
        t += basis(p) * H;
        p = p * lacunarity;
        H *= gain;
    }
    return t;
}

}  // namespace sbx
