// shaderbox_amd/csrc/sbx_noise.h — value noise + fBm for the kernels
// (/root/reference/src/noise_iq.h:5-29, src/fbm.h:6).
#pragma once
#include "sbx_vec.h"

namespace sbx {

// hash(n) = fract(sin(n) * 753.5453123)                                   noise_iq.h:5-9
#ifdef SBX_ABLATE_SIN     // timing experiment only (wrong pixels): what the sin of the lattice hash costs a kernel
__device__ __forceinline__ float hash1(float n) { return fract_((n * .318f) * 753.5453123f); }
#else
__device__ __forceinline__ float hash1(float n) { return fract_(sin_(n) * 753.5453123f); }
#endif
// hash1 for callers that have shown |n| <= 2^40 (B40; the lattice index of a bounded position): sin_b40_ of sbx_math.h, equal to
// sin_ on that whole range, three binary64 fma cheaper.  (Both forms behind a wave-wide test of n cost the hash pass' callers
// 40-60 B of scratch: the choice is a template parameter decided from the host's domain checks instead.)
template <bool B40>
__device__ __forceinline__ float hash1_b(float n) {
#if defined(SBX_ABLATE_SIN) || defined(SBX_NO_SIN_B40)
    return hash1(n);
#else
    return B40 ? fract_(sin_b40_(n) * 753.5453123f) : hash1(n);
#endif
}


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
