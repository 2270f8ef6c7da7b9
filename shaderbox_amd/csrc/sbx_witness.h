// shaderbox_amd/csrc/sbx_witness.h — fast forms of sqrt / normalize with a RECORDED domain, for kernels that cannot prove the domain.
#pragma once
#include "sbx_frame.h"

namespace sbx {

// WITNESSED SQUARE ROOTS (round 4).  The compiler's IEEE sqrt is 16 VALU instructions (input scaling for tiny arguments, v_sqrt_f32,
// a two-sided one-ulp fix-up, a class test for 0 / inf); sqrt_rs_ (sbx_math.h) is 5 and EQUAL to it for every argument in
// [2^-102, +inf) — all of them were run (profiles/r03_sqrt_rsq_exhaustive.txt) — but returns NaN for 0 and +inf and is inexact
// below 2^-102.  A squared length in an SDF is none of those except ON a primitive's axis or centre, which no kernel can rule out
// for an arbitrary frame.  So a lane RECORDS every argument outside the proved interval (two integer instructions, the flag
// accumulates in an SGPR pair) and the kernel, after its whole pixel, re-runs the pixel with the IEEE forms if any lane of the wave
// recorded one (wave-uniform branch; never taken on the frames measured).  No branch inside the SDF — a per-lane choice between
// the forms at each root cost more than it saved (sbx_math.h, sqrt_n_) — and a pixel's bits are the IEEE forms' either way:
// without a record every root it took is one of the exhaustively compared ones.
// Wit<false> is the plain form (no record), what every caller without a witness gets.
template <bool FAST> struct Wit {
    static constexpr bool fast = FAST;
    bool bad = false;
    unsigned lo = 0x0C800000u;                                   // 2^-102; the test build raises it (k_egg<., 2>) to exercise the re-run
    __device__ __forceinline__ float sqrt(float x) {
        if (!FAST) return sqrt_(x);
        bad |= (f2u(x) - lo) >= (0x7F800000u - lo);              // 0, tiny, +inf, NaN, negative: all outside [lo, +inf)
        return sqrt_rs_(x);
    }
    __device__ __forceinline__ float length(v2 v) { return sqrt(dot(v, v)); }
    __device__ __forceinline__ float length(v3 v) { return sqrt(dot(v, v)); }
    // normalize(v) = v / length(v) (sbx_vec.h: IEEE root, then three IEEE quotients through one binary64 reciprocal: ~150 issue
    // cycles).  Fast form: sqrt_rs_, v_rcp_f32 + one Newton step, three div3_ (sbx_math.h divn_: EQUAL to the IEEE quotient for
    // every pair of significands, all 2^47 run, wherever nothing under- or overflows and the dividend is not a zero, whose sign
    // div3_ loses) — 22 instructions.  Recorded: a squared length outside [lo, 2^40) and any component whose SQUARE is not a normal
    // number (zero, denormal, NaN).  Without a record: l in [2^-51, 2^20], every |v_i| in [2^-63, l], every quotient in [2^-83, 1],
    // the residual a - q0 l a multiple of 2^-109 that fits 24 bits — all normal, which is the scale-free case the exhaustive run
    // covers.
    __device__ __forceinline__ v3 normalize(v3 v) {
        if (!FAST) return sbx::normalize(v);
        const float xx = v.x * v.x, yy = v.y * v.y, zz = v.z * v.z;
        const float x = (xx + yy) + zz;                          // dot(v, v), operation for operation (sbx_vec.h)
        const float mn = u2f(0x00800000u);                       // 2^-126
        bad |= !(xx >= mn) || !(yy >= mn) || !(zz >= mn);
        bad |= (f2u(x) - lo) >= (0x53800000u - lo);              // [lo, 2^40)
#if defined(__HIP_DEVICE_COMPILE__)
        const float l = sqrt_rs_(x);
        const float r0 = __builtin_amdgcn_rcpf(l);
        const float e = __builtin_fmaf(-l, r0, 1.0f);
        const float r = __builtin_fmaf(e, r0, r0);
        return V3(div3_(v.x, l, r), div3_(v.y, l, r), div3_(v.z, l, r));
#else
        return sbx::normalize(v);
#endif
    }
};

// primary_dir (sbx_frame.h) with the witness's normalisation
template <class W>
__device__ __forceinline__ v3 primary_dir(const Camera& c, v2 pc, W& w) {
    return w.normalize(c.fwd + c.up * pc.y + c.right * pc.x);    // util.h:17
}

}  // namespace sbx
