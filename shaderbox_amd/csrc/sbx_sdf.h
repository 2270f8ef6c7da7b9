// shaderbox_amd/csrc/sbx_sdf.h — the SDF library of /root/reference/src/sdf.h for the kernels, with the
// point-independent part of the costly primitives split off into per-frame "frames" (host-evaluated).
#pragma once
#include "sbx_frame.h"
#include "sbx_witness.h"

namespace sbx {

struct D2 { float d, m; };   // (distance, material id); op_add keeps the nearer              sdf.h:5-11

__device__ __forceinline__ D2 op_add2(D2 a, D2 b) { return a.d < b.d ? a : b; }

__device__ __forceinline__ float op_blend(float a, float b, float k) {                         // sdf.h:38-47
    float h = clamp_(0.5f + 0.5f * (b - a) / k, 0.0f, 1.0f);
    return mix_(b, a, h) - k * h * (1.0f - h);
}
// HW = true: v_max_f32 / v_min_f32 instead of the spec's compare-and-select (sbx_math.h fmin_ / fmax_: two half-rate instructions and
// a VCC hazard each).  The hardware forms return the same value unless the FIRST operand is a NaN or the operands are zeros of
// opposite sign ((+0, -0) for min, (-0, +0) for max).  Callers pass HW only where neither can happen: a finite point p (no NaN: these
// primitives only add, subtract, multiply and take the root of a sum of squares), and every zero in them is +0 — a difference x - y
// of equal numbers is +0, |p| - b is never -0, the literal zeros are +0.
// The instructions are PINNED with inline asm: LLVM's minnum / maxnum leave the sign of a zero result and the NaN case to the
// target, so a constant fold or a compiler upgrade could pick the other zero; v_min_f32 / v_max_f32 are what the proof is about.
__device__ __forceinline__ float v_max_f32_(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float v_min_f32_(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// max(a, -b) with the negation as the instruction's source modifier (what the compiler did for the builtin; a separate v_xor per
// op_sub otherwise)
__device__ __forceinline__ float v_max_f32_neg_(float a, float b) { float r; asm("v_max_f32 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <bool HW> __device__ __forceinline__ float hmax_(float a, float b) { return HW ? v_max_f32_(a, b) : fmax_(a, b); }
template <bool HW> __device__ __forceinline__ float hmax_neg_(float a, float b) { return HW ? v_max_f32_neg_(a, b) : fmax_(a, -b); }
template <bool HW> __device__ __forceinline__ float hmin_(float a, float b) { return HW ? v_min_f32_(a, b) : fmin_(a, b); }
template <bool HW = false>
__device__ __forceinline__ float sd_box(v3 p, v3 b) {                                          // sdf.h:67-73
    return hmax_<HW>(abs_(p.x) - b.x, hmax_<HW>(abs_(p.y) - b.y, abs_(p.z) - b.z));
}
template <bool HW, class W>
__device__ __forceinline__ float sd_y_cylinder(v3 p, float r, float h, W& w) {                  // sdf.h:85-93
    return hmax_<HW>(w.length(V2(p.x, p.z)) - r, abs_(p.y) - h / 2.f);
}
template <bool HW = false>
__device__ __forceinline__ float sd_y_cylinder(v3 p, float r, float h) {
    Wit<false> w;
    return sd_y_cylinder<HW>(p, r, h, w);
}
__device__ __forceinline__ float det2(v2 a, v2 b) { return a.x * b.y - b.x * a.y; }             // sdf.h:114-119

// sd_bezier(a, b, c, p, thickness).x, point-dependent part (frame = bezier_frame(a, b, c))    sdf.h:120-159
template <class W>
__device__ __forceinline__ float sd_bezier_x(const BezierFrame& B, v3 p, float thickness, W& w) {
    const v3 q = p - B.b;
    const v3 p3 = V3(dot(q, B.u), dot(q, B.v), dot(q, B.w));
    const v2 pxy = V2(p3.x, p3.y);
    const v2 b0 = B.a2 - pxy, b1 = V2(0.f, 0.f) - pxy, b2 = B.c2 - pxy;
    const float a = det2(b0, b2);                                // sd_bezier_get_closest :120-139
    const float b = 2.0f * det2(b1, b0);
    const float d = 2.0f * det2(b2, b1);
    const float f = b * d - a * a;
    const v2 d21 = b2 - b1, d10 = b1 - b0, d20 = b2 - b0;
    v2 gf = 2.0f * (b * d21 + d * d10 + a * d20);
    gf = V2(gf.y, -gf.x);
    const v2 pp = (-f * gf) / dot(gf, gf);
    const v2 d0p = b0 - pp;
    const float ap = det2(d0p, d20);
    const float bp = 2.0f * det2(d10, d0p);
    const float t = clamp_((ap + bp) / (2.0f * a + b + d), 0.0f, 1.0f);
    const v2 cp = mix2(mix2(b0, b1, t), mix2(b1, b2, t), t);
    return 0.85f * (w.sqrt(dot(cp, cp) + p3.z * p3.z) - thickness);
}
__device__ __forceinline__ float sd_bezier_x(const BezierFrame& B, v3 p, float thickness) {
    Wit<false> w;
    return sd_bezier_x(B, p, thickness, w);
}
// Can the tube be left out of a union whose other members already give distance dmin >= 0 at p?  sd_bezier returns
// .85 * (D - thickness) with D the distance from p to a point of the curve, and the curve lies inside the sphere
// (bc, br) around its control triangle, so D >= |p - bc| - br.  With K = 1.18 (dmin + 1e-3) + 1.002 thickness +
// br + 1e-3, |p - bc| > K implies .85 * (D - thickness) > dmin + 1e-3 with room for every rounding on the way
// (1.18 > 1 / .85; br is already rounded up by .1 %; scene scale ~10): the tube cannot be the minimum, and a
// union (op_add2: `a.d < b.d ? a : b`) that gets +inf instead returns the same member.  NaN compares false.
__device__ __forceinline__ bool bezier_far(const BezierFrame& B, v3 p, float thickness, float dmin) {
    const v3 q = p - B.bc;
    const float K = (dmin + 1e-3f) * 1.18f + (thickness * 1.002f + B.br + 1e-3f);
    return dmin >= 0.f && dot(q, q) > K * K;
}
// sd_cylinder(P, 0, P1, R), point-dependent part (frame = cyl_frame(0, P1))                    sdf.h:95-109
template <bool HW, class W>      // (HW: dist = a length is never -0 and the negated planes are second operands)
__device__ __forceinline__ float sd_cylinder0(const CylFrame& C, v3 P, float R, W& w) {
    const float dist = w.length(cross(C.dir, P - V3(0.f, 0.f, 0.f)));
    const float plane_1 = dot(C.dir, P) + C.len1;
    const float plane_2 = dot(-C.dir, P) + (-C.len0);
    return hmax_neg_<HW>(hmax_neg_<HW>(dist, plane_1), plane_2) - R;   // op_sub(op_sub(dist, p1), p2) - R
}
template <bool HW = false>
__device__ __forceinline__ float sd_cylinder0(const CylFrame& C, v3 P, float R) {
    Wit<false> w;
    return sd_cylinder0<HW>(C, P, R, w);
}
// sd_capsule(p, a, b, r) with ab = b - a and rd = recip64(dot(ab, ab)) from the frame         sdf.h:162-171
template <class W>
__device__ __forceinline__ float sd_capsule_f(v3 p, v3 a, v3 ab, double rd, float r, W& w) {
    const float t = clamp_(div_by(dot(p - a, ab), rd), 0.f, 1.f);
    return w.length((ab * t + a) - p) - r;
}
__device__ __forceinline__ float sd_capsule_f(v3 p, v3 a, v3 ab, double rd, float r) {
    Wit<false> w;
    return sd_capsule_f(p, a, ab, rd, r, w);
}

}  // namespace sbx
