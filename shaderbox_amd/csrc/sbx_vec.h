// shaderbox_amd/csrc/sbx_vec.h — register-resident vector types for the kernels.
//
// GLSL semantics as fixed by the math spec: dot = ((a0*b0 + a1*b1) + a2*b2),
// normalize = v / sqrt(dot(v,v)) with three true divisions, mat3 column-major with
// M*v = (c0*v.x + c1*v.y) + c2*v.z and v*M = (dot(v,c0), dot(v,c1), dot(v,c2)).
#pragma once
#include "sbx_math.h"

namespace sbx {

struct v2 { float x, y; };
struct v3 { float x, y, z; };
struct v4 { float x, y, z, w; };
struct m3 { v3 c0, c1, c2; };

SBX_HD v2 V2(float x, float y) { return v2{x, y}; }
SBX_HD v3 V3(float x, float y, float z) { return v3{x, y, z}; }
SBX_HD v3 V3s(float s) { return v3{s, s, s}; }

SBX_HD v2 operator+(v2 a, v2 b) { return {a.x + b.x, a.y + b.y}; }
SBX_HD v2 operator-(v2 a, v2 b) { return {a.x - b.x, a.y - b.y}; }
SBX_HD v2 operator*(v2 a, float s) { return {a.x * s, a.y * s}; }
SBX_HD v2 operator*(float s, v2 a) { return {s * a.x, s * a.y}; }
SBX_HD v2 operator/(v2 a, float s) { return {a.x / s, a.y / s}; }
SBX_HD v2 operator-(v2 a) { return {-a.x, -a.y}; }

SBX_HD v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
SBX_HD v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
SBX_HD v3 operator*(v3 a, v3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
SBX_HD v3 operator*(v3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
SBX_HD v3 operator*(float s, v3 a) { return {s * a.x, s * a.y, s * a.z}; }
SBX_HD v3 operator/(v3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
SBX_HD v3 operator+(v3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
SBX_HD v3 operator+(float s, v3 a) { return {s + a.x, s + a.y, s + a.z}; }
SBX_HD v3 operator-(v3 a) { return {-a.x, -a.y, -a.z}; }

SBX_HD float dot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
SBX_HD float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
SBX_HD float length(v2 v) { return sqrt_(dot(v, v)); }
SBX_HD float length(v3 v) { return sqrt_(dot(v, v)); }
// three binary32 divisions by the same denominator: one binary64 reciprocal + three exact div_by (sbx_math.h) give the
// IEEE quotients bit for bit (89 against 126 issue cycles on MI355X)
SBX_HD v3 normalize(v3 v) {
    const float l = length(v);
#if defined(SBX_PLAIN_NORMALIZE)
    return {v.x / l, v.y / l, v.z / l};
#else
    const double rl = recip64(l);
    return {div_by(v.x, rl), div_by(v.y, rl), div_by(v.z, rl)};
#endif
}
SBX_HD v3 cross(v3 a, v3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
SBX_HD v3 abs3(v3 v) { return {abs_(v.x), abs_(v.y), abs_(v.z)}; }
SBX_HD v3 mix3(v3 a, v3 b, float t) { return {mix_(a.x, b.x, t), mix_(a.y, b.y, t), mix_(a.z, b.z, t)}; }
SBX_HD v2 mix2(v2 a, v2 b, float t) { return {mix_(a.x, b.x, t), mix_(a.y, b.y, t)}; }

SBX_HD m3 M3(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2) {
    return m3{{a0, a1, a2}, {b0, b1, b2}, {c0, c1, c2}};
}
SBX_HD v3 mul(const m3& m, v3 v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }
SBX_HD v3 mul(v3 v, const m3& m) { return {dot(v, m.c0), dot(v, m.c1), dot(v, m.c2)}; }
SBX_HD m3 mul(const m3& a, const m3& b) { return m3{mul(a, b.c0), mul(a, b.c1), mul(a, b.c2)}; }
SBX_HD m3 transpose(const m3& m) {
    return M3(m.c0.x, m.c1.x, m.c2.x, m.c0.y, m.c1.y, m.c2.y, m.c0.z, m.c1.z, m.c2.z);
}
// rotations take DEGREES (/root/reference/src/util.h:44-69)
SBX_HD m3 rotate_around_z(float deg) {
    float a = radians_(deg), s = sin_(a), c = cos_(a);
    return M3(c, -s, 0, s, c, 0, 0, 0, 1);
}
SBX_HD m3 rotate_around_y(float deg) {
    float a = radians_(deg), s = sin_(a), c = cos_(a);
    return M3(c, 0, s, 0, 1, 0, -s, 0, c);
}
SBX_HD m3 rotate_around_x(float deg) {
    float a = radians_(deg), s = sin_(a), c = cos_(a);
    return M3(1, 0, 0, 0, c, -s, 0, s, c);
}

}  // namespace sbx
