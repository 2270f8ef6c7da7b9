/* oracle/ovec.h — the small GLSL-like value types the CPU oracle is written in.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * The reference's shader headers assume an environment that supplies vec2/vec3/vec4,
 * mat3 and the GLSL built-ins (VML on the C++ path; /root/reference/README.md:8-9,
 * binding documented at /root/reference/util/ddsvolgen/src/ddsvolgen.cpp:26-38).
 * This header is that environment for the oracle, with every operation defined by the
 * sbx math spec (oracle/sbx_math_ref.h):
 *   - mat3 is column-major, m.c[col][row]  (GLSL);  M*v = sum_c col_c * v_c;
 *     v*M = (dot(v,col0), dot(v,col1), dot(v,col2));
 *   - dot(a,b) = ((a0*b0 + a1*b1) + a2*b2);   length = sqrt(dot(v,v));
 *   - normalize(v) = v / length(v)   (three true divisions).
 */
#ifndef SBX_OVEC_H
#define SBX_OVEC_H
#include "sbx_math_ref.h"

namespace sbxref {

struct vec2 {
    float x, y;
    vec2() : x(0), y(0) {}
    vec2(float a, float b) : x(a), y(b) {}
};
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    vec3(vec2 v, float c) : x(v.x), y(v.y), z(c) {}
    vec2 xy() const { return vec2(x, y); }
    vec2 xz() const { return vec2(x, z); }
};
struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(vec3 v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    vec3 rgb() const { return vec3(x, y, z); }
};

static inline vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }
static inline vec2 operator-(vec2 a, vec2 b) { return vec2(a.x - b.x, a.y - b.y); }
static inline vec2 operator*(vec2 a, vec2 b) { return vec2(a.x * b.x, a.y * b.y); }
static inline vec2 operator/(vec2 a, vec2 b) { return vec2(a.x / b.x, a.y / b.y); }
static inline vec2 operator*(vec2 a, float s) { return vec2(a.x * s, a.y * s); }
static inline vec2 operator*(float s, vec2 a) { return vec2(s * a.x, s * a.y); }
static inline vec2 operator/(vec2 a, float s) { return vec2(a.x / s, a.y / s); }
static inline vec2 operator-(vec2 a, float s) { return vec2(a.x - s, a.y - s); }
static inline vec2 operator-(vec2 a) { return vec2(-a.x, -a.y); }

static inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
static inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
static inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
static inline vec3 operator+(vec3 a, float s) { return vec3(a.x + s, a.y + s, a.z + s); }
static inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
static inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
static inline vec3& operator-=(vec3& a, vec3 b) { a = a - b; return a; }
static inline vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }
static inline vec3& operator*=(vec3& a, float s) { a = a * s; return a; }
static inline vec3& operator+=(vec3& a, float s) { a = a + s; return a; }

static inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
static inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float length(vec2 v) { return m_sqrt(dot(v, v)); }
static inline float length(vec3 v) { return m_sqrt(dot(v, v)); }
static inline vec3 normalize(vec3 v) { return v / length(v); }
static inline vec3 cross(vec3 a, vec3 b) {
    return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline vec3 vabs(vec3 v) { return vec3(m_abs(v.x), m_abs(v.y), m_abs(v.z)); }
static inline vec3 vmix(vec3 a, vec3 b, float t) {
    return vec3(m_mix(a.x, b.x, t), m_mix(a.y, b.y, t), m_mix(a.z, b.z, t));
}
static inline vec2 vmix(vec2 a, vec2 b, float t) { return vec2(m_mix(a.x, b.x, t), m_mix(a.y, b.y, t)); }
static inline vec3 vfloor(vec3 v) { return vec3(m_floor(v.x), m_floor(v.y), m_floor(v.z)); }
static inline vec3 vfract(vec3 v) { return vec3(m_fract(v.x), m_fract(v.y), m_fract(v.z)); }
static inline vec3 vexp(vec3 v) { return vec3(m_exp(v.x), m_exp(v.y), m_exp(v.z)); }

struct mat3 {
    vec3 c[3]; /* columns */
    mat3() {}
    /* GLSL constructor order: column by column */
    mat3(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2) {
        c[0] = vec3(a0, a1, a2); c[1] = vec3(b0, b1, b2); c[2] = vec3(c0, c1, c2);
    }
};
/* M * v */
static inline vec3 mul(const mat3& m, vec3 v) { return (m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z; }
/* v * M */
static inline vec3 mul(vec3 v, const mat3& m) { return vec3(dot(v, m.c[0]), dot(v, m.c[1]), dot(v, m.c[2])); }
/* A * B : column j of the product is A * (column j of B) */
static inline mat3 mul(const mat3& a, const mat3& b) {
    mat3 r;
    r.c[0] = mul(a, b.c[0]); r.c[1] = mul(a, b.c[1]); r.c[2] = mul(a, b.c[2]);
    return r;
}

} /* namespace sbxref */
#endif
