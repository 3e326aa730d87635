// rtm.h — the small slice of linear algebra the host scene/camera code needs.
//
// The reference host uses GLM (an un-vendored system dependency, version unpinned:
// Makefile:3) for vec3/vec4/mat4, cross/normalize/clamp/radians, translate/scale,
// yawPitchRoll, lookAt, inverse and perspective (src/scene.cpp:5-6,820-832;
// src/main.cpp:1466-1497).  This header restates the published GLM formulas for exactly
// those functions in plain binary32 arithmetic, so the scene vertices and the camera
// block are produced by the same operations.  Matrices are column-major (m.c[col]).
#pragma once
#include <cmath>

namespace rtm {

struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct mat4 { vec4 c[4]; };

inline vec3 v3(float x, float y, float z) { return {x, y, z}; }
inline vec4 v4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline vec4 v4(vec3 a, float w) { return {a.x, a.y, a.z, w}; }
inline vec3 xyz(vec4 a) { return {a.x, a.y, a.z}; }

inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
// glm::normalize(v) = v * inversesqrt(dot(v, v)), inversesqrt(x) = 1 / sqrt(x)
inline vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline float clampf(float x, float lo, float hi) { return std::fmin(std::fmax(x, lo), hi); }
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

inline mat4 identity() {
    return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};
}
// glm mat4 * vec4: (m0*x + m1*y) + (m2*z + m3*w)
inline vec4 mul(const mat4& m, vec4 v) { return (m.c[0] * v.x + m.c[1] * v.y) + (m.c[2] * v.z + m.c[3] * v.w); }
// glm mat4 * mat4: column j = a0*b[j].x + a1*b[j].y + a2*b[j].z + a3*b[j].w
inline mat4 mul(const mat4& a, const mat4& b) {
    mat4 r;
    for (int j = 0; j < 4; ++j)
        r.c[j] = a.c[0] * b.c[j].x + a.c[1] * b.c[j].y + a.c[2] * b.c[j].z + a.c[3] * b.c[j].w;
    return r;
}
// glm::translate(m, v): m[3] = m0*v.x + m1*v.y + m2*v.z + m3
inline mat4 translate(const mat4& m, vec3 v) {
    mat4 r = m;
    r.c[3] = m.c[0] * v.x + m.c[1] * v.y + m.c[2] * v.z + m.c[3];
    return r;
}
inline mat4 scale(const mat4& m, vec3 v) {
    mat4 r;
    r.c[0] = m.c[0] * v.x; r.c[1] = m.c[1] * v.y; r.c[2] = m.c[2] * v.z; r.c[3] = m.c[3];
    return r;
}
// glm::yawPitchRoll (gtx/euler_angles)
inline mat4 yawPitchRoll(float yaw, float pitch, float roll) {
    float ch = std::cos(yaw), sh = std::sin(yaw);
    float cp = std::cos(pitch), sp = std::sin(pitch);
    float cb = std::cos(roll), sb = std::sin(roll);
    mat4 r;
    r.c[0] = {ch * cb + sh * sp * sb, sb * cp, -sh * cb + ch * sp * sb, 0.0f};
    r.c[1] = {-ch * sb + sh * sp * cb, cb * cp, sb * sh + ch * sp * cb, 0.0f};
    r.c[2] = {sh * cp, -sp, ch * cp, 0.0f};
    r.c[3] = {0.0f, 0.0f, 0.0f, 1.0f};
    return r;
}
// glm::lookAt, right-handed
inline mat4 lookAt(vec3 eye, vec3 center, vec3 up) {
    vec3 f = normalize(center - eye);
    vec3 s = normalize(cross(f, up));
    vec3 u = cross(s, f);
    mat4 r = identity();
    r.c[0].x = s.x; r.c[1].x = s.y; r.c[2].x = s.z;
    r.c[0].y = u.x; r.c[1].y = u.y; r.c[2].y = u.z;
    r.c[0].z = -f.x; r.c[1].z = -f.y; r.c[2].z = -f.z;
    r.c[3].x = -dot(s, eye); r.c[3].y = -dot(u, eye); r.c[3].z = dot(f, eye);
    return r;
}
// glm::perspective, right-handed, depth -1..1
inline mat4 perspective(float fovy, float aspect, float zn, float zf) {
    float t = std::tan(fovy / 2.0f);
    mat4 r = {{{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}};
    r.c[0].x = 1.0f / (aspect * t);
    r.c[1].y = 1.0f / t;
    r.c[2].z = -(zf + zn) / (zf - zn);
    r.c[2].w = -1.0f;
    r.c[3].z = -(2.0f * zf * zn) / (zf - zn);
    return r;
}
// 4x4 inverse by cofactors (adjugate / determinant)
inline mat4 inverse(const mat4& m) {
    const float* a = &m.c[0].x;  // a[col*4 + row]
    float inv[16];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    float det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    float id = 1.0f / det;
    mat4 r;
    float* o = &r.c[0].x;
    for (int i = 0; i < 16; ++i) o[i] = inv[i] * id;
    return r;
}

}  // namespace rtm
