// rt_oracle.cpp — CPU restatement of the reference path tracer, shaders/raytracer.comp of
// VictorOrrios/Raytracer, function for function.  TEST INFRASTRUCTURE ONLY: it is the
// checker for the HIP kernels (tests/, the smoke test, the CPU-baseline leg of the bench);
// no product code may call, link or load it.
//
// PARITY STATUS: "parity unpinned" by the reference's own tests — the reference has no
// tests, golden images or known-answer vectors, and its shader cannot be executed in the
// build environment (GLSL needs a Vulkan driver and compiler; none is present).  This file
// is pinned instead by hand-derived known answers (tests/test_oracle_known_answers.py) and
// by fixtures generated from it (tests/golden/).
//
// Arithmetic contract (what "the shader's result" means here): every GLSL operator is one
// IEEE-754 binary32 operation evaluated in source order, no fused multiply-add (compile
// with -ffp-contract=off), division and sqrt correctly rounded.  Built-ins whose precision
// Vulkan leaves to the driver come in two selectable flavours:
//   math_mode 0: explicit binary32 polynomial evaluations (rm_* below), which a GPU can
//                reproduce bit for bit;
//   math_mode 1: the C library's sinf/cosf/tanf/acosf/exp2f/log2f, as a neutral second
//                opinion on the images.
// min/max/clamp return the non-NaN operand when exactly one is NaN (the behaviour of GPU
// min/max instructions); GLSL leaves that case undefined.  Variables the shader leaves
// unwritten (L / pdf for POINT, CONE and AREA lights, raytracer.comp:628-651; L when there
// are no lights, 583-586) are taken as zero.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "rt_oracle.h"

namespace {

const float PI = 3.14159265359f;  // raytracer.comp:4
const float PINF = std::numeric_limits<float>::infinity();

struct Ctx;  // per-invocation state (the shader's globals)

// ----------------------------------------------------------------------------- scalar helpers
inline uint32_t bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
inline float fbits(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
inline bool isnan_(float x) { return x != x; }
inline float qnan() { return fbits(0x7fc00000u); }

inline float gl_max(float x, float y) { if (isnan_(x)) return y; if (isnan_(y)) return x; return (x < y) ? y : x; }
inline float gl_min(float x, float y) { if (isnan_(x)) return y; if (isnan_(y)) return x; return (y < x) ? y : x; }
inline float gl_clamp(float x, float lo, float hi) { return gl_min(gl_max(x, lo), hi); }
inline float gl_abs(float x) { return fbits(bits(x) & 0x7fffffffu); }
inline float gl_sign(float x) { if (x > 0.0f) return 1.0f; if (x < 0.0f) return -1.0f; return 0.0f; }
inline float gl_mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }

// ----------------------------------------------------------------------------- math_mode 0
// Octant reduction (Cody & Waite, three-part pi/4) and minimax polynomials, all binary32.
const float FOPI = 1.27323954473516f;
const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
const float TRIG_MAX = 1.0e9f;

inline float sin_poly(float x, float z) {
    return ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
}
inline float cos_poly(float z) {
    return ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
}
float rm_sin(float xx) {
    float x = gl_abs(xx);
    if (!(x <= TRIG_MAX)) return xx - xx;
    bool neg = xx < 0.0f;
    int j = (int)(FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { neg = !neg; j -= 4; }
    x = ((x - y * DP1) - y * DP2) - y * DP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? cos_poly(z) : sin_poly(x, z);
    return neg ? -r : r;
}
float rm_cos(float xx) {
    float x = gl_abs(xx);
    if (!(x <= TRIG_MAX)) return xx - xx;
    bool neg = false;
    int j = (int)(FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { j -= 4; neg = !neg; }
    if (j > 1) neg = !neg;
    x = ((x - y * DP1) - y * DP2) - y * DP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? sin_poly(x, z) : cos_poly(z);
    return neg ? -r : r;
}
float rm_tan(float xx) {
    float x = gl_abs(xx);
    if (!(x <= TRIG_MAX)) return xx - xx;
    bool neg = xx < 0.0f;
    int j = (int)(FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    float z = ((x - y * DP1) - y * DP2) - y * DP3;
    float zz = z * z;
    float r;
    if (x > 1.0e-4f) {
        r = (((((9.38540185543e-3f * zz + 3.11992232697e-3f) * zz + 2.44301354525e-2f) * zz + 5.34112807005e-2f) * zz +
              1.33387994085e-1f) * zz + 3.33331568548e-1f) * zz * z + z;
    } else {
        r = z;
    }
    if (j & 2) r = -1.0f / r;
    return neg ? -r : r;
}
float rm_asin(float xx) {
    float a = gl_abs(xx);
    if (!(a <= 1.0f)) return qnan();
    float x, z;
    bool big = a > 0.5f;
    if (big) { z = 0.5f * (1.0f - a); x = std::sqrt(z); } else { x = a; z = x * x; }
    float r = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z +
               1.6666752422e-1f) * z * x + x;
    if (big) { r = r + r; r = 1.5707963267948966192f - r; }
    return (xx < 0.0f) ? -r : r;
}
float rm_acos(float x) {
    if (!(gl_abs(x) <= 1.0f)) return qnan();
    if (x < -0.5f) return PI - 2.0f * rm_asin(std::sqrt(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * rm_asin(std::sqrt(0.5f * (1.0f - x)));
    return 1.5707963267948966192f - rm_asin(x);
}
float rm_log2(float x) {
    if (isnan_(x) || x < 0.0f) return qnan();
    if (x == 0.0f) return -PINF;
    if (x == PINF) return PINF;
    uint32_t ix = bits(x);
    int e = 0;
    if ((ix & 0x7f800000u) == 0u) { x = x * 8388608.0f; ix = bits(x); e = -23; }
    e += (int)((ix >> 23) & 0xffu) - 126;
    float m = fbits((ix & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m +
                   1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m +
               3.3333331174e-1f) * m * z;
    y = y - 0.5f * z;
    const float LOG2EA = 0.44269504088896340735992f;
    float r = y * LOG2EA;
    r = r + m * LOG2EA;
    r = r + y;
    r = r + m;
    r = r + (float)e;
    return r;
}
float rm_exp2(float x) {
    if (isnan_(x)) return x;
    if (x >= 128.0f) return PINF;
    if (x < -126.0f) return 0.0f;
    float fl = std::floor(x);
    int i0 = (int)fl;
    float f = x - fl;
    if (f > 0.5f) { i0 += 1; f = f - 1.0f; }
    float p = (((((1.535336188319500e-4f * f + 1.339887440266574e-3f) * f + 9.618437357674640e-3f) * f +
                 5.550332471162809e-2f) * f + 2.402264791363012e-1f) * f + 6.931472028550421e-1f) * f + 1.0f;
    int h = i0 / 2;
    float s1 = fbits((uint32_t)(h + 127) << 23);
    float s2 = fbits((uint32_t)(i0 - h + 127) << 23);
    return (p * s1) * s2;
}

// ----------------------------------------------------------------------------- vectors
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float a) : x(a), y(a), z(a) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
};
inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec3 operator-(float s, vec3 a) { return vec3(s - a.x, s - a.y, s - a.z); }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline vec3 gl_max(vec3 a, vec3 b) { return vec3(gl_max(a.x, b.x), gl_max(a.y, b.y), gl_max(a.z, b.z)); }
inline vec3 gl_clamp(vec3 a, float lo, float hi) { return vec3(gl_clamp(a.x, lo, hi), gl_clamp(a.y, lo, hi), gl_clamp(a.z, lo, hi)); }
inline vec3 gl_mix(vec3 x, vec3 y, float a) { return vec3(gl_mix(x.x, y.x, a), gl_mix(x.y, y.y, a), gl_mix(x.z, y.z, a)); }
inline vec3 reflect(vec3 I, vec3 N) { return I - 2.0f * dot(N, I) * N; }
inline vec3 refract(vec3 I, vec3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return vec3(0.0f);
    return eta * I - (eta * d + std::sqrt(k)) * N;
}

// ----------------------------------------------------------------------------- shader structs
struct Ray { vec3 orig, dir; };                                     // raytracer.comp:37-40
struct Interval { float minV, maxV; };                              // 42-45
struct Hit { vec3 p, normal; int mat; float t; bool front_face; };  // 108-114
struct Material {                                                   // 54-82
    vec3 albedo; float albedo_a;
    vec3 subsurface; float subsurface_a;
    vec3 specular_tint; float specular_tint_a;
    vec3 emission_color; float emission_a;
    float roughness, metallic, ior, trs_weight;
};

// Everything one invocation of main() sees: the bound buffers, the uniform blocks, the
// global `seed`, plus the bookkeeping outputs the shader does not have.
struct Ctx {
    const RtoScene* sc;
    const RtoCamera* cam;
    const RtoPushConstants* pc;
    int math_mode;
    int skybox = 0;
    int width, height;
    int px, py;        // gl_GlobalInvocationID.xy
    uint32_t seed;     // raytracer.comp:189
    // bookkeeping
    uint64_t rays;
    int last_id;       // primitive id of the last hit_scene winner (-1: none)
    uint32_t path_hash;
    int primary_id;
    bool primary_taken;

    float m_sin(float x) const { return math_mode ? sinf(x) : rm_sin(x); }
    float m_cos(float x) const { return math_mode ? cosf(x) : rm_cos(x); }
    float m_tan(float x) const { return math_mode ? tanf(x) : rm_tan(x); }
    float m_acos(float x) const { return math_mode ? acosf(x) : rm_acos(x); }
    // pow(x,y) := exp2(y * log2(x)), the GLSL definition
    float m_pow(float x, float y) const { return math_mode ? exp2f(y * log2f(x)) : rm_exp2(y * rm_log2(x)); }
    vec3 m_pow(vec3 v, float y) const { return vec3(m_pow(v.x, y), m_pow(v.y, y), m_pow(v.z, y)); }
};

Material load_material(const RtoScene* sc, int i) {
    const RtoMaterial& m = sc->materials[i];
    Material r;
    r.albedo = vec3(m.albedo[0], m.albedo[1], m.albedo[2]); r.albedo_a = m.albedo[3];
    r.subsurface = vec3(m.subsurface[0], m.subsurface[1], m.subsurface[2]); r.subsurface_a = m.subsurface[3];
    r.specular_tint = vec3(m.specular_tint[0], m.specular_tint[1], m.specular_tint[2]); r.specular_tint_a = m.specular_tint[3];
    r.emission_color = vec3(m.emission_color[0], m.emission_color[1], m.emission_color[2]); r.emission_a = m.emission_color[3];
    r.roughness = m.roughness; r.metallic = m.metallic; r.ior = m.ior; r.trs_weight = m.trs_weight;
    return r;
}

// ----------------------------------------------------------------------------- RNG, raytracer.comp:193-247
uint32_t hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
uint32_t xorshift(uint32_t& state) {
    state ^= state << 13; state ^= state >> 17; state ^= state << 5;
    return state;
}
// The literal 4294967295.0 is a binary32 constant: it rounds to 2^32 (raytracer.comp:210).
float random(Ctx& c) { return (float)xorshift(c.seed) / 4294967295.0f; }

vec3 random_unit_vec(Ctx& c) {
    float phi = 2.0f * PI * random(c);
    float theta = c.m_acos(2.0f * random(c) - 1.0f);
    float sin_theta = c.m_sin(theta);
    return vec3(sin_theta * c.m_cos(phi), sin_theta * c.m_sin(phi), c.m_cos(theta));
}
vec3 random_vec_on_hemisphere(Ctx& c, vec3 normal) {
    vec3 rvec = random_unit_vec(c);
    if (dot(rvec, normal) > 0.0f) return rvec;
    return -rvec;
}

// ----------------------------------------------------------------------------- math, raytracer.comp:250-292
float power_heuristics(float a, float b) {
    float a2 = a * a, b2 = b * b;
    return a2 / gl_max((a2 + b2), 0.000001f);
}
vec3 sphere_to_cartesian(const Ctx& c, float theta, float phi) {
    float sin_t = c.m_sin(theta);
    return vec3(sin_t * c.m_cos(phi), sin_t * c.m_sin(phi), c.m_cos(theta));
}
vec3 align_to_world(vec3 X, vec3 N) {
    vec3 up;
    if (gl_abs(N.z) > 0.9999999f) up = vec3(1.0f, 0.0f, 0.0f); else up = vec3(0.0f, 0.0f, 1.0f);
    vec3 T = normalize(cross(up, N));
    vec3 B = cross(N, T);
    return T * X.x + B * X.y + N * X.z;
}
vec3 at(const Ray& r, float t) { return r.orig + t * r.dir; }                       // 297-299
bool surrounds(Interval i, float x) { return i.minV < x && i.maxV > x; }           // 310-312
void set_face_normal(Hit& h, const Ray& r, vec3 outward_normal) {                   // 322-329
    h.front_face = dot(r.dir, outward_normal) < 0.0f;
    h.normal = h.front_face ? outward_normal : -outward_normal;
}

// ----------------------------------------------------------------------------- intersection, raytracer.comp:334-508
bool hit_sphere(const RtoSphere& s, Interval ray_t, const Ray& r, Hit& rec) {
    vec3 spos(s.pos[0], s.pos[1], s.pos[2]);
    vec3 oc = spos - r.orig;
    float a = dot(r.dir, r.dir);
    float h = dot(r.dir, oc);
    float c = dot(oc, oc) - s.r * s.r;
    float discriminant = h * h - a * c;
    if (discriminant < 0.0f) return false;
    float srtd = std::sqrt(discriminant);
    float root = (h - srtd) / a;
    if (!surrounds(ray_t, root)) {
        root = (h + srtd) / a;
        if (!surrounds(ray_t, root)) return false;
    }
    rec.t = root;
    rec.p = at(r, root);
    rec.mat = s.mat;
    vec3 outward_normal = (rec.p - spos) / s.r;
    set_face_normal(rec, r, outward_normal);
    return true;
}

// Shared Moeller-Trumbore body of hit_triangle (366-406) and hit_mesh (412-449).
inline bool moller_trumbore(vec3 v0, vec3 v1, vec3 v2, Interval ray_t, const Ray& r, float& t_out, vec3& e1_out, vec3& e2_out) {
    const float EPSILON = 1e-6f;
    vec3 edge1 = v1 - v0;
    vec3 edge2 = v2 - v0;
    vec3 h = cross(r.dir, edge2);
    float a = dot(edge1, h);
    if (gl_abs(a) < EPSILON) return false;
    float f = 1.0f / a;
    vec3 s = r.orig - v0;
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    vec3 q = cross(s, edge1);
    float v = f * dot(r.dir, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t_r = f * dot(edge2, q);
    if (!surrounds(ray_t, t_r)) return false;
    t_out = t_r;
    e1_out = edge1;
    e2_out = edge2;
    return true;
}

bool hit_triangle(const RtoTriangle& t, Interval ray_t, const Ray& r, Hit& rec) {
    float t_r; vec3 e1, e2;
    if (!moller_trumbore(vec3(t.v0[0], t.v0[1], t.v0[2]), vec3(t.v1[0], t.v1[1], t.v1[2]), vec3(t.v2[0], t.v2[1], t.v2[2]), ray_t, r, t_r, e1, e2))
        return false;
    rec.t = t_r;
    rec.p = at(r, t_r);
    rec.mat = t.mat;
    set_face_normal(rec, r, vec3(t.normal[0], t.normal[1], t.normal[2]));
    return true;
}

bool hit_mesh(const Ctx& c, const RtoMeshInfo& mesh_info, Interval ray_t, const Ray& r, Hit& rec, int& tri_index) {
    bool hit_anything = false;
    float closest_so_far = ray_t.maxV;
    const RtoScene* sc = c.sc;
    // i + 2 < index_end: a trailing partial triple would read beyond index_end in the shader
    for (uint32_t i = mesh_info.index_start; i + 2 < mesh_info.index_end; i += 3) {
        const float* p0 = sc->vertices[sc->indices[i]].pos;
        const float* p1 = sc->vertices[sc->indices[i + 1]].pos;
        const float* p2 = sc->vertices[sc->indices[i + 2]].pos;
        float t_r; vec3 e1, e2;
        if (!moller_trumbore(vec3(p0[0], p0[1], p0[2]), vec3(p1[0], p1[1], p1[2]), vec3(p2[0], p2[1], p2[2]), ray_t, r, t_r, e1, e2))
            continue;
        if (t_r > closest_so_far) continue;
        closest_so_far = t_r;
        hit_anything = true;
        rec.t = t_r;
        rec.p = at(r, t_r);
        vec3 outward_normal = normalize(cross(e1, e2));
        set_face_normal(rec, r, outward_normal);
        tri_index = (int)(i / 3);
    }
    rec.mat = mesh_info.material;
    return hit_anything;
}

bool hit_scene(Ctx& c, const Ray& r, Interval ray_t, Hit& rec) {
    Hit temp_rec;
    temp_rec.mat = 0; temp_rec.t = 0; temp_rec.front_face = false;
    bool hit_anything = false;
    float closest_so_far = ray_t.maxV;
    const RtoPushConstants* pc = c.pc;
    const RtoScene* sc = c.sc;
    int id = -1;
    for (int i = 0; i < pc->total_spheres; i++) {
        if (hit_sphere(sc->spheres[i], ray_t, r, temp_rec)) {
            hit_anything = true;
            if (closest_so_far > temp_rec.t) { closest_so_far = temp_rec.t; rec = temp_rec; id = i; }
        }
    }
    for (int i = 0; i < pc->total_triangles; i++) {
        if (hit_triangle(sc->triangles[i], ray_t, r, temp_rec)) {
            hit_anything = true;
            if (closest_so_far > temp_rec.t) { closest_so_far = temp_rec.t; rec = temp_rec; id = pc->total_spheres + i; }
        }
    }
    for (int i = 0; i < pc->total_meshes; i++) {
        int tri = -1;
        if (hit_mesh(c, sc->meshes[i], ray_t, r, temp_rec, tri)) {
            hit_anything = true;
            if (closest_so_far > temp_rec.t) { closest_so_far = temp_rec.t; rec = temp_rec; id = pc->total_spheres + pc->total_triangles + tri; }
        }
    }
    // bookkeeping (not in the shader): ray count, id of the winner, hash of all winners
    c.rays++;
    c.last_id = id;
    c.path_hash = (c.path_hash ^ (uint32_t)(id + 1)) * 0x01000193u;
    return hit_anything;
}

bool shadow_ray(Ctx& c, const Ray& r) {  // 511-515
    Hit h;
    Interval i = {0.005f, PINF};
    return hit_scene(c, r, i, h);
}

// raytracer.comp:520-577: the five skybox functions; skybox_color() calls the grey one (575-577),
// `kind` makes the one-line choice a parameter.  smoothstep and mix per the GLSL definitions.
inline float gl_smoothstep(float e0, float e1, float x) {
    const float t = gl_clamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
vec3 skybox_color(int kind, const Ray& r) {
    if (kind == 0) return vec3(0.3f);  // skybox_color_grey, 570-572
    if (kind == 3) return vec3(1.0f);  // skybox_color_white, 564-566
    if (kind == 4) return vec3(0.0f);  // skybox_color_black, 567-569
    const float size = 0.999f, bleed = 0.0003f;
    const vec3 dir_unit = normalize(r.dir);
    const float a = 0.5f * (dir_unit.y + 1.0f);
    if (kind == 1) {  // skybox_color_day, 522-540: sun_dir is used as written, not normalised
        const vec3 sky_gradient = gl_mix(vec3(0.231f, 0.756f, 0.945f), vec3(1.0f), a);
        const float sun_dot = dot(vec3(-0.33f, 0.67f, -0.67f), dir_unit);
        return gl_mix(sky_gradient, vec3(1.0f), gl_smoothstep(size - bleed, size + bleed, sun_dot));
    }
    // skybox_color_night, 543-561
    const vec3 sky_gradient = gl_mix(vec3(0.0f), vec3(0.005f), a);
    const float moon_dot = dot(normalize(vec3(0.33f, 0.67f, -0.67f)), dir_unit);
    return gl_mix(sky_gradient, vec3(0.9f, 0.9f, 0.8f), gl_smoothstep(size - bleed, size + bleed, moon_dot));
}

// ----------------------------------------------------------------------------- lights, raytracer.comp:582-679
vec3 sample_light(Ctx& c, vec3 point, vec3 normal, vec3& L, float& pdf) {
    const RtoPushConstants* pc = c.pc;
    const RtoScene* sc = c.sc;
    if (pc->total_lights == 0 || pc->lights_strength_sum <= 0.0f) {
        pdf = 0.0f;
        return vec3(0.0f);
    }
    float rand_strength = random(c) * pc->lights_strength_sum;
    int low = 0, high = pc->total_lights - 1;
    while (low < high) {
        int mid = (low + high) / 2;
        if (rand_strength <= sc->lights[mid].accumulated_str) high = mid; else low = mid + 1;
    }
    const RtoLight& picked_light = sc->lights[low];
    vec3 color_rgb(picked_light.color_str[0], picked_light.color_str[1], picked_light.color_str[2]);
    float color_a = picked_light.color_str[3];
    vec3 pos_xyz(picked_light.pos_angle_aux[0], picked_light.pos_angle_aux[1], picked_light.pos_angle_aux[2]);

    switch (picked_light.type) {
        case 0: {  // AMBIENT
            L = random_vec_on_hemisphere(c, normal);
            pdf = 1.0f / 2.0f / PI;
            return color_rgb;
        }
        case 1: {  // SPHERE
            vec3 center = pos_xyz;
            float radius = picked_light.pos_angle_aux[3];
            vec3 center_to_point = point - center;
            vec3 sphere_point = random_vec_on_hemisphere(c, normalize(center_to_point)) * radius + center;
            vec3 point_to_spoint = sphere_point - point;
            float d_to_spoint = length(point_to_spoint);
            L = normalize(point_to_spoint);
            Ray s_ray = {point, L};
            Hit h;
            Interval i = {0.005f, PINF};
            if (hit_scene(c, s_ray, i, h)) {
                if (gl_abs(d_to_spoint - h.t) < 0.1f) {
                    float d2 = d_to_spoint * d_to_spoint;
                    pdf = 1.0f / 2.0f / PI;
                    return gl_clamp(color_rgb * color_a / d2, 0.0f, 1.0f);
                }
            }
            L = normal;
            pdf = 0.00001f;
            return vec3(0.0f);
        }
        case 3: {  // DIRECTIONAL
            vec3 light_dir = -pos_xyz;
            if (dot(normal, light_dir) > 0.0f) {
                Ray s_ray = {point, light_dir};
                if (!shadow_ray(c, s_ray)) {
                    L = light_dir;
                    pdf = 1.0f;
                    return color_rgb;
                }
            }
            L = normal;
            pdf = 0.00001f;
            return vec3(0.0f);
        }
        case 6: {  // TRIANGLE
            int tri_index = (int)picked_light.pos_angle_aux[0];
            const RtoTriangle& t = sc->triangles[tri_index];
            float e1 = std::sqrt(random(c)), e2 = random(c);
            vec3 tri_point = (1.0f - e1) * vec3(t.v0[0], t.v0[1], t.v0[2]) + e1 * (1.0f - e2) * vec3(t.v1[0], t.v1[1], t.v1[2]) +
                             e1 * e2 * vec3(t.v2[0], t.v2[1], t.v2[2]);
            vec3 point_to_spoint = tri_point - point;
            float d_to_spoint = length(point_to_spoint);
            L = normalize(point_to_spoint);
            Ray s_ray = {point, L};
            Hit h;
            Interval i = {0.005f, PINF};
            if (hit_scene(c, s_ray, i, h)) {
                if (gl_abs(d_to_spoint - h.t) < 0.1f) {
                    float d2 = d_to_spoint * d_to_spoint;
                    pdf = 1.0f;
                    return gl_clamp(color_rgb * color_a / d2, 0.0f, 1.0f);
                }
            }
            L = normal;
            pdf = 0.00001f;
            return vec3(0.0f);
        }
        default:  // POINT (2), CONE (4), AREA (5): the switch falls out with L, pdf unwritten
            break;
    }
    return vec3(1.0f);
}

// ----------------------------------------------------------------------------- BSDF, raytracer.comp:683-867
vec3 reflectance(const Ctx& c, float cos_theta, vec3 F0) {
    return gl_clamp(F0 + (1.0f - F0) * c.m_pow(1.0f - cos_theta, 5.0f), 0.0f, 1.0f);
}
float fresnel_dielectric(const Ctx& c, float cos_theta_i, float eta) {
    float r0 = (1.0f - eta) / (1.0f + eta);
    float F0 = r0 * r0;
    return F0 + (1.0f - F0) * c.m_pow(1.0f - cos_theta_i, 5.0f);
}
float ggx_distribution(float alpha, vec3 N, vec3 H) {
    float alpha_squared = alpha * alpha;
    float dot_product = dot(N, H);
    if (dot_product == 0.0f) dot_product = 0.000001f;
    float x = dot_product * dot_product * (alpha_squared - 1.0f) + 1.0f;
    return alpha_squared / (PI * x * x);
}
float G1_GGX(const Ctx& c, vec3 v, vec3 N, vec3 /*H*/, float alpha) {
    float voN = dot(v, N);
    if (voN == 0.0f) voN = 0.0000001f;
    float tan_theta_x = c.m_tan(c.m_acos(voN));
    float alpha_tan = alpha * tan_theta_x;
    return 2.0f / (1.0f + std::sqrt(1.0f + alpha_tan * alpha_tan));
}
vec3 sample_ggx(Ctx& c, float roughness, vec3 V, vec3 N) {
    float e1 = random(c), e2 = random(c);
    float alpha = roughness * roughness;
    float theta = c.m_acos(std::sqrt((1.0f - e1) / (1.0f + (alpha - 1.0f) * e1)));
    float phi = 2.0f * PI * e2;
    vec3 H_tan = sphere_to_cartesian(c, theta, phi);
    vec3 H = align_to_world(H_tan, N);
    if (dot(V, H) < 0.0f) H = -H;
    return normalize(H);
}
vec3 sample_r(Ctx& c, const Material& mat, vec3 V, vec3 N) {
    vec3 H = sample_ggx(c, mat.roughness, V, N);
    return reflect(-V, H);
}
vec3 sample_t(Ctx& c, const Material& mat, float eta, vec3 V, vec3 N) {
    vec3 H = sample_ggx(c, mat.roughness, V, N);
    float cos_theta = gl_min(1.0f, dot(V, H));
    float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
    bool cannot_refract = eta * sin_theta > 1.0f;
    float refl = fresnel_dielectric(c, cos_theta, eta);
    if (cannot_refract || refl > random(c)) return reflect(-V, H);
    return refract(-V, H, eta);
}
vec3 sample_mat(Ctx& c, const Material& mat, vec3 V, const Hit& rec) {
    if (mat.trs_weight < random(c)) return normalize(sample_r(c, mat, V, rec.normal));
    float eta_i = rec.front_face ? 1.0f : mat.ior;
    float eta_o = rec.front_face ? mat.ior : 1.0f;
    float eta = eta_i / eta_o;
    return normalize(sample_t(c, mat, eta, V, rec.normal));
}
vec3 eval_brdf(const Ctx& c, const Material& mat, vec3 L, vec3 V, const Hit& rec, float& pdf) {
    vec3 N = rec.normal;
    float NdotL = dot(L, N);
    float NdotV = dot(V, N);
    vec3 H = gl_sign(NdotV) * normalize(L + V);
    float VdotH = dot(V, H);
    float NdotH = dot(N, H);
    float ior_scale = gl_mix(0.0f, 2.0f, mat.specular_tint_a);
    float ri = rec.front_face ? (1.0f / (mat.ior * ior_scale)) : mat.ior * ior_scale;
    float dielectric_F0 = (1.0f - ri) / (1.0f + ri);
    vec3 dielectric_F0_vec = vec3(dielectric_F0 * dielectric_F0);
    vec3 F0 = gl_mix(dielectric_F0_vec, mat.albedo, mat.metallic);
    vec3 f_diffuse = mat.albedo / PI;
    float alpha = mat.roughness * mat.roughness;
    float D = ggx_distribution(alpha, N, H);
    float G = G1_GGX(c, L, N, H, alpha) * G1_GGX(c, V, N, H, alpha);
    vec3 F = reflectance(c, VdotH, F0);
    float ks = gl_max(gl_max(F.x, F.y), F.z);
    float kd = (1.0f - ks) * (1.0f - mat.metallic);
    float jacobian = 1.0f / gl_max(0.00001f, (4.0f * NdotV * NdotL));
    vec3 f_specular = mat.specular_tint * D * G * F * jacobian;
    float pdf_specular = gl_clamp(D * NdotH * jacobian, 0.0f, 1.0f);
    float pdf_diffuse = gl_clamp(NdotL / PI, 0.0f, 1.0f);
    pdf = kd * pdf_diffuse + ks * pdf_specular;
    return kd * f_diffuse + f_specular;
}
vec3 eval_btdf(const Ctx& c, const Material& mat, vec3 L, vec3 V, const Hit& rec, float& pdf) {
    L = normalize(L);
    V = normalize(V);
    vec3 N = normalize(rec.normal);
    float eta_i = rec.front_face ? 1.0f : mat.ior;
    float eta_o = rec.front_face ? mat.ior : 1.0f;
    float eta = eta_i / eta_o;
    vec3 H = -normalize(L + eta * V);
    float VoH = dot(V, H);
    float LoH = dot(L, H);
    float NoH = dot(N, H);
    float VoN = dot(V, N);
    float LoN = dot(L, N);
    float alpha = mat.roughness * mat.roughness;
    float D = ggx_distribution(alpha, N, H);
    float G = G1_GGX(c, L, N, H, alpha) * G1_GGX(c, V, N, H, alpha);
    float F = fresnel_dielectric(c, gl_abs(VoH), eta);
    float x = gl_abs(VoH) / gl_max(0.00001f, gl_abs(VoN) * gl_abs(LoN));
    float denom = (eta_i * VoH + eta_o * LoH);
    float jacobian = (eta_o * eta_o * gl_abs(LoH)) / gl_max(0.00001f, (denom * denom));
    pdf = D * gl_abs(NoH) * jacobian;
    return mat.subsurface * x * jacobian * D * G * (1.0f - F);
}
vec3 eval_mat(const Ctx& c, const Material& mat, vec3 L, vec3 V, const Hit& rec, float& pdf) {
    L = normalize(L);
    vec3 N = normalize(rec.normal);
    if (dot(L, N) >= 0.0f) return eval_brdf(c, mat, L, V, rec, pdf);
    return eval_btdf(c, mat, L, V, rec, pdf);
}

// raytracer.comp:870-881
vec3 direct_light(Ctx& c, const Hit& rec, const Ray& ray, float& pdf) {
    vec3 L_dir(0.0f);
    float light_pdf = 0.0f, mat_pdf;
    vec3 L_emission = sample_light(c, rec.p, rec.normal, L_dir, light_pdf);
    float cos_theta = gl_max(0.0f, dot(rec.normal, L_dir));
    vec3 fr = eval_mat(c, load_material(c.sc, rec.mat), L_dir, -ray.dir, rec, mat_pdf);
    pdf = power_heuristics(light_pdf, mat_pdf);
    return gl_clamp(L_emission * fr * cos_theta / pdf, 0.0f, 1.0f);
}

// raytracer.comp:885-930 (returns the rgb of the vec4; alpha is always 1)
vec3 ray_color(Ctx& c, Ray r, int max_bounces) {
    vec3 color(0.0f);
    vec3 attenuation(1.0f);
    Hit h;
    h.mat = 0; h.t = 0; h.front_face = false;
    for (int bounce = 0; bounce <= max_bounces; bounce++) {
        Interval iv = {0.005f, PINF};
        bool hit = hit_scene(c, r, iv, h);
        if (!c.primary_taken) {  // bookkeeping: id of sample 0's primary hit
            c.primary_id = c.last_id;
            c.primary_taken = true;
        }
        if (hit) {
            Material mat = load_material(c.sc, h.mat);
            if (mat.albedo_a < 1.0f && mat.albedo_a < random(c)) {
                r.orig = h.p;
                continue;
            }
            if (mat.emission_a > 0.0f) {
                color = color + attenuation * mat.emission_color;
                break;
            }
            if (bounce == 0) {
                float light_pdf;
                color = color + direct_light(c, h, r, light_pdf) * attenuation;
            }
            vec3 bounce_dir = sample_mat(c, mat, -r.dir, h);
            float mat_pdf;
            vec3 fr = eval_mat(c, mat, bounce_dir, -r.dir, h, mat_pdf);
            float cos_theta = gl_abs(dot(h.normal, bounce_dir));
            attenuation = attenuation * gl_max(vec3(0.0f), fr * cos_theta / gl_max(0.00001f, mat_pdf));
            r.orig = h.p;
            r.dir = bounce_dir;
        } else {
            color = color + attenuation * skybox_color(c.skybox, r);
            return gl_clamp(color, 0.0f, 1.0f);
        }
    }
    return gl_clamp(color, 0.0f, 1.0f);
}

// raytracer.comp:933-952
Ray get_ray(Ctx& c) {
    float sx = random(c) - 0.5f;  // sample_square(): x first, then y (245-247)
    float sy = random(c) - 0.5f;
    float ox = (float)c.px + sx;
    float oy = (float)c.py + sy;
    float ndcX = 2.0f * ox / (float)c.width - 1.0f;
    float ndcY = 1.0f - 2.0f * oy / (float)c.height;
    float aspectRatio = (float)c.width / (float)c.height;  // 187
    vec3 rayDirCameraSpace = normalize(vec3(ndcX * aspectRatio * c.cam->tanHalfFOV, ndcY * c.cam->tanHalfFOV, -1.0f));
    // viewInv * vec4(dir, 0.0): sum of column * component, left to right
    const float* m = c.cam->viewInv;
    vec3 d = rayDirCameraSpace;
    vec3 w(m[0] * d.x + m[4] * d.y + m[8] * d.z + m[12] * 0.0f,
           m[1] * d.x + m[5] * d.y + m[9] * d.z + m[13] * 0.0f,
           m[2] * d.x + m[6] * d.y + m[10] * d.z + m[14] * 0.0f);
    Ray ray;
    ray.orig = vec3(c.cam->position[0], c.cam->position[1], c.cam->position[2]);
    ray.dir = normalize(w);
    return ray;
}

uint32_t unorm8(float x) {  // rgba8 store: clamp, scale, round to nearest even; NaN -> 0
    float v = gl_clamp(x, 0.0f, 1.0f);
    return (uint32_t)std::nearbyint(v * 255.0f);
}

// raytracer.comp:956-996 for one pixel
void shader_main(Ctx& c, const RtoParams* p, const RtoOutputs* out) {
    const RtoPushConstants* pc = c.pc;
    uint32_t time_u = (pc->time >= 0.0f && pc->time < 4294967296.0f) ? (uint32_t)pc->time : 0u;
    c.seed = hash(time_u * 1920u) ^ hash(pc->frameCount) ^ hash((uint32_t)(c.px + c.py * 1920));
    c.path_hash = 0;
    c.primary_taken = false;
    c.primary_id = -1;

    vec3 color(0.0f);
    for (int i = 0; i < p->spp; i++) {
        Ray ray = get_ray(c);
        color = color + ray_color(c, ray, p->max_bounces);
    }
    color = color / (float)p->spp;
    color = c.m_pow(color, 1.0f / 2.2f);

    size_t idx = (size_t)c.py * (size_t)c.width + (size_t)c.px;
    float* acc = out->accum + 4 * idx;
    if (pc->reset_frame_accumulation) {
        acc[0] = acc[1] = acc[2] = acc[3] = 0.0f;
        out->counts[idx] = 0;
    }
    acc[0] += color.x; acc[1] += color.y; acc[2] += color.z; acc[3] += 1.0f;
    out->counts[idx] += 1;
    int cnt = out->counts[idx];
    float d = (float)(cnt > 1 ? cnt : 1);
    if (out->bgra) {  // imageStore(final_color.zyxw) into rgba8: bytes B, G, R, A
        uint8_t* o = out->bgra + 4 * idx;
        o[0] = (uint8_t)unorm8(acc[2] / d);
        o[1] = (uint8_t)unorm8(acc[1] / d);
        o[2] = (uint8_t)unorm8(acc[0] / d);
        o[3] = (uint8_t)unorm8(acc[3] / d);
    }
    if (out->primary_ids) out->primary_ids[idx] = c.primary_id;
    if (out->path_hash) out->path_hash[idx] = c.path_hash;
}

}  // namespace

extern "C" {

int rto_render(const RtoScene* scene, const RtoCamera* camera, const RtoPushConstants* pc, const RtoParams* params,
               const RtoOutputs* out) {
    if (!scene || !camera || !pc || !params || !out || !out->accum || !out->counts) return -1;
    if (params->width <= 0 || params->height <= 0 || params->spp <= 0 || params->max_bounces < 0) return -1;
    int nthreads = params->threads > 0 ? params->threads : 1;
    const int H = params->height, W = params->width;
    // rows this call renders: the strip partition of the device path, or [row_begin, row_end)
    std::vector<int> rows;
    int rb = params->row_begin, re = params->row_end > 0 ? params->row_end : H;
    for (int y = rb; y < re && y < H; ++y) {
        if (params->strip_rows > 0 && ((y / params->strip_rows) % params->strip_count) != params->strip_index) continue;
        rows.push_back(y);
    }
    std::vector<uint64_t> ray_counts((size_t)nthreads, 0);
    auto worker = [&](int tid) {
        Ctx c;
        c.sc = scene; c.cam = camera; c.pc = pc;
        c.math_mode = params->math_mode;
        c.skybox = params->skybox;
        c.width = W; c.height = H;
        c.rays = 0; c.last_id = -1; c.path_hash = 0; c.primary_id = -1; c.primary_taken = false; c.seed = 0;
        for (size_t k = (size_t)tid; k < rows.size(); k += (size_t)nthreads) {
            c.py = rows[k];
            for (int x = 0; x < W; ++x) {
                c.px = x;
                shader_main(c, params, out);
            }
        }
        ray_counts[(size_t)tid] = c.rays;
    };
    if (nthreads == 1) {
        worker(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(worker, t);
        for (auto& t : th) t.join();
    }
    uint64_t total = 0;
    for (uint64_t r : ray_counts) total += r;
    if (out->rays) *out->rays = total;
    return 0;
}

// Scalar entry points for unit tests of the arithmetic.
float rto_math(int fn, int math_mode, float x, float y) {
    Ctx c;
    c.math_mode = math_mode;
    switch (fn) {
        case 0: return c.m_sin(x);
        case 1: return c.m_cos(x);
        case 2: return c.m_tan(x);
        case 3: return c.m_acos(x);
        case 4: return c.m_pow(x, y);
        case 5: return rm_log2(x);
        case 6: return rm_exp2(x);
        case 7: return rm_asin(x);
        default: return 0.0f;
    }
}
uint32_t rto_hash(uint32_t x) { return hash(x); }
uint32_t rto_xorshift(uint32_t state) { return xorshift(state); }
float rto_random_from(uint32_t state) {
    Ctx c;
    c.seed = state;
    return random(c);
}
uint32_t rto_pixel_seed(float time, uint32_t frame_count, int x, int y) {
    uint32_t time_u = (time >= 0.0f && time < 4294967296.0f) ? (uint32_t)time : 0u;
    return hash(time_u * 1920u) ^ hash(frame_count) ^ hash((uint32_t)(x + y * 1920));
}
// One nearest-hit query: returns the winner's primitive id (-1 = miss) and its t.
int rto_hit_scene(const RtoScene* scene, const RtoPushConstants* pc, const float orig[3], const float dir[3], float* t_out) {
    Ctx c;
    c.sc = scene; c.pc = pc; c.cam = nullptr; c.math_mode = 0; c.rays = 0; c.path_hash = 0; c.last_id = -1;
    Ray r = {vec3(orig[0], orig[1], orig[2]), vec3(dir[0], dir[1], dir[2])};
    Hit h;
    h.t = 0; h.mat = 0; h.front_face = false;
    Interval iv = {0.005f, PINF};
    bool hit = hit_scene(c, r, iv, h);
    if (t_out) *t_out = hit ? h.t : PINF;
    return c.last_id;
}

}  // extern "C"
