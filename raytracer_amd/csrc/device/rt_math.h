// rt_math.h — device-side scalar and vector arithmetic for the path-tracing kernels.
//
// Arithmetic contract: every GLSL operator of shaders/raytracer.comp is ONE IEEE-754
// binary32 operation, evaluated in source order, with no fused multiply-add (this file is
// compiled with -ffp-contract=off) and with correctly rounded division and square root.
// GLSL built-ins whose precision Vulkan leaves to the implementation (sin, cos, tan, acos,
// pow/exp2/log2) are fixed here to explicit polynomial evaluations in binary32 so that a
// host restatement using the same formulas reproduces the device bit for bit.
//   normalize(v) = v * (1 / sqrt(dot(v,v)))      dot = (x*x' + y*y') + z*z'
//   min/max/clamp return the non-NaN operand (as shipping GPUs' min/max instructions do)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtd {

#define RT_DEV __device__ __forceinline__

static constexpr float kPi = 3.14159265359f;  // raytracer.comp:4
static constexpr float kInf = __builtin_huge_valf();

RT_DEV uint32_t f2u(float x) { return __float_as_uint(x); }
RT_DEV float u2f(uint32_t x) { return __uint_as_float(x); }
RT_DEV bool is_nan(float x) { return x != x; }

RT_DEV float g_max(float x, float y) { return is_nan(x) ? y : (is_nan(y) ? x : ((x < y) ? y : x)); }
RT_DEV float g_min(float x, float y) { return is_nan(x) ? y : (is_nan(y) ? x : ((y < x) ? y : x)); }
RT_DEV float g_clamp(float x, float lo, float hi) { return g_min(g_max(x, lo), hi); }
RT_DEV float g_abs(float x) { return u2f(f2u(x) & 0x7fffffffu); }
RT_DEV float g_sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
RT_DEV float g_mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }

// ---- correctly rounded 1/x and sqrt(x) in short form -------------------------------------
// The compiler's expansions of IEEE division and square root (11 and 14 instructions, about 47
// and 55 issue cycles per wave) spend half of that on range scaling and special-case fix-up.
// Inside the ranges below the hardware estimate plus one Newton round in fused arithmetic gives
// the correctly rounded result already: compared with the expansions over ALL 2^32 bit patterns
// on the device (rt_debug_math sweeps, tools/ubench/exact_math.hip), no input inside the ranges
// differs.  Outside (zero, subnormal, huge, infinite, NaN) the expansions themselves run.
RT_DEV float ieee_div(float a, float b) { return a / b; }
RT_DEV float ieee_sqrt(float x) { return __builtin_sqrtf(x); }  // IEEE under -fhip-fp32-correctly-rounded-divide-sqrt
// biased exponent 1..252: 2^-126 <= |x| < 2^126 (the reciprocal of anything larger is subnormal)
RT_DEV bool rcp_short_ok(float x) { return ((f2u(x) & 0x7f800000u) - 0x00800000u) < 0x7e000000u; }
// positive, biased exponent 26..254: 2^-101 <= x < 2^128
RT_DEV bool sqrt_short_ok(float x) { return (f2u(x) - 0x0d000000u) < 0x72800000u; }
RT_DEV float rcp_short(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
RT_DEV float sqrt_short(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    float g = x * r;
    const float h = 0.5f * r;
    const float e = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, e, g);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
// The choice is made per wave (one scalar branch; a per-lane branch costs six scalar instructions of
// exec-mask bookkeeping): the expansion is right for every input, so a wave with one lane out of
// range runs it for all its lanes.
RT_DEV bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }
RT_DEV float g_rcp(float x) {  // 1.0f / x
    if (__builtin_expect(wave_all(rcp_short_ok(x)), 1)) return rcp_short(x);
    return ieee_div(1.0f, x);
}
RT_DEV float g_sqrt(float x) {
    if (__builtin_expect(wave_all(sqrt_short_ok(x)), 1)) return sqrt_short(x);
    return ieee_sqrt(x);
}
// 1.0f / sqrt(x), two roundings: the root of an in-range x lies in [2^-51, 2^64), inside the reciprocal's range
RT_DEV float g_rcp_sqrt(float x) {
    if (__builtin_expect(wave_all(sqrt_short_ok(x)), 1)) return rcp_short(sqrt_short(x));
    return ieee_div(1.0f, ieee_sqrt(x));
}

// ---- trigonometry: Cody-Waite reduction by pi/4 octants + minimax polynomials ----------
static constexpr float kFOPI = 1.27323954473516f;
static constexpr float kDP1 = 0.78515625f;
static constexpr float kDP2 = 2.4187564849853515625e-4f;
static constexpr float kDP3 = 3.77489497744594108e-8f;
static constexpr float kTrigMax = 1.0e9f;  // beyond this the octant no longer fits an int

RT_DEV float poly_sin(float x, float z) {
    return ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
}
RT_DEV float poly_cos(float z) {
    return ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
}

RT_DEV float g_sin(float xx) {
    float x = g_abs(xx);
    if (!(x <= kTrigMax)) return xx - xx;  // inf, NaN and out-of-range -> NaN
    bool neg = xx < 0.0f;
    int j = (int)(kFOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { neg = !neg; j -= 4; }
    x = ((x - y * kDP1) - y * kDP2) - y * kDP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? poly_cos(z) : poly_sin(x, z);
    return neg ? -r : r;
}

RT_DEV float g_cos(float xx) {
    float x = g_abs(xx);
    if (!(x <= kTrigMax)) return xx - xx;
    bool neg = false;
    int j = (int)(kFOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { j -= 4; neg = !neg; }
    if (j > 1) neg = !neg;
    x = ((x - y * kDP1) - y * kDP2) - y * kDP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? poly_sin(x, z) : poly_cos(z);
    return neg ? -r : r;
}

// sin and cos of one angle: the two functions above share the octant reduction and both
// polynomials; evaluated together they cost little more than one of them.  Bit for bit
// the same values as g_sin(xx) and g_cos(xx).
RT_DEV void g_sincos(float xx, float& s, float& c) {
    float x = g_abs(xx);
    if (!(x <= kTrigMax)) {
        s = xx - xx;
        c = s;
        return;
    }
    int j = (int)(kFOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    bool sneg = xx < 0.0f, cneg = false;
    if (j > 3) { j -= 4; sneg = !sneg; cneg = !cneg; }
    if (j > 1) cneg = !cneg;
    x = ((x - y * kDP1) - y * kDP2) - y * kDP3;
    const float z = x * x;
    const float ps = poly_sin(x, z), pc = poly_cos(z);
    const bool swap = (j == 1 || j == 2);
    const float rs = swap ? pc : ps, rc = swap ? ps : pc;
    s = sneg ? -rs : rs;
    c = cneg ? -rc : rc;
}

RT_DEV float g_tan(float xx) {
    float x = g_abs(xx);
    if (!(x <= kTrigMax)) return xx - xx;
    bool neg = xx < 0.0f;
    int j = (int)(kFOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    float z = ((x - y * kDP1) - y * kDP2) - y * kDP3;
    float zz = z * z;
    float r;
    if (x > 1.0e-4f) {
        r = (((((9.38540185543e-3f * zz + 3.11992232697e-3f) * zz + 2.44301354525e-2f) * zz + 5.34112807005e-2f) * zz +
              1.33387994085e-1f) * zz + 3.33331568548e-1f) * zz * z + z;
    } else {
        r = z;
    }
    if (j & 2) r = -g_rcp(r);
    return neg ? -r : r;
}

// asin polynomial on [0, 0.5]: x + x * z * P(z), z = x * x
RT_DEV float poly_asin(float x, float z) {
    return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z +
            1.6666752422e-1f) * z * x + x;
}

// asin on [-1,1] (outside: NaN): |x| <= 0.5: the polynomial; above: pi/2 - 2 asin(sqrt((1 - |x|) / 2)).
// Both ranges run the one polynomial on per-lane selected arguments (z is (1 - |x|) / 2 itself
// in the upper range, not the square of its root).
RT_DEV float g_asin(float xx) {
    float a = g_abs(xx);
    if (!(a <= 1.0f)) return u2f(0x7fc00000u);
    const bool flag = a > 0.5f;
    const float zf = 0.5f * (1.0f - a);
    const float x = flag ? g_sqrt(zf) : a;
    const float z = flag ? zf : a * a;
    float r = poly_asin(x, z);
    if (flag) {
        r = r + r;
        r = 1.5707963267948966192f - r;
    }
    return (xx < 0.0f) ? -r : r;
}

// acos(x) = pi - 2 asin(sqrt((1 + x) / 2)) below -0.5, 2 asin(sqrt((1 - x) / 2)) above 0.5,
// pi/2 - asin(x) between.  The inner asin arguments never exceed 0.5, so every range is the
// one polynomial on a per-lane selected argument w (z = w * w in all three ranges).
RT_DEV float g_acos(float x) {
    const float a = g_abs(x);
    if (!(a <= 1.0f)) return u2f(0x7fc00000u);
    const bool tail = a > 0.5f;
    // 0.5 * (1 + x) for x < -0.5 and 0.5 * (1 - x) for x > 0.5 are the same operation on |x|
    const float w = tail ? g_sqrt(0.5f * (1.0f - a)) : a;
    const float r = poly_asin(w, w * w);
    if (tail) return (x < 0.0f) ? (kPi - 2.0f * r) : (2.0f * r);
    return 1.5707963267948966192f - ((x < 0.0f) ? -r : r);
}

// ---- pow(x, y) = exp2(y * log2(x)), the GLSL definition ---------------------------------
RT_DEV float g_log2(float x) {
    if (is_nan(x) || x < 0.0f) return u2f(0x7fc00000u);
    if (x == 0.0f) return -kInf;
    if (x == kInf) return kInf;
    uint32_t ix = f2u(x);
    int e = 0;
    if ((ix & 0x7f800000u) == 0u) {  // subnormal: renormalise by 2^23
        x = x * 8388608.0f;
        ix = f2u(x);
        e = -23;
    }
    e += (int)((ix >> 23) & 0xffu) - 126;
    float m = u2f((ix & 0x007fffffu) | 0x3f000000u);  // mantissa in [0.5, 1)
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m +
                   1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m +
               3.3333331174e-1f) * m * z;
    y = y - 0.5f * z;  // ln(1+m) - m
    const float kLog2EA = 0.44269504088896340735992f;  // log2(e) - 1
    float r = y * kLog2EA;
    r = r + m * kLog2EA;
    r = r + y;
    r = r + m;
    r = r + (float)e;
    return r;
}

RT_DEV float g_exp2(float x) {
    if (is_nan(x)) return x;
    if (x >= 128.0f) return kInf;
    if (x < -126.0f) return 0.0f;
    float fl = __builtin_floorf(x);
    int i0 = (int)fl;
    float f = x - fl;
    if (f > 0.5f) {
        i0 += 1;
        f = f - 1.0f;
    }
    float p = (((((1.535336188319500e-4f * f + 1.339887440266574e-3f) * f + 9.618437357674640e-3f) * f +
                 5.550332471162809e-2f) * f + 2.402264791363012e-1f) * f + 6.931472028550421e-1f) * f + 1.0f;
    // i0 in [-126, 128]: scale in two exact steps so that 2^128 is never formed
    int h = i0 / 2;
    float s1 = u2f((uint32_t)(h + 127) << 23);
    float s2 = u2f((uint32_t)(i0 - h + 127) << 23);
    return (p * s1) * s2;
}

RT_DEV float g_pow(float x, float y) { return g_exp2(y * g_log2(x)); }

// ---- vec3 ------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
RT_DEV V3 v3(float x, float y, float z) { return {x, y, z}; }
RT_DEV V3 v3(float s) { return {s, s, s}; }
RT_DEV V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
RT_DEV V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_DEV V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
RT_DEV V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
RT_DEV V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
RT_DEV V3 operator+(V3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
RT_DEV V3 operator-(float s, V3 a) { return {s - a.x, s - a.y, s - a.z}; }
RT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
RT_DEV float length(V3 a) { return g_sqrt(dot(a, a)); }
RT_DEV V3 normalize(V3 a) { return a * g_rcp_sqrt(dot(a, a)); }
RT_DEV V3 g_max(V3 a, V3 b) { return {g_max(a.x, b.x), g_max(a.y, b.y), g_max(a.z, b.z)}; }
RT_DEV V3 g_clamp(V3 a, float lo, float hi) { return {g_clamp(a.x, lo, hi), g_clamp(a.y, lo, hi), g_clamp(a.z, lo, hi)}; }
RT_DEV V3 g_mix(V3 x, V3 y, float a) { return {g_mix(x.x, y.x, a), g_mix(x.y, y.y, a), g_mix(x.z, y.z, a)}; }
RT_DEV V3 g_pow(V3 a, float e) { return {g_pow(a.x, e), g_pow(a.y, e), g_pow(a.z, e)}; }
// GLSL reflect / refract
RT_DEV V3 g_reflect(V3 I, V3 N) { return I - 2.0f * dot(N, I) * N; }
RT_DEV V3 g_refract(V3 I, V3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return v3(0.0f);
    return eta * I - (eta * d + g_sqrt(k)) * N;
}

// ---- RNG (raytracer.comp:193-211) ----------------------------------------------------------
RT_DEV uint32_t rng_hash(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
RT_DEV uint32_t rng_xorshift(uint32_t& s) {
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s;
}
// float(state) / 4294967295.0: the divisor literal is a binary32 constant and rounds to
// 2^32, so the result lies in [0, 1] and CAN equal 1.0.
RT_DEV float rng_random(uint32_t& s) { return (float)rng_xorshift(s) / 4294967296.0f; }

}  // namespace rtd
