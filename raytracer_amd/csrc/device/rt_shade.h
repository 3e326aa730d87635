// rt_shade.h — material evaluation / sampling and hit-record helpers of the path tracer,
// following shaders/raytracer.comp function by function (line ranges cited per function).
#pragma once
#include "rt_math.h"

namespace rtd {

// What a ray that leaves the scene sees (raytracer.comp:520-577): `kind` selects among the shader's
// skybox functions (0 grey, the one the reference compiles in; 1 day; 2 night; 3 white; 4 black).
// smoothstep(e0, e1, x) = t*t*(3 - 2t), t = clamp((x - e0) / (e1 - e0), 0, 1); mix as g_mix.
RT_DEV float g_smoothstep(float e0, float e1, float x) {
    const float t = g_clamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
RT_DEV V3 skybox_color(int kind, V3 dir) {
    if (kind == 0) return v3(0.3f);
    if (kind == 3) return v3(1.0f);
    if (kind == 4) return v3(0.0f);
    const float size = 0.999f, bleed = 0.0003f;
    const V3 d = normalize(dir);
    const float a = 0.5f * (d.y + 1.0f);
    if (kind == 1) {  // skybox_color_day, raytracer.comp:522-540 (its sun direction is used unnormalised)
        const V3 grad = g_mix(v3(0.231f, 0.756f, 0.945f), v3(1.0f), a);
        const float mask = g_smoothstep(size - bleed, size + bleed, dot(v3(-0.33f, 0.67f, -0.67f), d));
        return g_mix(grad, v3(1.0f), mask);
    }
    // skybox_color_night, raytracer.comp:543-561
    const V3 grad = g_mix(v3(0.0f), v3(0.005f), a);
    // normalize(vec3(0.33, 0.67, -0.67)) evaluated in binary32 under the arithmetic contract (dot, sqrt,
    // reciprocal, three products), written out so that it is a literal and not run-time work
    const V3 kMoonDir = v3(0x1.50cb2cp-2f, 0x1.55e586p-1f, -0x1.55e586p-1f);
    const float mask = g_smoothstep(size - bleed, size + bleed, dot(kMoonDir, d));
    return g_mix(grad, v3(0.9f, 0.9f, 0.8f), mask);
}

struct Material {  // raytracer.comp:54-82
    V3 albedo;        float albedo_a;
    V3 subsurface;    float subsurface_a;
    V3 specular_tint; float specular_tint_a;
    V3 emission;      float emission_a;
    float roughness, metallic, ior, trs_weight;
};

struct Hit {  // raytracer.comp:108-114
    V3 p;
    V3 normal;
    int mat;
    float t;
    bool front_face;
};

// Row `index` of a table of `row_bytes`-byte rows: the offset is formed in 32 bits, so that a load from a uniform base
// compiles to base-in-scalar-registers + one 32-bit lane offset instead of 64-bit vector address arithmetic (every table
// the kernels index - records, nodes, materials, lights, triangles - is below 4 GB: the host refuses larger scenes).
RT_DEV const float4* table_row(const float4* __restrict__ base, int index, uint32_t row_bytes) {
    return (const float4*)((const char*)base + (uint32_t)index * row_bytes);
}

RT_DEV Material load_material(const float4* __restrict__ mats, int idx) {
    const float4* m = table_row(mats, idx, 80u);
    float4 a = m[0], s = m[1], sp = m[2], e = m[3], g = m[4];
    Material r;
    r.albedo = v3(a.x, a.y, a.z);         r.albedo_a = a.w;
    r.subsurface = v3(s.x, s.y, s.z);     r.subsurface_a = s.w;
    r.specular_tint = v3(sp.x, sp.y, sp.z); r.specular_tint_a = sp.w;
    r.emission = v3(e.x, e.y, e.z);       r.emission_a = e.w;
    r.roughness = g.x; r.metallic = g.y; r.ior = g.z; r.trs_weight = g.w;
    return r;
}

// raytracer.comp:225-243
RT_DEV V3 random_unit_vec(uint32_t& seed) {
    float phi = (2.0f * kPi) * rng_random(seed);
    float theta = g_acos(2.0f * rng_random(seed) - 1.0f);
    float sin_theta, cos_theta, sin_phi, cos_phi;
    g_sincos(theta, sin_theta, cos_theta);
    g_sincos(phi, sin_phi, cos_phi);
    return v3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
}
RT_DEV V3 random_vec_on_hemisphere(V3 normal, uint32_t& seed) {
    V3 r = random_unit_vec(seed);
    return (dot(r, normal) > 0.0f) ? r : -r;
}

// raytracer.comp:250-254
RT_DEV float power_heuristics(float a, float b) {
    float a2 = a * a;
    float b2 = b * b;
    return a2 / g_max(a2 + b2, 0.000001f);
}

// raytracer.comp:260-267
RT_DEV V3 sphere_to_cartesian(float theta, float phi) {
    float sin_t, cos_t, sin_p, cos_p;
    g_sincos(theta, sin_t, cos_t);
    g_sincos(phi, sin_p, cos_p);
    return v3(sin_t * cos_p, sin_t * sin_p, cos_t);
}

// raytracer.comp:279-292
RT_DEV V3 align_to_world(V3 X, V3 N) {
    V3 up = (g_abs(N.z) > 0.9999999f) ? v3(1.0f, 0.0f, 0.0f) : v3(0.0f, 0.0f, 1.0f);
    V3 T = normalize(cross(up, N));
    V3 B = cross(N, T);
    return T * X.x + B * X.y + N * X.z;
}

// raytracer.comp:683-695
RT_DEV float fresnel_dielectric(float cos_theta_i, float eta) {
    float r0 = (1.0f - eta) / (1.0f + eta);
    float F0 = r0 * r0;
    return F0 + (1.0f - F0) * g_pow(1.0f - cos_theta_i, 5.0f);
}

// raytracer.comp:699-705
RT_DEV float ggx_distribution(float alpha, V3 N, V3 H) {
    float alpha_squared = alpha * alpha;
    float d = dot(N, H);
    if (d == 0.0f) d = 0.000001f;
    float x = d * d * (alpha_squared - 1.0f) + 1.0f;
    return alpha_squared / (kPi * x * x);
}

// raytracer.comp:716-727 (H is unused by the reference too)
RT_DEV float G1_GGX(V3 v, V3 N, float alpha) {
    float voN = dot(v, N);
    if (voN == 0.0f) voN = 0.0000001f;
    float tan_theta_x = g_tan(g_acos(voN));
    float alpha_tan = alpha * tan_theta_x;
    return 2.0f / (1.0f + g_sqrt(1.0f + alpha_tan * alpha_tan));
}

// raytracer.comp:736-748
RT_DEV V3 sample_ggx(float roughness, V3 V, V3 N, uint32_t& seed) {
    float e1 = rng_random(seed);
    float e2 = rng_random(seed);
    float alpha = roughness * roughness;
    float theta = g_acos(g_sqrt((1.0f - e1) / (1.0f + (alpha - 1.0f) * e1)));
    float phi = (2.0f * kPi) * e2;
    V3 H_tan = sphere_to_cartesian(theta, phi);
    V3 H = align_to_world(H_tan, N);
    if (dot(V, H) < 0.0f) H = -H;
    return normalize(H);
}

// raytracer.comp:751-785
// Both branches (sample_r, sample_t) start with the same sample_ggx call right after the
// branch draw, so the half vector is sampled once, for all lanes, before the branch.
RT_DEV V3 sample_mat(const Material& mat, V3 V, const Hit& rec, uint32_t& seed) {
    const bool reflect_only = mat.trs_weight < rng_random(seed);
    const V3 H = sample_ggx(mat.roughness, V, rec.normal, seed);
    if (reflect_only) return normalize(g_reflect(-V, H));  // sample_r
    float eta_i = rec.front_face ? 1.0f : mat.ior;
    float eta_o = rec.front_face ? mat.ior : 1.0f;
    float eta = eta_i / eta_o;
    // sample_t
    float cos_theta = g_min(1.0f, dot(V, H));
    float sin_theta = g_sqrt(1.0f - cos_theta * cos_theta);
    bool cannot_refract = eta * sin_theta > 1.0f;
    float refl = fresnel_dielectric(cos_theta, eta);
    V3 d;
    if (cannot_refract || refl > rng_random(seed)) {
        d = g_reflect(-V, H);
    } else {
        d = g_refract(-V, H, eta);
    }
    return normalize(d);
}

// eval_mat (raytracer.comp:859-867) with eval_brdf (789-824) and eval_btdf (826-855) folded
// into one body.  A lane takes exactly one of the two branches, but a wave usually holds
// lanes of both kinds; the expensive pieces the branches have in common — the GGX
// distribution, the two G1 terms (acos, tan, sqrt) and the Fresnel power — are therefore
// evaluated ONCE for the whole wave on per-lane selected inputs.  Every lane still
// performs, operation for operation, what its own branch of the shader performs.
RT_DEV V3 eval_mat(const Material& mat, V3 L, V3 V, const Hit& rec, float& pdf) {
    L = normalize(L);
    const V3 Nn = normalize(rec.normal);
    const bool reflect_side = dot(L, Nn) >= 0.0f;
    const float alpha = mat.roughness * mat.roughness;

    // per-lane inputs of the shared part
    V3 Ls, Vs, Ns, Hs;
    float cos_f;  // argument of the Fresnel power
    // branch-private values carried to the epilogue
    float NdotL = 0.0f, NdotV = 0.0f, NdotH = 0.0f;                       // BRDF
    float VoH = 0.0f, LoH = 0.0f, NoH = 0.0f, VoN = 0.0f, LoN = 0.0f;     // BTDF
    float eta = 1.0f, eta_i = 1.0f, eta_o = 1.0f;
    if (reflect_side) {  // eval_brdf: N = rec.normal as stored, V as passed
        const V3 N = rec.normal;
        NdotL = dot(L, N);
        NdotV = dot(V, N);
        const V3 H = g_sign(NdotV) * normalize(L + V);
        const float VdotH = dot(V, H);
        NdotH = dot(N, H);
        Ls = L; Vs = V; Ns = N; Hs = H;
        cos_f = VdotH;
    } else {  // eval_btdf: everything re-normalised
        const V3 L2 = normalize(L);
        const V3 V2 = normalize(V);
        const V3 N2 = normalize(rec.normal);
        eta_i = rec.front_face ? 1.0f : mat.ior;
        eta_o = rec.front_face ? mat.ior : 1.0f;
        eta = eta_i / eta_o;
        const V3 H = -normalize(L2 + eta * V2);
        VoH = dot(V2, H);
        LoH = dot(L2, H);
        NoH = dot(N2, H);
        VoN = dot(V2, N2);
        LoN = dot(L2, N2);
        Ls = L2; Vs = V2; Ns = N2; Hs = H;
        cos_f = g_abs(VoH);
    }

    // shared: D, G, (1 - cos)^5
    const float D = ggx_distribution(alpha, Ns, Hs);
    const float G = G1_GGX(Ls, Ns, alpha) * G1_GGX(Vs, Ns, alpha);
    const float pw = g_pow(1.0f - cos_f, 5.0f);

    if (reflect_side) {
        float ior_scale = g_mix(0.0f, 2.0f, mat.specular_tint_a);
        float ri = rec.front_face ? g_rcp(mat.ior * ior_scale) : mat.ior * ior_scale;
        float dielectric_F0 = (1.0f - ri) / (1.0f + ri);
        V3 dielectric_F0_vec = v3(dielectric_F0 * dielectric_F0);
        V3 F0 = g_mix(dielectric_F0_vec, mat.albedo, mat.metallic);
        V3 f_diffuse = mat.albedo / kPi;
        V3 F = g_clamp(F0 + (1.0f - F0) * pw, 0.0f, 1.0f);  // reflectance(VdotH, F0), raytracer.comp:686-688
        float ks = g_max(g_max(F.x, F.y), F.z);
        float kd = (1.0f - ks) * (1.0f - mat.metallic);
        float jacobian = g_rcp(g_max(0.00001f, 4.0f * NdotV * NdotL));
        V3 f_specular = mat.specular_tint * D * G * F * jacobian;
        float pdf_specular = g_clamp(D * NdotH * jacobian, 0.0f, 1.0f);
        float pdf_diffuse = g_clamp(NdotL / kPi, 0.0f, 1.0f);
        pdf = kd * pdf_diffuse + ks * pdf_specular;
        return kd * f_diffuse + f_specular;
    }
    // fresnel_dielectric(abs(VoH), eta), raytracer.comp:691-695
    float r0 = (1.0f - eta) / (1.0f + eta);
    float F0d = r0 * r0;
    float F = F0d + (1.0f - F0d) * pw;
    float x = g_abs(VoH) / g_max(0.00001f, g_abs(VoN) * g_abs(LoN));
    float denom = eta_i * VoH + eta_o * LoH;
    float jacobian = (eta_o * eta_o * g_abs(LoH)) / g_max(0.00001f, denom * denom);
    pdf = D * g_abs(NoH) * jacobian;
    return mat.subsurface * x * jacobian * D * G * (1.0f - F);
}

}  // namespace rtd
