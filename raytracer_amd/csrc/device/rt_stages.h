// rt_stages.h — the per-path pieces of the shader's program between two nearest-hit
// searches, operating on one slot of path state in device memory (PathPool).  Shared by the
// staged pipeline (rt_wavefront.hip) and the pooled pipeline (rt_pool.hip).
#pragma once
#include "rt_trace.h"

namespace rtd {

RT_DEV int lane_id() { return (int)(threadIdx.x & 63u); }
RT_DEV unsigned long long lanes_below() { return (1ull << lane_id()) - 1ull; }
RT_DEV V3 xyz(float4 v) { return v3(v.x, v.y, v.z); }
RT_DEV float4 f4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }

// A path result (kind 0: bounce 0, kind 1: later bounces; raytracer.comp:891-921) or a
// light-query result (kind 2; raytracer.comp:617-671, 870-881) for slot s, followed for
// the lanes that reach it by the bounce itself: sample_mat + eval_mat (raytracer.comp:913-921).
// Returns true when the sample ends here (end_item follows); otherwise next_stage is what the
// slot waits for now.
RT_DEV bool shade_item(const DeviceScene& sc, const FrameArgs& fa, const PathPool& pp, const float4* __restrict__ recs, int kind,
                       uint32_t s, uint32_t& next_stage) {
    const uint4 c = pp.ctl[s];
    uint32_t seed = c.x;
    const uint32_t pixel = c.y;
    int bounce = (int)c.w;
    const float4 a4 = pp.att[s];
    V3 att = xyz(a4);
    uint32_t phash = __float_as_uint(a4.w);
    bool do_bsdf = false, to_end = false;
    next_stage = WF_IDLE;
    Hit h;
    h.p = v3(0.0f); h.normal = v3(0.0f); h.mat = 0; h.t = 0.0f; h.front_face = false;
    V3 V = v3(0.0f);
    Material mat;

    if (kind == 2) {
        const uint32_t st = pp.stage[s];
        const float4 o4 = pp.ray_o[s];   // w: distance to the sampled light point
        const float4 d4 = pp.ray_d[s];   // xyz: direction to it; w: light index
        const float4 p4 = pp.hp[s];
        const uint32_t mbits = __float_as_uint(p4.w);
        h.p = xyz(p4);
        h.mat = (int)(mbits & 0x7fffffffu);
        h.front_face = (mbits >> 31) != 0u;
        h.normal = xyz(pp.hn[s]);
        V = -xyz(pp.vdir[s]);
        const int light = (int)__float_as_uint(d4.w);
        const float4 lc = sc.lights[3 * light + 1];
        const int ltype = __float_as_int(sc.lights[3 * light + 2].x);
        V3 Le = v3(0.0f), Ldir = v3(0.0f);
        float light_pdf = 0.0f;
        if (st == WF_LIGHT_SEARCH) {
            const uint2 r = pp.hit[s];
            const int hit_idx = (int)r.y;
            const int prim_id = (hit_idx >= 0) ? __float_as_int(recs[3 * hit_idx].w) : -1;
            phash = (phash ^ (uint32_t)(prim_id + 1)) * 0x01000193u;
            light_arrival(ltype, lc, hit_idx, __uint_as_float(r.x), o4.w, xyz(d4), h.normal, Le, Ldir, light_pdf);
        } else if (st == WF_LIGHT_AMBIENT) {
            Ldir = xyz(d4);
            light_pdf = 1.0f / 2.0f / kPi;
            Le = v3(lc.x, lc.y, lc.z);
        } else if (st == WF_LIGHT_UNDEF) {
            Le = v3(1.0f);
        }
        mat = load_material(sc.materials, h.mat);
        if (Le.x != 0.0f || Le.y != 0.0f || Le.z != 0.0f) {
            // the first thing a sample adds to its radiance (raytracer.comp:907-910)
            V3 col = v3(0.0f) + direct_term(mat, h, V, Le, Ldir, light_pdf, att);
            pp.col[s] = f4(col, 0.0f);
        }
        do_bsdf = true;
    } else {
        const V3 ro = xyz(pp.ray_o[s]);
        const V3 rd = xyz(pp.ray_d[s]);
        const uint2 r = pp.hit[s];
        const int hit_idx = (int)r.y;
        int prim_id = -1;
        if (hit_idx < 0) {
            V3 col = xyz(pp.col[s]) + att * v3(0.3f);  // skybox_color_grey, raytracer.comp:570-577, 924
            pp.col[s] = f4(col, 0.0f);
            to_end = true;
        } else {
            make_hit(sc, recs, hit_idx, __uint_as_float(r.x), ro, rd, h, prim_id);
            mat = load_material(sc.materials, h.mat);
            if (mat.albedo_a < 1.0f && mat.albedo_a < rng_random(seed)) {  // raytracer.comp:895-898
                pp.ray_o[s] = f4(h.p, 0.0f);
                bounce++;
                if (bounce > fa.max_bounces) to_end = true; else next_stage = WF_PATH_NEXT;
            } else if (mat.emission_a > 0.0f) {  // raytracer.comp:901-904
                V3 col = xyz(pp.col[s]) + att * mat.emission;
                pp.col[s] = f4(col, 0.0f);
                to_end = true;
            } else if (kind == 0 && !(fa.total_lights == 0 || fa.lights_strength_sum <= 0.0f)) {
                int light;
                V3 s_L;
                float s_d;
                const int q = light_pick(sc, fa, h, seed, light, s_L, s_d);
                pp.hp[s] = f4(h.p, __uint_as_float((uint32_t)h.mat | (h.front_face ? 0x80000000u : 0u)));
                pp.hn[s] = f4(h.normal, 0.0f);
                pp.vdir[s] = f4(rd, 0.0f);
                pp.ray_o[s] = f4(h.p, s_d);
                pp.ray_d[s] = f4(s_L, __uint_as_float((uint32_t)light));
                next_stage = WF_LIGHT_SEARCH + (uint32_t)q;
            } else {
                V = -rd;
                do_bsdf = true;
            }
        }
        if (kind == 0 && c.z == 0u && (fa.flags & 1u) && fa.primary_ids != nullptr) fa.primary_ids[pixel] = prim_id;
        phash = (phash ^ (uint32_t)(prim_id + 1)) * 0x01000193u;
    }

    if (do_bsdf) {
        V3 bounce_dir = sample_mat(mat, V, h, seed);
        float mat_pdf;
        V3 fr = eval_mat(mat, bounce_dir, V, h, mat_pdf);
        float cos_theta = g_abs(dot(h.normal, bounce_dir));
        att = att * g_max(v3(0.0f), fr * cos_theta / g_max(0.00001f, mat_pdf));
        pp.ray_o[s] = f4(h.p, 0.0f);
        pp.ray_d[s] = f4(bounce_dir, 0.0f);
        bounce++;
        if (bounce > fa.max_bounces) to_end = true; else next_stage = WF_PATH_NEXT;
    }
    pp.ctl[s] = make_uint4(seed, pixel, c.z, (uint32_t)bounce);
    pp.att[s] = f4(att, __uint_as_float(phash));
    if (!to_end) pp.stage[s] = next_stage;
    return to_end;
}

// End of a sample (raytracer.comp:925/929, 974) and either the next sample's primary ray
// (get_ray, raytracer.comp:933-952) or the pixel's gamma, accumulation and rgba8 store
// (raytracer.comp:978-995).
RT_DEV void end_item(const FrameArgs& fa, const PathPool& pp, uint32_t s, bool& alive) {
    const uint4 c = pp.ctl[s];
    uint32_t seed = c.x;
    const uint32_t pixel = c.y;
    const int sample = (int)c.z + 1;
    V3 pix = xyz(pp.pix[s]) + g_clamp(xyz(pp.col[s]), 0.0f, 1.0f);
    const float phash_bits = pp.att[s].w;
    if (sample < fa.spp) {
        const int px = (int)(pixel % (uint32_t)fa.width), py = (int)(pixel / (uint32_t)fa.width);
        V3 ro, rd;
        generate_ray(fa, px, py, seed, ro, rd);
        pp.ctl[s] = make_uint4(seed, pixel, (uint32_t)sample, 0u);
        pp.ray_o[s] = f4(ro, 0.0f);
        pp.ray_d[s] = f4(rd, 0.0f);
        pp.att[s] = make_float4(1.0f, 1.0f, 1.0f, phash_bits);
        pp.col[s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        pp.pix[s] = f4(pix, 0.0f);
        pp.stage[s] = WF_PATH_FIRST;
        alive = true;
    } else {
        V3 cc = g_pow(pix / (float)fa.spp, 1.0f / 2.2f);
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        int n = 0;
        if (!fa.reset) {
            a = fa.accum[pixel];
            n = fa.counts[pixel];
        }
        a.x += cc.x; a.y += cc.y; a.z += cc.z; a.w += 1.0f;
        n += 1;
        fa.accum[pixel] = a;
        fa.counts[pixel] = n;
        fa.bgra[pixel] = resolve_pixel(a, n);
        if ((fa.flags & 2u) && fa.path_hash != nullptr) fa.path_hash[pixel] = __float_as_uint(phash_bits);
        pp.stage[s] = WF_IDLE;
        alive = false;
    }
}

}  // namespace rtd
