// rt_stages.h — the per-path pieces of the shader's program between two nearest-hit
// searches, written against an accessor of one slot of path state (the pooled pipeline keeps
// the hot part of it in LDS, rt_pool.hip).
#pragma once
#include "rt_trace.h"

namespace rtd {

RT_DEV int lane_id() { return (int)(threadIdx.x & 63u); }
RT_DEV unsigned long long lanes_below() { return (1ull << lane_id()) - 1ull; }
RT_DEV V3 xyz(float4 v) { return v3(v.x, v.y, v.z); }
RT_DEV float4 f4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }

// What a slot keeps across a light query besides the query ray itself (whose origin is the
// shading point): normal, material and facing of the shading point, the direction of the
// path ray that reached it, the distance to the sampled light point and the light's index.
struct LightQuery {
    V3 normal;
    V3 vdir;
    int mat;
    bool front_face;
    float s_d;
    int light;
};

// A path result (kind 0: bounce 0, kind 1: later bounces; raytracer.comp:891-921) or a
// light-query result (kind 2, `stage` = which of the WF_LIGHT_* cases; raytracer.comp:617-671,
// 870-881), followed for the lanes that reach it by the bounce itself: sample_mat + eval_mat
// (raytracer.comp:913-921).  Returns true when the sample ends here (end_item follows);
// otherwise next_stage is what the slot waits for now.
template <typename Slot>
RT_DEV bool shade_item(const DeviceScene& sc, const FrameArgs& fa, const Slot& sl, const float4* __restrict__ recs, int kind,
                       uint32_t stage, uint32_t& next_stage) {
    uint32_t seed, pixel, sample, phash;
    int bounce;
    sl.get_ctl(seed, pixel, sample, bounce);
    V3 att = sl.get_att(phash);
    bool do_bsdf = false, to_end = false;
    next_stage = WF_IDLE;
    Hit h;
    h.p = v3(0.0f); h.normal = v3(0.0f); h.mat = 0; h.t = 0.0f; h.front_face = false;
    V3 V = v3(0.0f);
    Material mat;

    if (kind == 2) {
        LightQuery q;
        sl.get_light_query(q);
        h.p = sl.get_ro();
        const V3 s_L = sl.get_rd();
        h.mat = q.mat;
        h.front_face = q.front_face;
        h.normal = q.normal;
        V = -q.vdir;
        const float4 lc = sc.lights[3 * q.light + 1];
        const int ltype = __float_as_int(sc.lights[3 * q.light + 2].x);
        V3 Le = v3(0.0f), Ldir = v3(0.0f);
        float light_pdf = 0.0f;
        if (stage == WF_LIGHT_SEARCH) {
            float hit_t;
            int hit_idx;
            sl.get_hit(hit_t, hit_idx);
            const int prim_id = (hit_idx >= 0) ? __float_as_int(recs[3 * hit_idx].w) : -1;
            phash = (phash ^ (uint32_t)(prim_id + 1)) * 0x01000193u;
            light_arrival(ltype, lc, hit_idx, hit_t, q.s_d, s_L, h.normal, Le, Ldir, light_pdf);
        } else if (stage == WF_LIGHT_AMBIENT) {
            Ldir = s_L;
            light_pdf = 1.0f / 2.0f / kPi;
            Le = v3(lc.x, lc.y, lc.z);
        } else if (stage == WF_LIGHT_UNDEF) {
            Le = v3(1.0f);
        }
        mat = load_material(sc.materials, h.mat);
        if (Le.x != 0.0f || Le.y != 0.0f || Le.z != 0.0f) {
            // the first thing a sample adds to its radiance (raytracer.comp:907-910)
            sl.put_col(v3(0.0f) + direct_term(mat, h, V, Le, Ldir, light_pdf, att));
        }
        do_bsdf = true;
    } else {
        const V3 ro = sl.get_ro();
        const V3 rd = sl.get_rd();
        float hit_t;
        int hit_idx;
        sl.get_hit(hit_t, hit_idx);
        int prim_id = -1;
        if (hit_idx < 0) {
            sl.put_col(sl.get_col() + att * skybox_color(fa.skybox, rd));  // raytracer.comp:575-577, 924
            to_end = true;
        } else {
            make_hit(sc, recs, hit_idx, hit_t, ro, rd, h, prim_id);
            mat = load_material(sc.materials, h.mat);
            if (mat.albedo_a < 1.0f && mat.albedo_a < rng_random(seed)) {  // raytracer.comp:895-898
                sl.put_ro(h.p);
                bounce++;
                if (bounce > fa.max_bounces) to_end = true; else next_stage = WF_PATH_NEXT;
            } else if (mat.emission_a > 0.0f) {  // raytracer.comp:901-904
                sl.put_col(sl.get_col() + att * mat.emission);
                to_end = true;
            } else if (kind == 0 && !(fa.total_lights == 0 || fa.lights_strength_sum <= 0.0f)) {
                LightQuery q;
                V3 s_L;
                const int k = light_pick(sc, fa, h, seed, q.light, s_L, q.s_d);
                q.normal = h.normal;
                q.vdir = rd;
                q.mat = h.mat;
                q.front_face = h.front_face;
                sl.put_light_query(q);
                sl.put_ro(h.p);
                sl.put_rd(s_L);
                if (k == 0) {
                    // A light-visibility query only asks whether the nearest hit lies within 0.1 of the
                    // sampled point's distance s_d (raytracer.comp:617-626, 662-671), or, for a directional
                    // light, whether anything is hit at all (636-644): its search may be bounded to
                    // t < s_d + 0.2 and end at the first hit nearer than s_d - 0.2.  Not when the per-pixel
                    // hash of nearest-hit ids is recorded, which needs the true nearest hit.
                    float t_max = kInf, t_occl = -kInf;
                    if (!(fa.flags & 2u)) {
                        if (__float_as_int(sc.lights[3 * q.light + 2].x) == 3 /* DIRECTIONAL */) {
                            t_occl = kInf;
                        } else {
                            t_max = q.s_d + 0.2f;
                            t_occl = q.s_d - 0.2f;
                        }
                    }
                    sl.put_search_bounds(t_max, t_occl);
                }
                next_stage = WF_LIGHT_SEARCH + (uint32_t)k;
            } else {
                V = -rd;
                do_bsdf = true;
            }
        }
        if (kind == 0 && sample == 0u && (fa.flags & 1u) && fa.primary_ids != nullptr) fa.primary_ids[pixel] = prim_id;
        phash = (phash ^ (uint32_t)(prim_id + 1)) * 0x01000193u;
    }

    if (do_bsdf) {
        V3 bounce_dir = sample_mat(mat, V, h, seed);
        float mat_pdf;
        V3 fr = eval_mat(mat, bounce_dir, V, h, mat_pdf);
        float cos_theta = g_abs(dot(h.normal, bounce_dir));
        att = att * g_max(v3(0.0f), fr * cos_theta / g_max(0.00001f, mat_pdf));
        sl.put_ro(h.p);
        sl.put_rd(bounce_dir);
        bounce++;
        if (bounce > fa.max_bounces) to_end = true; else next_stage = WF_PATH_NEXT;
    }
    sl.put_ctl(seed, pixel, sample, bounce);
    sl.put_att(att, phash);
    return to_end;
}

// First primary ray of a sample (get_ray, raytracer.comp:933-952) into the slot.
template <typename Slot>
RT_DEV void start_sample(const FrameArgs& fa, const Slot& sl, uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t phash) {
    const int px = (int)(pixel % (uint32_t)fa.width), py = (int)(pixel / (uint32_t)fa.width);
    V3 ro, rd;
    generate_ray(fa, sl.camera(fa), px, py, seed, ro, rd);
    sl.put_ctl(seed, pixel, sample, 0);
    sl.put_ro(ro);
    sl.put_rd(rd);
    sl.put_att(v3(1.0f), phash);
    sl.put_col(v3(0.0f));
}

// End of a sample (raytracer.comp:925/929, 974) and either the next sample's primary ray or
// the pixel's gamma, accumulation and rgba8 store (raytracer.comp:978-995).
template <typename Slot>
RT_DEV void end_item(const FrameArgs& fa, const Slot& sl, bool& alive) {
    uint32_t seed, pixel, sample, phash;
    int bounce;
    sl.get_ctl(seed, pixel, sample, bounce);
    sample += 1u;
    float cost;  // path segments of the pixel so far (a small integer, exact in binary32)
    V3 pix = sl.get_pix(cost) + g_clamp(sl.get_col(), 0.0f, 1.0f);
    cost += (float)(bounce + 1);
    (void)sl.get_att(phash);
    if ((int)sample < fa.spp) {
        sl.put_pix(pix, cost);
        start_sample(fa, sl, seed, pixel, sample, phash);
        alive = true;
    } else {
        const auto& ff = sl.late_frame(fa);  // the frame's arguments, possibly loaded only here
        V3 cc = g_pow(pix / (float)ff.spp, 1.0f / 2.2f);
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        int n = 0;
        if (!ff.reset) {
            a = ff.accum[pixel];
            n = ff.counts[pixel];
        }
        a.x += cc.x; a.y += cc.y; a.z += cc.z; a.w += 1.0f;
        n += 1;
        ff.accum[pixel] = a;
        ff.counts[pixel] = n;
        ff.bgra[pixel] = resolve_pixel(a, n);
        if ((ff.flags & 2u) && ff.path_hash != nullptr) ff.path_hash[pixel] = phash;
        const int py = (int)(pixel / (uint32_t)ff.width);
        report_pixel_cost(ff, (int)pixel - py * ff.width, py, (uint32_t)cost);
        alive = false;
    }
}

// A slot's first sample of pixel (x, y): seed (raytracer.comp:959-961), empty pixel sum.
template <typename Slot>
RT_DEV void start_pixel(const FrameArgs& fa, const Slot& sl, int x, int y) {
    const uint32_t seed = rng_hash(fa.time_u * 1920u) ^ rng_hash(fa.frame_count) ^ rng_hash((uint32_t)(x + y * 1920));
    sl.put_pix(v3(0.0f), 0.0f);
    start_sample(fa, sl, seed, (uint32_t)(y * fa.width + x), 0u, 0u);
}

}  // namespace rtd
