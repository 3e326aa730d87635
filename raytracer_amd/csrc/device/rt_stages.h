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

// A sample's light query (raytracer.comp:907-910 -> 870-881 -> 582-679) runs BESIDE its path: every
// random draw of sample_light and of sample_mat happens before either of the two searches that follow a
// first-bounce hit, and the query's only product is one term of the sample's radiance.  So the stage
// that handles the first hit picks the light, notes what the term needs, sends off the visibility ray
// and the bounce ray together, and the sample's end evaluates the term if the light proved visible.
// Per-slot flag bits:
enum : uint32_t { kPixelFrameShift = 27, kPixelIndexMask = (1u << 27) - 1u };  // a slot's pixel word: index | frame of the launch << 27
static_assert(kMaxBatchFrames <= (1 << (32 - kPixelFrameShift)), "the frame of a launch must fit the pixel word's top bits");
enum : uint32_t {
    LF_PENDING = 1u,   // this sample has a light query in flight or finished
    LF_DONE = 2u,      // its search has ended
    LF_LIT = 4u,       // ... and the light is visible from the shading point
    LF_PATH_DONE = 8u, // the path has ended; the sample waits for the query only
    LF_FOLDED = 16u,   // (per-pixel hash only) the query's hit id has been folded into the hash
    LF_FETCHED = 32u   // a lane has taken the query's ray (which sits in slot fields the path will reuse)
};

// What the light's term needs at the sample's end: the shading point's normal, material and facing, the
// direction of the path ray that reached it, the direction and distance of the sampled light point and the
// light's index.
struct LightRecord {
    V3 normal, vdir, ldir;
    float s_d;
    int mat, light;
    bool front_face;
};

// A path result: `kind` 0 = the hit of a sample's first ray (bounce 0), 1 = of a later one
// (raytracer.comp:891-921), followed for the lanes that reach it by the bounce itself: sample_mat +
// eval_mat (913-921).  Returns true when the PATH ends here.  Otherwise the slot's next path ray is in
// its ro / rd.  light_kind: 0 = no light query to send, 2 = a visibility ray towards a sampled point,
// 3 = towards a directional light; its origin is the slot's ro (the bounce starts there too), the vector
// to the light point sits in the slot's sample radiance (unused until the path adds to it).
// `retry`: the item cannot be handled yet and must be queued again - its sample's light query has not been
// taken by a lane yet (its ray would be overwritten), or, in hash mode, has not reported its hit id.
template <typename Slot>
RT_DEV bool shade_item(const DeviceScene& sc, const FrameArgs& fa, const Slot& sl, const float4* __restrict__ recs, int kind,
                       int& light_kind, bool& retry) {
    uint32_t seed, pixel, sample, phash;
    int bounce;
    light_kind = 0;
    retry = false;
    const bool hash_mode = (fa.flags & 2u) != 0u;
    if (kind == 1) {
        const uint32_t f = sl.get_flags();
        if (f & LF_PENDING) {
            if (!(f & LF_FETCHED)) { retry = true; return false; }
            if (hash_mode && !(f & LF_FOLDED)) {
                // the reference folds ids in program order: primary hit, light query hit, later hits
                if (!(f & LF_DONE)) { retry = true; return false; }
                (void)sl.get_att(phash);
                sl.put_hash((phash ^ (uint32_t)(sl.get_light_hit_id() + 1)) * 0x01000193u);
                sl.or_flags(LF_FOLDED);
            }
        }
    }
    sl.get_ctl(seed, pixel, sample, bounce);
    V3 att = sl.get_att(phash);
    bool do_bsdf = false, to_end = false;
    Hit h;
    h.p = v3(0.0f); h.normal = v3(0.0f); h.mat = 0; h.t = 0.0f; h.front_face = false;
    Material mat;
    const V3 ro = sl.get_ro();
    const V3 rd = sl.get_rd();
    float hit_t;
    int hit_idx;
    sl.get_hit(hit_t, hit_idx);
    int prim_id = -1;
    uint32_t flags = 0;  // kind 0: the new sample's flags
    if (hit_idx < 0) {
        sl.put_col(sl.get_col() + att * skybox_color(fa.skybox, rd));  // raytracer.comp:575-577, 924
        to_end = true;
    } else {
        make_hit(sc, recs, hit_idx, hit_t, ro, rd, h, prim_id);
        mat = load_material(sc.materials, h.mat);
        if (mat.albedo_a < 1.0f && mat.albedo_a < rng_random(seed)) {  // raytracer.comp:895-898
            sl.put_ro(h.p);
            bounce++;
            if (bounce > fa.max_bounces) to_end = true;
        } else if (mat.emission_a > 0.0f) {  // raytracer.comp:901-904
            sl.put_col(sl.get_col() + att * mat.emission);
            to_end = true;
        } else {
            if (kind == 0 && !(fa.total_lights == 0 || fa.lights_strength_sum <= 0.0f)) {
                // direct_light (raytracer.comp:870-881), first half: pick and sample the light (582-616,
                // 633-635, 653-661)
                LightRecord q;
                V3 to_light;
                const int k = light_pick(sc, fa, h, seed, q.light, q.ldir, q.s_d, to_light);
                if (k != 2) {  // k == 2: a directional light below the horizon, nothing arrives
                    q.normal = h.normal;
                    q.vdir = rd;
                    q.mat = h.mat;
                    q.front_face = h.front_face;
                    sl.put_light_record(q);
                    if (k == 0) {  // visibility to be searched
                        sl.put_col(to_light);  // the query's ray, until a lane has taken it
                        light_kind = (__float_as_int(table_row(sc.lights, q.light, 48u)[2].x) == 3 /* DIRECTIONAL */) ? 3 : 2;
                        flags = LF_PENDING;
                    } else {  // no search: AMBIENT (not shadow-tested) and the shader's undefined light types count as lit
                        flags = LF_PENDING | LF_FETCHED | LF_DONE | LF_LIT | LF_FOLDED;
                    }
                }
            }
            do_bsdf = true;
        }
    }
    if (kind == 0) {
        sl.put_flags(flags);
#if !defined(RT_TAIL_DIAG)
        if (sample == 0u && (fa.flags & 1u) && fa.primary_ids != nullptr) fa.primary_ids[pixel & kPixelIndexMask] = prim_id;
#endif
    }
    phash = (phash ^ (uint32_t)(prim_id + 1)) * 0x01000193u;

    if (do_bsdf) {
        const V3 V = -rd;
        V3 bounce_dir = sample_mat(mat, V, h, seed);
        float mat_pdf;
        V3 fr = eval_mat(mat, bounce_dir, V, h, mat_pdf);
        float cos_theta = g_abs(dot(h.normal, bounce_dir));
        att = att * g_max(v3(0.0f), fr * cos_theta / g_max(0.00001f, mat_pdf));
        sl.put_ro(h.p);
        sl.put_rd(bounce_dir);
        bounce++;
        if (bounce > fa.max_bounces) to_end = true;
    }
    sl.put_ctl(seed, pixel, sample, bounce);
    sl.put_att(att, phash);
    return to_end;
}

// The light's term of a sample whose light proved visible or needs no search (raytracer.comp:602-606, 617-626,
// 636-644, 662-671, then 870-881), evaluated from the record its first hit left: the sample's radiance after the term.
template <typename Slot>
RT_DEV V3 lit_term(const DeviceScene& sc, const Slot& sl) {
    LightRecord q;
    sl.get_light_record(q);
    Hit h;
    h.p = v3(0.0f); h.t = 0.0f;
    h.normal = q.normal; h.mat = q.mat; h.front_face = q.front_face;
    const float4 lc = table_row(sc.lights, q.light, 48u)[1];
    const int ltype = __float_as_int(table_row(sc.lights, q.light, 48u)[2].x);
    V3 Le;
    float light_pdf;
    if (ltype == 3 /* DIRECTIONAL */) {
        light_pdf = 1.0f;
        Le = v3(lc.x, lc.y, lc.z);
    } else if (ltype == 0 /* AMBIENT */) {
        light_pdf = 1.0f / 2.0f / kPi;
        Le = v3(lc.x, lc.y, lc.z);
    } else if (ltype == 1 /* SPHERE */ || ltype == 6 /* TRIANGLE */) {
        light_pdf = (ltype == 1) ? (1.0f / 2.0f / kPi) : 1.0f;
        Le = g_clamp(v3(lc.x, lc.y, lc.z) * lc.w / (q.s_d * q.s_d), 0.0f, 1.0f);
    } else {  // POINT / CONE / AREA fall through the shader's switch: vec3(1) with L and pdf never written (zero here)
        light_pdf = 0.0f;
        Le = v3(1.0f);
    }
    // with no emission arriving the term is +0 whatever the BSDF says
    if (!(Le.x != 0.0f || Le.y != 0.0f || Le.z != 0.0f)) return v3(0.0f);
    const Material mat = load_material(sc.materials, q.mat);
    return v3(0.0f) + direct_term(mat, h, -q.vdir, Le, q.ldir, light_pdf, v3(1.0f));  // the attenuation at bounce 0 is 1
}

// First primary ray of a sample (get_ray, raytracer.comp:933-952) into the slot.
template <typename Slot>
RT_DEV void start_sample(const FrameArgs& fa, const Slot& sl, uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t phash) {
    const uint32_t index = pixel & kPixelIndexMask;  // (the frame of a multi-frame launch rides in the top bits)
    const int px = (int)(index % (uint32_t)fa.width), py = (int)(index / (uint32_t)fa.width);
    V3 ro, rd;
    generate_ray(fa, sl.camera(fa), px, py, seed, ro, rd);
    sl.put_ctl(seed, pixel, sample, 0);
    sl.put_ro(ro);
    sl.put_rd(rd);
    sl.put_att(v3(1.0f), phash);
    sl.put_col(v3(0.0f));
}

// The three things a sample's end may consist of are separate stages, so that each runs with full waves: the
// light's term (only for samples whose light proved visible), the next sample's primary ray (15 of 16 ends at 16
// spp), the pixel's end (the last sample's).

// The light's term joins the sample's radiance: added first, as the shader does at bounce 0 (raytracer.comp:907-910).
template <typename Slot>
RT_DEV void lit_item(const DeviceScene& sc, const Slot& sl) {
    sl.put_col(lit_term(sc, sl) + sl.get_col());
}

// Sum of the pixel's samples so far with the one that just ended (raytracer.comp:925/929, 974); folds the light
// query's hit id into the per-pixel hash if no later bounce did.
template <typename Slot>
RT_DEV V3 end_of_sample(const FrameArgs& fa, const Slot& sl, int bounce, float& cost, uint32_t& phash) {
    const uint32_t lf = sl.get_flags();
    V3 pix = sl.get_pix(cost) + g_clamp(sl.get_col(), 0.0f, 1.0f);
    cost += (float)(bounce + 1);  // path segments of the pixel so far (a small integer, exact in binary32)
    (void)sl.get_att(phash);
    if ((fa.flags & 2u) && (lf & LF_PENDING) && !(lf & LF_FOLDED)) phash = (phash ^ (uint32_t)(sl.get_light_hit_id() + 1)) * 0x01000193u;
    return pix;
}

// A sample that is not the pixel's last ends: the next sample's primary ray (raytracer.comp:972-975).
template <typename Slot>
RT_DEV void end_sample_item(const FrameArgs& fa, const Slot& sl) {
    uint32_t seed, pixel, sample, phash;
    int bounce;
    sl.get_ctl(seed, pixel, sample, bounce);
    float cost;
    const V3 pix = end_of_sample(fa, sl, bounce, cost, phash);
    sl.put_pix(pix, cost);
    start_sample(fa, sl, seed, pixel, sample + 1u, phash);
}

// The pixel's last sample ends: gamma, accumulation and rgba8 store (raytracer.comp:978-995).  False: not yet -
// in a launch of several frames the pixel's previous frame has not been stored; the caller asks again later.
template <bool BATCH, typename Slot>
RT_DEV bool end_pixel_item(const FrameArgs& fa, const Slot& sl) {
    uint32_t seed, pixel_word, sample, phash;
    int bounce;
    sl.get_ctl(seed, pixel_word, sample, bounce);
    const uint32_t pixel = pixel_word & kPixelIndexMask, frame = pixel_word >> kPixelFrameShift;
    const auto& ff = sl.late_frame(fa);  // the frame's arguments, possibly loaded only here
    const int py = (int)(pixel / (uint32_t)ff.width), px = (int)pixel - py * ff.width;
    const bool batch = BATCH;  // (a launch of several frames runs the kernel compiled for it: the other pays nothing for this)
    uint32_t block = 0;
    if (batch) {
        // frame k of a pixel follows frame k - 1 of all the pixels of its image block (same XCD: see FrameArgs)
        uint32_t inside;
        block = block_of_pixel(ff, px, py, inside);
        if (frame != 0u && l2_load(&ff.block_done[block]) < frame * inside) return false;
    }
    float cost;
    const V3 pix = end_of_sample(fa, sl, bounce, cost, phash);
    V3 cc = g_pow(pix / (float)ff.spp, 1.0f / 2.2f);
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    int n = 0;
    if (!(ff.reset && frame == 0u)) {
        if (batch) {  // what another CU of this XCD may have stored a moment ago: read where it was written to
            const uint32_t* w = (const uint32_t*)&ff.accum[pixel];
            a = make_float4(__uint_as_float(l2_load(w + 0)), __uint_as_float(l2_load(w + 1)), __uint_as_float(l2_load(w + 2)),
                            __uint_as_float(l2_load(w + 3)));
            n = (int)l2_load((const uint32_t*)&ff.counts[pixel]);
        } else {
            a = ff.accum[pixel];
            n = ff.counts[pixel];
        }
    }
    a.x += cc.x; a.y += cc.y; a.z += cc.z; a.w += 1.0f;
    n += 1;
    ff.accum[pixel] = a;
    ff.counts[pixel] = n;
    ff.bgra[pixel] = resolve_pixel(a, n);
    if ((ff.flags & 2u) && ff.path_hash != nullptr) ff.path_hash[pixel] = phash;
    report_pixel_cost(ff, px, py, (uint32_t)cost);
#if defined(RT_TAIL_DIAG)
    // diagnostic build (tools/tail_diag.py): instead of the primary-hit id, when the pixel started and ended (units of
    // 40.96 us of the constant 100 MHz counter, 12 bits each) and what it cost (path segments, clipped to 255)
    if (ff.primary_ids != nullptr) {
        const uint32_t now = (uint32_t)(__builtin_amdgcn_s_memrealtime() >> 12) & 0xfffu;
        const uint32_t c8 = (uint32_t)cost > 255u ? 255u : (uint32_t)cost;
        ff.primary_ids[pixel] = (int)((sl.get_start() << 20) | (now << 8) | c8);
    }
#endif
    if (batch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have reached the L2 before the block's count says so
        (void)l2_add(&ff.block_done[block], 1u);
    }
    return true;
}

// A slot's first sample of pixel (x, y) in frame `frame` of the launch: seed (raytracer.comp:959-961), empty pixel sum.
template <bool BATCH, typename Slot>
RT_DEV void start_pixel(const FrameArgs& fa, const Slot& sl, int x, int y, uint32_t frame = 0u) {
    const uint32_t frame_count = BATCH ? sl.frame_count_of(frame) : fa.frame_count;
    const uint32_t seed = rng_hash(fa.time_u * 1920u) ^ rng_hash(frame_count) ^ rng_hash((uint32_t)(x + y * 1920));
    sl.put_pix(v3(0.0f), 0.0f);
#if defined(RT_TAIL_DIAG)
    sl.put_start((uint32_t)(__builtin_amdgcn_s_memrealtime() >> 12) & 0xfffu);
#endif
    start_sample(fa, sl, seed, (uint32_t)(y * fa.width + x) | (frame << kPixelFrameShift), 0u, 0u);
}

}  // namespace rtd
