// rt_kernels.hip — the path-tracing kernel for gfx950 (MI355X).
//
// One launch = one execution of shaders/raytracer.comp::main for every pixel this context
// owns (raytracer.comp:956-996).  The per-pixel program of the shader is restructured as
// a per-lane state machine so that the only expensive operation, the nearest-hit search
// (hit_scene, raytracer.comp:469-508), is executed at ONE program point by all 64 lanes
// of a wave together — whatever sample, bounce or purpose (path vertex or light
// visibility) each lane is at.  Lanes whose pixel is finished pull the next pixel from a
// wave-local reserve refilled by one atomic per 64 pixels (ballot + popcount ranking), so
// terminated paths do not leave lanes idle.  Each pixel is still computed by exactly one
// lane, sequentially, with its own RNG state: results do not depend on scheduling.
#include "rt_trace.h"

namespace rtd {

namespace {

// Resumable nearest-hit search over the tree - WIDE: its four-wide form, two primitives per leaf step; else the
// binary one (both child boxes in the parent), one primitive per leaf step - with a per-lane
// stack in LDS (column `tid` of a [depth][block] array, lane-private, bank = tid mod 32 at
// every depth, so conflict-free).  Each loop iteration the wave votes (ballot + popcount)
// for ONE kind of step — visit an internal node, or test a leaf's next primitive(s) — and only the
// lanes whose cursor is of that kind take it; the minority waits a step instead of dragging
// the whole wave through both code paths.  The loop returns as soon as `exit_quota` lanes
// have finished (or none is left), so that finished lanes go on to shading and to their
// next ray while the stragglers keep their cursor, stack and best hit and resume on the
// next call: a long traversal no longer holds 63 idle lanes hostage.
template <bool WIDE, bool STATS>
RT_DEV void trace_bvh_steps(const DeviceScene& sc, V3 ro, V3 rd, const LdsStack stack, Trav& tv, float t_occl, bool wants_result,
                            int exit_quota, uint32_t& n_nodes, uint32_t& n_tests, uint32_t& n_iters,
                            unsigned long long& wave_iters) {
    const Slab sl = make_slab(ro, rd);

    for (;;) {
        const unsigned long long m_node = __ballot(tv.cur >= 0 && tv.cur != kTravDone);
        const unsigned long long m_leaf = __ballot(tv.cur < 0);
        if ((m_node | m_leaf) == 0ull) break;
        const unsigned long long m_done = __ballot(wants_result && tv.cur == kTravDone);
        if (__popcll(m_done) >= exit_quota) break;
        if (STATS) wave_iters++;
        if ((int)__popcll(m_node) >= (int)__popcll(m_leaf)) {
            if (tv.cur >= 0 && tv.cur != kTravDone) {
                if (WIDE) trav_wide_step<false>(sc.nodes, sl, stack, tv);
                else trav_node_step(sc.nodes, sl, stack, tv);
                if (STATS) { n_nodes++; n_iters++; }
            }
        } else {
            if (tv.cur < 0) {
                if (STATS) { n_tests += (WIDE && ((tv.cur >> 28) & 7) > 0) ? 2u : 1u; n_iters++; }
                if (WIDE) trav_leaf_step2(sc.prims_bvh, ro, rd, stack, tv);  // the wide form comes with leaf pairs
                else trav_leaf_step(sc.prims_bvh, ro, rd, stack, tv);
                // a light-visibility ray is decided once something lies clearly in front of the light
                if (tv.best_t < t_occl) tv.cur = kTravDone;
            }
        }
    }
}

// ST_BOUNCE: the shading point is in `sh` and its direct term is done; sample_mat + eval_mat follow.
// A lane that owes a direct term evaluates the BSDF for it in one round and for its bounce in
// the next, so that a round has ONE eval_mat site shared by both kinds of lanes.
enum { ST_IDLE = 0, ST_TRACE_PATH = 1, ST_TRACE_LIGHT = 2, ST_BOUNCE = 3 };

template <int ACCEL, bool STATS>
__global__ void __launch_bounds__(kBlock, 4) render_kernel(const DeviceScene sc, const FrameArgs fa) {
    __shared__ uint32_t s_stack[(ACCEL >= 1) ? (kStackDepth * kBlock) : 1];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const float4* __restrict__ recs = (ACCEL >= 1) ? sc.prims_bvh : sc.prims_flat;

    // ---- per-lane path state
    int state = ST_IDLE;
    uint32_t seed = 0;
    int px = 0, py = 0, sample = 0, bounce = 0;
    V3 ro = v3(0.0f), rd = v3(0.0f, 0.0f, 1.0f);  // path ray
    V3 att = v3(1.0f), col = v3(0.0f), pix = v3(0.0f);
    uint32_t phash = 0;
    uint32_t cost = 0;  // path segments of the pixel so far
    // saved across a light-visibility search
    Hit sh;
    sh.p = v3(0.0f); sh.normal = v3(0.0f); sh.mat = 0; sh.t = 0.0f; sh.front_face = false;
    V3 s_L = v3(0.0f);
    float s_d = 0.0f;
    float t_occl = -kInf;  // light-visibility rays: a hit nearer than this settles the query (see below)
    int s_light = 0;
    // counters
    uint32_t n_rays = 0, n_nodes = 0, n_tests = 0, n_iters = 0;
    unsigned long long cyc_gen = 0, cyc_trace = 0, cyc_shade = 0, wave_iters = 0;  // STATS only
    unsigned long long cyc_part[5] = {0, 0, 0, 0, 0};  // continuation: hit+light pick, direct BSDF, sampling, indirect BSDF, sample end
    // wave-uniform pixel reserve
    uint32_t res_next = 0, res_left = 0;
    bool exhausted = false;
    // resumable search state; `fresh` marks a ray whose search has not started yet.  The ray handed
    // to the search is the path ray (ro, rd), or for a light query (sh.p, s_L)
    Trav tv;
    tv.cur = kTravDone; tv.sp = 0; tv.best_t = kInf; tv.best_prio = 0x7fffffff; tv.best_idx = -1;
    bool fresh = false;

    const float aspect = (float)fa.width / (float)fa.height;  // raytracer.comp:187

    for (;;) {
        long long stamp0 = 0, stamp1 = 0, stamp2 = 0;
        if (STATS) stamp0 = (long long)__builtin_readcyclecounter();
        // ------------------------------------------------------------------ refill idle lanes
        const unsigned long long idle = __ballot(state == ST_IDLE);
        if (idle != 0ull) {
            if (res_left == 0u && !exhausted) {
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(&fa.counters[0], 64u);
                b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                if (b >= fa.total_slots) {
                    exhausted = true;
                } else {
                    res_next = b;
                    res_left = fa.total_slots - b;
                    if (res_left > 64u) res_left = 64u;
                }
            }
            if (res_left != 0u) {
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (state == ST_IDLE && rank < res_left) {
                    const FrameArgs ff = kernarg_frame();  // once per pixel: loaded here, not held in scalar registers
                    int x, y;
                    if (slot_to_pixel(ff, res_next + rank, x, y)) {
                        px = x;
                        py = y;
                        // raytracer.comp:959-961
                        seed = rng_hash(ff.time_u * 1920u) ^ rng_hash(ff.frame_count) ^ rng_hash((uint32_t)(x + y * 1920));
                        pix = v3(0.0f);
                        sample = 0;
                        phash = 0;
                        cost = 0;
                        state = -1;  // start a sample below
                    }
                }
                uint32_t want = (uint32_t)__popcll(idle);
                uint32_t granted = want < res_left ? want : res_left;
                res_next += granted;
                res_left -= granted;
            }
        }

        // ------------------------------------------------------------------ begin a sample: get_ray, raytracer.comp:933-952
        if (state == -1) {
            float sx = rng_random(seed) - 0.5f;
            float sy = rng_random(seed) - 0.5f;
            float fx = (float)px + sx;
            float fy = (float)py + sy;
            float ndcX = 2.0f * fx / (float)fa.width - 1.0f;
            float ndcY = 1.0f - 2.0f * fy / (float)fa.height;
            const KernargFloats m = kernarg_camera();  // view_inv[16], cam_pos[3], tan_half_fov
            const float tan_half_fov = m[19];
            V3 dc = normalize(v3(ndcX * aspect * tan_half_fov, ndcY * tan_half_fov, -1.0f));
            V3 dw = v3(m[0] * dc.x + m[4] * dc.y + m[8] * dc.z + m[12] * 0.0f,
                       m[1] * dc.x + m[5] * dc.y + m[9] * dc.z + m[13] * 0.0f,
                       m[2] * dc.x + m[6] * dc.y + m[10] * dc.z + m[14] * 0.0f);
            rd = normalize(dw);
            ro = v3(m[16], m[17], m[18]);
            att = v3(1.0f);
            col = v3(0.0f);
            bounce = 0;
            state = ST_TRACE_PATH;
            fresh = true;
        }

        if (__ballot(state != ST_IDLE) == 0ull) {
            if (exhausted) break;
            continue;
        }

        // ------------------------------------------------------------------ nearest-hit search
        if (STATS) stamp1 = (long long)__builtin_readcyclecounter();
        const V3 qo = (state == ST_TRACE_LIGHT) ? sh.p : ro;
        const V3 qd = (state == ST_TRACE_LIGHT) ? s_L : rd;
        if (state != ST_IDLE && fresh) {
            fresh = false;
            n_rays++;
            tv.cur = (ACCEL >= 1) ? 0 : kTravDone;  // node 0 is always an internal node
            tv.sp = 0;
            tv.best_t = kInf;
            tv.best_prio = 0x7fffffff;
            tv.best_idx = -1;
            // A light-visibility query only asks whether the nearest hit lies within 0.1 of the
            // sampled point's distance s_d (raytracer.comp:617-626, 662-671), or, for a
            // directional light, whether anything is hit at all (636-644).  Its search is
            // therefore bounded to t < s_d + 0.2 and ends at the first hit nearer than s_d - 0.2:
            // either way the comparison the shader makes comes out the same.  Not when the
            // per-pixel hash of nearest-hit ids is recorded, which needs the true nearest hit.
            t_occl = -kInf;
            if (ACCEL >= 1 && state == ST_TRACE_LIGHT && !(fa.flags & 2u)) {
                if (__float_as_int(table_row(sc.lights, s_light, 48u)[2].x) == 3 /* DIRECTIONAL */) {
                    t_occl = kInf;
                } else {
                    t_occl = s_d - 0.2f;
                    tv.best_t = s_d + 0.2f;
                }
            }
            if (ACCEL == 0) {
                trace_flat(sc, qo, qd, tv.best_t, tv.best_idx);
                if (STATS) n_tests += (uint32_t)sc.n_prims;
            }
        }
        if (ACCEL >= 1) {
            // leave the loop once half of the wave's busy lanes have a result to shade
            const int busy_lanes = __popcll(__ballot(state != ST_IDLE));
            int quota = (busy_lanes * fa.exit_quota_pct + 99) / 100;
            quota = quota < 1 ? 1 : quota;
            trace_bvh_steps<ACCEL == 2, STATS>(sc, qo, qd, LdsStack{&s_stack[tid]}, tv, t_occl, state != ST_IDLE, quota, n_nodes, n_tests, n_iters, wave_iters);
        }
        if (STATS) stamp2 = (long long)__builtin_readcyclecounter();
        // lanes whose search is still running keep their state and skip the continuation
        const bool ready = (state != ST_IDLE) && (tv.cur == kTravDone);
        const float hit_t = tv.best_t;
        const int hit_idx = tv.best_idx;
        // ------------------------------------------------------------------ continuation
        bool need_shade = false, have_direct = false, end_sample = false;
        Hit& h = sh;  // a lane that takes a new path hit no longer needs the shading point it kept
        const bool is_primary = (ready && state == ST_TRACE_PATH && sample == 0 && bounce == 0);
        const bool traced = ready && (state == ST_TRACE_PATH || state == ST_TRACE_LIGHT);
        V3 Le = v3(0.0f), Ldir = v3(0.0f);
        float light_pdf = 0.0f;
        int prim_id = -1;

        if (ready && state == ST_TRACE_LIGHT) {
            // second half of sample_light (raytracer.comp:617-626, 636-644, 662-671)
            if (hit_idx >= 0) prim_id = __float_as_int(table_row(recs, hit_idx, 48u)[0].w);
            const float4 lc = table_row(sc.lights, s_light, 48u)[1];
            const int ltype = __float_as_int(table_row(sc.lights, s_light, 48u)[2].x);
            Ldir = sh.normal;
            light_pdf = 0.00001f;
            if (ltype == 3 /* DIRECTIONAL */) {
                if (hit_idx < 0) {
                    Ldir = s_L;
                    light_pdf = 1.0f;
                    Le = v3(lc.x, lc.y, lc.z);
                }
            } else if (hit_idx >= 0 && g_abs(s_d - hit_t) < 0.1f) {
                float d2 = s_d * s_d;
                Ldir = s_L;
                light_pdf = (ltype == 1 /* SPHERE */) ? (1.0f / 2.0f / kPi) : 1.0f;
                Le = g_clamp(v3(lc.x, lc.y, lc.z) * lc.w / d2, 0.0f, 1.0f);
            }
            have_direct = true;
            need_shade = true;
        } else if (ready && state == ST_TRACE_PATH) {
            if (hit_idx < 0) {
                col = col + att * skybox_color(fa.skybox, rd);  // raytracer.comp:575-577, 924
                end_sample = true;
            } else {
                make_hit(sc, recs, hit_idx, hit_t, ro, rd, h, prim_id);
                const Material mat = load_material(sc.materials, h.mat);
                if (mat.albedo_a < 1.0f && mat.albedo_a < rng_random(seed)) {  // raytracer.comp:895-898
                    ro = h.p;
                    fresh = true;
                    bounce++;
                    if (bounce > fa.max_bounces) end_sample = true;
                } else if (mat.emission_a > 0.0f) {  // raytracer.comp:901-904
                    col = col + att * mat.emission;
                    end_sample = true;
                } else if (bounce == 0) {
                    // first half of sample_light (raytracer.comp:582-616, 633-635, 653-661)
                    need_shade = true;
                    have_direct = true;
                    if (!(fa.total_lights == 0 || fa.lights_strength_sum <= 0.0f)) {
                        float rand_strength = rng_random(seed) * fa.lights_strength_sum;
                        int low = 0, high = fa.total_lights - 1;
                        while (low < high) {
                            int mid = (low + high) / 2;
                            if (rand_strength <= table_row(sc.lights, mid, 48u)[2].y) high = mid; else low = mid + 1;
                        }
                        const float4 lp = table_row(sc.lights, low, 48u)[0];
                        const float4 lc = table_row(sc.lights, low, 48u)[1];
                        const int ltype = __float_as_int(table_row(sc.lights, low, 48u)[2].x);
                        bool search = false;
                        if (ltype == 0 /* AMBIENT */) {
                            Ldir = random_vec_on_hemisphere(h.normal, seed);
                            light_pdf = 1.0f / 2.0f / kPi;
                            Le = v3(lc.x, lc.y, lc.z);
                        } else if (ltype == 1 /* SPHERE */) {
                            V3 center = v3(lp.x, lp.y, lp.z);
                            V3 center_to_point = h.p - center;
                            V3 sphere_point = random_vec_on_hemisphere(normalize(center_to_point), seed) * lp.w + center;
                            V3 pts = sphere_point - h.p;
                            s_d = length(pts);
                            s_L = normalize(pts);
                            search = true;
                        } else if (ltype == 3 /* DIRECTIONAL */) {
                            V3 light_dir = -v3(lp.x, lp.y, lp.z);
                            if (dot(h.normal, light_dir) > 0.0f) {
                                s_L = light_dir;
                                s_d = 0.0f;
                                search = true;
                            } else {
                                Ldir = h.normal;
                                light_pdf = 0.00001f;
                            }
                        } else if (ltype == 6 /* TRIANGLE */) {
                            const int ti = (int)lp.x;
                            const float4 t0 = table_row(sc.triangles, ti, 64u)[0];
                            const float4 t1 = table_row(sc.triangles, ti, 64u)[1];
                            const float4 t2 = table_row(sc.triangles, ti, 64u)[2];
                            float e1 = g_sqrt(rng_random(seed));
                            float e2 = rng_random(seed);
                            V3 tri_point = (1.0f - e1) * v3(t0.x, t0.y, t0.z) + e1 * (1.0f - e2) * v3(t1.x, t1.y, t1.z) +
                                           e1 * e2 * v3(t2.x, t2.y, t2.z);
                            V3 pts = tri_point - h.p;
                            s_d = length(pts);
                            s_L = normalize(pts);
                            search = true;
                        } else {
                            // POINT / CONE / AREA fall through the shader's switch (raytracer.comp:628-651)
                            // and return vec3(1) with L and pdf never written; they are taken as zero.
                            Le = v3(1.0f);
                        }
                        if (search) {
                            s_light = low;
                            fresh = true;
                            state = ST_TRACE_LIGHT;
                            need_shade = false;
                        }
                    }
                } else {
                    need_shade = true;
                }
            }
        } else if (ready && state == ST_BOUNCE) {
            need_shade = true;
        }
        if (is_primary && (fa.flags & 1u) && fa.primary_ids != nullptr) fa.primary_ids[py * fa.width + px] = prim_id;
        if (traced) phash = (phash ^ (uint32_t)(prim_id + 1)) * 0x01000193u;

        long long sp1 = 0, sp2 = 0, sp3 = 0, sp4 = 0;
        if (STATS) sp1 = (long long)__builtin_readcyclecounter();
        if (need_shade) {
            const Material mat = load_material(sc.materials, h.mat);
            const V3 V = -rd;
            // direct_light, raytracer.comp:870-881.  With no emission arriving (Le == 0: occluded,
            // facing away, no lights) the term is clamp(0 * fr * cos / pdf) = +0 whatever fr is
            // (a NaN clamps to 0 too), and the attenuation at bounce 0 is 1: nothing to add.
            const bool direct_now = have_direct && (Le.x != 0.0f || Le.y != 0.0f || Le.z != 0.0f);
            if (STATS) sp2 = (long long)__builtin_readcyclecounter();
            // indirect bounce, raytracer.comp:913-921
            // Registers are scarcest across eval_mat: only the direction, its cosine with the
            // normal, and (parked in the light-query variables, which this lane no longer needs)
            // the arriving radiance and the light's pdf stay live.
            V3 Lq;
            if (direct_now) {
                Lq = Ldir;
                s_L = Le;
                s_d = light_pdf;
            } else {
                Lq = sample_mat(mat, V, h, seed);
            }
            if (STATS) sp3 = (long long)__builtin_readcyclecounter();
            const float cos_raw = dot(h.normal, Lq);
            float mat_pdf;
            V3 fr = eval_mat(mat, Lq, V, h, mat_pdf);
            if (direct_now) {
                float cos_theta = g_max(0.0f, cos_raw);
                float pdf = power_heuristics(s_d, mat_pdf);
                col = col + g_clamp(s_L * fr * cos_theta / pdf, 0.0f, 1.0f) * att;
                state = ST_BOUNCE;
            } else {
                float cos_theta = g_abs(cos_raw);
                att = att * g_max(v3(0.0f), fr * cos_theta / g_max(0.00001f, mat_pdf));
                ro = h.p;
                rd = Lq;
                fresh = true;
                bounce++;
                state = ST_TRACE_PATH;
                if (bounce > fa.max_bounces) end_sample = true;
            }
        }
        if (STATS) {
            sp4 = (long long)__builtin_readcyclecounter();
            // sp2 / sp3 were taken inside a divergent branch: spread the value to every lane
            for (int off = 32; off > 0; off >>= 1) {
                long long o2 = __shfl_xor(sp2, off), o3 = __shfl_xor(sp3, off);
                sp2 = o2 > sp2 ? o2 : sp2;
                sp3 = o3 > sp3 ? o3 : sp3;
            }
            if (sp2 == 0) { sp2 = sp1; sp3 = sp1; }  // no lane of this wave shaded this round
        }

        if (end_sample) {
            pix = pix + g_clamp(col, 0.0f, 1.0f);  // raytracer.comp:925/929, 974
            sample++;
            cost += (uint32_t)(bounce + 1);
            if (sample < fa.spp) {
                state = -1;
            } else {
                // raytracer.comp:978-995
                const FrameArgs ff = kernarg_frame();
                V3 c = g_pow(pix / (float)ff.spp, 1.0f / 2.2f);
                const int idx = py * ff.width + px;
                float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                int n = 0;
                if (!ff.reset) {
                    a = ff.accum[idx];
                    n = ff.counts[idx];
                }
                a.x += c.x; a.y += c.y; a.z += c.z; a.w += 1.0f;
                n += 1;
                ff.accum[idx] = a;
                ff.counts[idx] = n;
                ff.bgra[idx] = resolve_pixel(a, n);
                if ((ff.flags & 2u) && ff.path_hash != nullptr) ff.path_hash[idx] = phash;
                report_pixel_cost(ff, px, py, cost);
                state = ST_IDLE;
            }
        }
        if (STATS) {
            long long stamp3 = (long long)__builtin_readcyclecounter();
            cyc_gen += (unsigned long long)(stamp1 - stamp0);
            cyc_trace += (unsigned long long)(stamp2 - stamp1);
            cyc_shade += (unsigned long long)(stamp3 - stamp2);
            cyc_part[0] += (unsigned long long)(sp1 - stamp2);
            cyc_part[1] += (unsigned long long)(sp2 - sp1);
            cyc_part[2] += (unsigned long long)(sp3 - sp2);
            cyc_part[3] += (unsigned long long)(sp4 - sp3);
            cyc_part[4] += (unsigned long long)(stamp3 - sp4);
        }
    }

    // ---- counters: one atomic per wave
    unsigned long long r = n_rays, nn = n_nodes, nt = n_tests, ni = n_iters;
    for (int off = 32; off > 0; off >>= 1) {
        r += __shfl_down(r, off);
        if (STATS) {
            nn += __shfl_down(nn, off);
            nt += __shfl_down(nt, off);
            ni += __shfl_down(ni, off);
        }
    }
    if (lane == 0) {
        atomicAdd((unsigned long long*)(fa.counters + 2), r);
        atomicAdd((unsigned long long*)(fa.counters + 8), r);
        if (STATS) {
            atomicAdd((unsigned long long*)(fa.counters + 4), nn);
            atomicAdd((unsigned long long*)(fa.counters + 6), nt);
            atomicAdd((unsigned long long*)(fa.counters + 10), cyc_gen);
            atomicAdd((unsigned long long*)(fa.counters + 12), cyc_trace);
            atomicAdd((unsigned long long*)(fa.counters + 14), cyc_shade);
            atomicAdd((unsigned long long*)(fa.counters + 16), ni);
            atomicAdd((unsigned long long*)(fa.counters + 18), wave_iters);
            for (int k = 0; k < 5; ++k) atomicAdd((unsigned long long*)(fa.counters + 20 + 2 * k), cyc_part[k]);
        }
    }
}

__global__ void __launch_bounds__(256) resolve_kernel(const float4* __restrict__ accum, const int32_t* __restrict__ counts,
                                                      uint32_t* __restrict__ bgra, int n) {
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) bgra[i] = resolve_pixel(accum[i], counts[i]);
}

// Evaluates one arithmetic primitive of rt_math.h per element (self-test of the contract
// that the device reproduces a host binary32 evaluation bit for bit).
__global__ void __launch_bounds__(256) math_probe_kernel(int fn, const float* __restrict__ x, const float* __restrict__ y,
                                                         float* __restrict__ out, int n) {
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    float a = x[i], b = y[i], r = 0.0f;
    switch (fn) {
        case 0: r = g_sin(a); break;
        case 1: r = g_cos(a); break;
        case 2: r = g_tan(a); break;
        case 3: r = g_acos(a); break;
        case 4: r = g_pow(a, b); break;
        case 5: r = g_log2(a); break;
        case 6: r = g_exp2(a); break;
        case 7: r = g_asin(a); break;
        case 8: r = a / b; break;
        case 9: r = g_sqrt(a); break;
        case 10: { uint32_t s = __float_as_uint(a); r = rng_random(s); } break;
        case 11: r = a * b + a; break;       // must stay two roundings (no contraction)
        case 12: r = g_rcp_sqrt(a * a + b * b); break;
        case 13: r = g_rcp(a); break;
        case 14: r = g_rcp_sqrt(a); break;
        // 100..102: sweeps over all 2^32 bit patterns, element i taking patterns i, i + n, ...: the number of
        // patterns on which the short form and the compiler's IEEE expansion differ (two NaNs are equal)
        // 103: the unguarded reciprocal form, which must differ outside its range (a check of the sweep itself)
        case 100: case 101: case 102: case 103: {
            uint32_t bad = 0;
            for (uint64_t p = (uint64_t)i; p < (1ull << 32); p += (uint64_t)n) {
                const float v = __uint_as_float((uint32_t)p);
                const float s = fn == 100 ? g_rcp(v) : fn == 101 ? g_sqrt(v) : fn == 102 ? g_rcp_sqrt(v) : rcp_short(v);
                const float e = (fn == 100 || fn == 103) ? ieee_div(1.0f, v) : fn == 101 ? ieee_sqrt(v) : ieee_div(1.0f, ieee_sqrt(v));
                if (__float_as_uint(s) != __float_as_uint(e) && !(is_nan(s) && is_nan(e))) ++bad;
            }
            r = (float)bad;
        } break;
        default: break;
    }
    out[i] = r;
}

}  // namespace

namespace {
__global__ void __launch_bounds__(64) xcd_probe_kernel(uint32_t* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}
// one thread: every XCD's slot counter must have passed the XCD's last slot (see take_pixels)
__global__ void __launch_bounds__(64) batch_check_kernel(const uint32_t* xcd_next, uint32_t n_blocks, int n_xcds, int n_frames, uint32_t* abort_flag) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int x = 0; x < n_xcds; ++x) {
        const uint32_t mine = n_blocks > (uint32_t)x ? (n_blocks - (uint32_t)x + (uint32_t)n_xcds - 1u) / (uint32_t)n_xcds : 0u;
        if (xcd_next[32 * x] < (mine << kBlockSlotsLog2) * (uint32_t)n_frames) *(volatile uint32_t*)abort_flag = 1u;
    }
}
}  // namespace

hipError_t probe_xcds(int cu_count, uint32_t* scratch, int* n_xcds, hipStream_t stream) {
    *n_xcds = 0;
    if (cu_count < 1 || cu_count > 4096) return hipSuccess;
    hipLaunchKernelGGL(xcd_probe_kernel, dim3((unsigned)cu_count), dim3(64), 0, stream, scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    uint32_t ids[4096];
    e = hipMemcpyAsync(ids, scratch, (size_t)cu_count * 4, hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    int seen[16] = {0}, top = -1;
    for (int i = 0; i < cu_count; ++i) {
        if (ids[i] > 15u) return hipSuccess;
        seen[ids[i]]++;
        top = (int)ids[i] > top ? (int)ids[i] : top;
    }
    for (int k = 0; k <= top; ++k)
        if (seen[k] == 0) return hipSuccess;
    if (top + 1 <= kXcds) *n_xcds = top + 1;
    return hipSuccess;
}

hipError_t launch_batch_check(const FrameArgs& fa, hipStream_t stream) {
    hipLaunchKernelGGL(batch_check_kernel, dim3(1), dim3(64), 0, stream, fa.xcd_next, fa.total_slots >> kBlockSlotsLog2, fa.n_xcds, fa.n_frames,
                       fa.abort_flag);
    return hipGetLastError();
}

hipError_t launch_math_probe(int fn, const float* x, const float* y, float* out, int n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, fn, x, y, out, n);
    return hipGetLastError();
}

namespace {
template <typename K>
hipError_t launch_persistent(K kernel, const DeviceScene& sc, const FrameArgs& fa, int cu_count, hipStream_t stream) {
    // Persistent grid: as many workgroups as are resident at once (registers and the LDS
    // traversal stacks decide), fewer when the image is small.
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    long long want = ((long long)fa.total_slots + kBlock - 1) / kBlock;
    long long cap = (long long)cu_count * per_cu;
    unsigned grid = (unsigned)(want < cap ? want : cap);
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, stream, sc, fa);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_render(const DeviceScene& sc, const FrameArgs& fa, int accel, bool stats, int cu_count,
                         hipStream_t stream) {
    if (accel == 2)
        return stats ? launch_persistent(render_kernel<2, true>, sc, fa, cu_count, stream)
                     : launch_persistent(render_kernel<2, false>, sc, fa, cu_count, stream);
    if (accel == 1)
        return stats ? launch_persistent(render_kernel<1, true>, sc, fa, cu_count, stream)
                     : launch_persistent(render_kernel<1, false>, sc, fa, cu_count, stream);
    return stats ? launch_persistent(render_kernel<0, true>, sc, fa, cu_count, stream)
                 : launch_persistent(render_kernel<0, false>, sc, fa, cu_count, stream);
}

namespace {

constexpr int kOrderThreads = 1024;
constexpr int kOrderWaves = kOrderThreads / 64;
constexpr int kMaxClasses = 64;
constexpr uint32_t kOrderMaxTiles = 128u * 1024u;
constexpr uint32_t kOrderMaxGroups = kOrderMaxTiles / kOrderThreads;  // workgroups: one tile per thread
// Work area of the three kernels: word 0 = the dearest tile's cost (zero between calls), from byte 256 the
// table of tiles per (class, workgroup), then one class byte per tile.
constexpr size_t kOrderAuxTable = 256, kOrderAuxClass = kOrderAuxTable + (size_t)kMaxClasses * kOrderMaxGroups * 4;
constexpr size_t kOrderAuxBytes = kOrderAuxClass + kOrderMaxTiles;

// Exclusive prefix sum of one value per thread over the workgroup (kOrderThreads threads).
RT_DEV uint32_t order_block_scan(uint32_t v, uint32_t* s_wave) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(inc, off);
        if ((int)lane >= off) inc += up;
    }
    __syncthreads();  // s_wave may still be read from an earlier call
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
    return before + inc - v;
}

// The frame's tiles are put into `classes` classes by cost relative to the dearest tile and listed class
// by class, dearest first, in image order within a class (neighbouring tiles stay neighbours: their rays
// share nodes and materials).  Long pixels thus start early and the frame ends on short ones.  Three
// launches of one tile per thread (one workgroup took 50 us for a 1080p frame: it is instruction-bound):
// the dearest tile; every tile's class and the tiles per (class, workgroup); the placement.

__global__ void __launch_bounds__(kOrderThreads) tile_max_kernel(const uint32_t* __restrict__ cost, uint32_t n, uint32_t* __restrict__ aux) {
    __shared__ uint32_t s_max;
    const uint32_t i = blockIdx.x * kOrderThreads + threadIdx.x;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    uint32_t m = i < n ? cost[i] : 0u;
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(m, off);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63u) == 0u) atomicMax(&s_max, m);
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&aux[0], s_max);
}

__global__ void __launch_bounds__(kOrderThreads) tile_class_kernel(uint32_t* __restrict__ cost, uint32_t n, int classes, uint32_t* __restrict__ aux) {
    __shared__ uint32_t s_hist[kMaxClasses];
    const uint32_t i = blockIdx.x * kOrderThreads + threadIdx.x;
    if (threadIdx.x < (uint32_t)kMaxClasses) s_hist[threadIdx.x] = 0;
    __syncthreads();
    // class = classes - 1 - floor(cost * classes / (max + 1)) in binary32 (monotonic in the cost, which is all
    // the order needs)
    const float per_cost = (float)classes / ((float)aux[0] + 1.0f);
    if (i < n) {
        int c = (int)((float)cost[i] * per_cost);
        c = classes - 1 - (c > classes - 1 ? classes - 1 : c);
        ((uint8_t*)aux + kOrderAuxClass)[i] = (uint8_t)c;
        cost[i] = 0u;  // cleared for the next frame
        atomicAdd(&s_hist[c], 1u);
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)kMaxClasses)
        (aux + kOrderAuxTable / 4)[threadIdx.x * gridDim.x + blockIdx.x] = s_hist[threadIdx.x];
}

__global__ void __launch_bounds__(kOrderThreads) tile_place_kernel(uint32_t* __restrict__ order, uint32_t n, uint32_t* __restrict__ aux,
                                                                   uint32_t* __restrict__ stamp, const uint32_t k0, const uint32_t k1,
                                                                   const uint32_t k2, const uint32_t k3, const uint32_t k4, const uint32_t k5,
                                                                   const uint32_t k6, const uint32_t k7) {
    __shared__ uint32_t s_first[kMaxClasses];                // first position of this workgroup's tiles of each class
    __shared__ uint32_t s_count[kMaxClasses * kOrderWaves];  // [class][wave]: tiles of the class in the wave -> their rank in the workgroup
    __shared__ uint32_t s_wave[kOrderWaves];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t G = gridDim.x, g = blockIdx.x;
    s_count[tid] = 0;
    // positions before this workgroup's tiles of class c = exclusive prefix of the (class, workgroup) table,
    // in that order, at entry (c, g)
    const uint32_t* const table = aux + kOrderAuxTable / 4;
    const uint32_t entries = (uint32_t)kMaxClasses * G, per = (entries + kOrderThreads - 1u) / kOrderThreads;
    const uint32_t e_begin = tid * per < entries ? tid * per : entries, e_end = e_begin + per < entries ? e_begin + per : entries;
    uint32_t mine = 0;
    for (uint32_t e = e_begin; e < e_end; ++e) mine += table[e];
    uint32_t prefix = order_block_scan(mine, s_wave);
    for (uint32_t e = e_begin; e < e_end; ++e) {
        if (e % G == g) s_first[e / G] = prefix;
        prefix += table[e];
    }
    // every lane finds the lanes of its wave with its own class by six ballots (one per bit of the class):
    // its rank among them, and their number
    const uint32_t i = g * kOrderThreads + tid;
    const bool valid = i < n;
    const uint32_t c = valid ? (uint32_t)((const uint8_t*)aux + kOrderAuxClass)[i] : 0u;
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 6; ++bit) {
        const bool set = ((c >> bit) & 1u) != 0u;
        const unsigned long long m = __ballot(set);
        same &= set ? m : ~m;
    }
    const uint32_t rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
    __syncthreads();  // s_count zeroed, s_first written
    if (valid && rank == 0u) s_count[c * kOrderWaves + wave] = (uint32_t)__popcll(same);
    __syncthreads();
    const uint32_t scanned = order_block_scan(s_count[tid], s_wave);  // [class][wave] order
    __syncthreads();
    s_count[tid] = scanned;
    __syncthreads();
    if (valid) order[s_first[c] + (s_count[c * kOrderWaves + wave] - s_count[c * kOrderWaves]) + rank] = i;
    if (g == 0u && tid == 0u) aux[0] = 0u;  // no workgroup of this launch reads it
    if (stamp != nullptr && g == 0u && tid < 8u) {
        const uint32_t k[8] = {k0, k1, k2, k3, k4, k5, k6, k7};
        uint32_t v = k[0];
        for (uint32_t j = 1; j < 8u; ++j) v = (tid == j) ? k[j] : v;
        stamp[tid] = v;  // read by later launches on the same stream only
    }
}
static_assert(kMaxClasses * kOrderWaves == kOrderThreads && kMaxClasses == 64, "one thread per (class, wave) count in the scan; six class bits");

__global__ void __launch_bounds__(256) tile_order_check_kernel(uint32_t* __restrict__ order, uint32_t n, const uint32_t* __restrict__ stamp,
                                                              const uint32_t k0, const uint32_t k1, const uint32_t k2, const uint32_t k3,
                                                              const uint32_t k4, const uint32_t k5, const uint32_t k6, const uint32_t k7) {
    const bool same = stamp[0] == k0 && stamp[1] == k1 && stamp[2] == k2 && stamp[3] == k3 && stamp[4] == k4 && stamp[5] == k5 &&
                      stamp[6] == k6 && stamp[7] == k7;
    if (same) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[i] = i;
}

}  // namespace

uint32_t tile_order_max_tiles() { return kOrderMaxTiles; }
size_t tile_order_aux_bytes() { return kOrderAuxBytes; }

hipError_t launch_tile_order(uint32_t* tile_cost, uint32_t* tile_order, uint32_t n_tiles, int classes, uint32_t* aux, uint32_t* stamp,
                             const uint32_t key[8], hipStream_t stream) {
    if (n_tiles == 0) return hipSuccess;
    if (n_tiles > kOrderMaxTiles || aux == nullptr) return hipErrorInvalidValue;
    classes = classes < 1 ? 1 : (classes > kMaxClasses ? kMaxClasses : classes);
    static const uint32_t zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t* k = key ? key : zero;
    const dim3 grid((n_tiles + kOrderThreads - 1u) / kOrderThreads), block(kOrderThreads);
    hipLaunchKernelGGL(tile_max_kernel, grid, block, 0, stream, tile_cost, n_tiles, aux);
    hipLaunchKernelGGL(tile_class_kernel, grid, block, 0, stream, tile_cost, n_tiles, classes, aux);
    hipLaunchKernelGGL(tile_place_kernel, grid, block, 0, stream, tile_order, n_tiles, aux, stamp, k[0], k[1], k[2], k[3], k[4], k[5], k[6],
                       k[7]);
    return hipGetLastError();
}

hipError_t launch_tile_order_check(uint32_t* tile_order, uint32_t n_tiles, const uint32_t* stamp, const uint32_t key[8],
                                   hipStream_t stream) {
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(tile_order_check_kernel, dim3((n_tiles + 255u) / 256u), dim3(256), 0, stream, tile_order, n_tiles, stamp, key[0],
                       key[1], key[2], key[3], key[4], key[5], key[6], key[7]);
    return hipGetLastError();
}

hipError_t launch_resolve(float4* accum, int32_t* counts, uint32_t* bgra, int n_pixels, hipStream_t stream) {
    if (n_pixels <= 0) return hipSuccess;
    hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, stream, accum, counts, bgra,
                       n_pixels);
    return hipGetLastError();
}

}  // namespace rtd
