// rt_trace.h — nearest-hit search building blocks shared by the single-kernel pipeline
// (rt_kernels.hip) and the pooled pipeline (rt_pool.hip): one primitive test with the
// reference's tie rules, the resumable BVH step, the hit record and the rgba8 store.
#pragma once
#include <cstddef>
#include "rt_kernels.h"
#include "rt_shade.h"

namespace rtd {

RT_DEV bool surrounds(float x) { return 0.005f < x && kInf > x; }  // Interval(0.005, PINF), raytracer.comp:310-312

// One primitive against one ray; keeps the winner under the reference's tie rules
// (lower t; equal t: lower prio).  Triangle: raytracer.comp:366-406 / 412-449 with the two
// edges precomputed; sphere: raytracer.comp:334-361.
RT_DEV void intersect_loaded(float4 A, float4 B, float4 C, int idx, V3 ro, V3 rd, float& best_t, int& best_prio, int& best_idx) {
    const int prio = __float_as_int(B.w);
    float t;
    if (__float_as_int(C.w) < 0) {  // sphere: A.xyz centre, B.x radius
        V3 oc = v3(A.x, A.y, A.z) - ro;
        float a = dot(rd, rd);
        float h = dot(rd, oc);
        float c = dot(oc, oc) - B.x * B.x;
        float disc = h * h - a * c;
        if (disc < 0.0f) return;
        float srtd = g_sqrt(disc);
        t = (h - srtd) / a;
        if (!surrounds(t)) {
            t = (h + srtd) / a;
            if (!surrounds(t)) return;
        }
    } else {  // triangle: A.xyz = v0, B.xyz = v1 - v0, C.xyz = v2 - v0
        V3 e1 = v3(B.x, B.y, B.z);
        V3 e2 = v3(C.x, C.y, C.z);
        V3 h = cross(rd, e2);
        float a = dot(e1, h);
        if (g_abs(a) < 1e-6f) return;
        float f = g_rcp(a);
        V3 s = ro - v3(A.x, A.y, A.z);
        float u = f * dot(s, h);
        if (u < 0.0f || u > 1.0f) return;
        V3 q = cross(s, e1);
        float v = f * dot(rd, q);
        if (v < 0.0f || u + v > 1.0f) return;
        t = f * dot(e2, q);
        if (!surrounds(t)) return;
    }
    if (t < best_t || (t == best_t && prio < best_prio)) {
        best_t = t;
        best_prio = prio;
        best_idx = idx;
    }
}
RT_DEV void intersect_record(const float4* __restrict__ recs, int idx, V3 ro, V3 rd, float& best_t, int& best_prio,
                             int& best_idx) {
    const float4* __restrict__ r = table_row(recs, idx, 48u);
    intersect_loaded(r[0], r[1], r[2], idx, ro, rd, best_t, best_prio, best_idx);
}

// The reference's search: every primitive, in the order of its three loops.
RT_DEV void trace_flat(const DeviceScene& sc, V3 ro, V3 rd, float& best_t, int& best_idx) {
    int best_prio = 0x7fffffff;
    for (int i = 0; i < sc.n_prims; ++i) intersect_record(sc.prims_flat, i, ro, rd, best_t, best_prio, best_idx);
}

// Traversal cursor of one lane: >= 0 an internal node index (below kTravDone); < 0 a leaf,
// bit 31 | (count-1) << 28 | first record; kTravDone when the search has finished.
static constexpr int kTravDone = 0x7fffffff;
RT_DEV int encode_child(int child, int count) { return count > 0 ? (int)(0x80000000u | ((uint32_t)(count - 1) << 28) | (uint32_t)child) : child; }

struct Trav {
    int cur;        // cursor (see above)
    int sp;         // entries on this lane's LDS stack
    float best_t;   // winner so far
    int best_prio;
    int best_idx;   // record index of the winner, -1 = none
};

RT_DEV void trav_reset(Trav& tv, int cur) {
    tv.cur = cur;
    tv.sp = 0;
    tv.best_t = kInf;
    tv.best_prio = 0x7fffffff;
    tv.best_idx = -1;
}

// Per-lane traversal stack.  LdsStack: column of a [depth][block] LDS array (bank = tid mod
// 32 at every depth, so conflict-free).  ShortStack: the same for the first `depth` entries,
// deeper entries in a lane-private overflow area in device memory (rarely reached: an entry
// is pushed only where the ray enters both children).
struct LdsStack {
    uint32_t* base;
    RT_DEV void put(int sp, uint32_t v) const { base[sp * kBlock] = v; }
    RT_DEV uint32_t get(int sp) const { return base[sp * kBlock]; }
};
// The two parts are addressed through pointers of their own address space: a pointer that may
// mean either compiles to flat loads, which are slower than LDS reads and make every earlier
// LDS store wait for completion first.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(1))) uint32_t glb_u32;
struct ShortStack {
    lds_u32* base;
    glb_u32* overflow;
    int depth;
    int stride;  // threads per workgroup
    RT_DEV void put(int sp, uint32_t v) const {
        if (sp < depth) base[sp * stride] = v; else overflow[sp - depth] = v;
    }
    RT_DEV uint32_t get(int sp) const {
        uint32_t v;
        if (sp < depth) v = base[sp * stride]; else v = overflow[sp - depth];
        return v;
    }
};

template <typename Stack>
RT_DEV void trav_pop(Trav& tv, const Stack& stack) {
    if (tv.sp == 0) {
        tv.cur = kTravDone;
    } else {
        tv.sp--;
        tv.cur = (int)stack.get(tv.sp);
    }
}

// Constants of the slab test for one ray: t = box * inv + off per axis.  The arithmetic
// runs on a direction with no zero component; the boxes are padded on the host by more
// than the origin perturbation the fused form implies (host/bvh_builder.cpp).  The box test only has to
// be conservative, so the hardware's approximate reciprocal (1 ulp) is enough here; the
// exact arithmetic is kept for the primitive tests.
struct Slab {
    float ix, iy, iz, ox, oy, oz;
};
RT_DEV Slab make_slab(V3 ro, V3 rd) {
    float dx = (g_abs(rd.x) < 1e-30f) ? __builtin_copysignf(1e-30f, rd.x) : rd.x;
    float dy = (g_abs(rd.y) < 1e-30f) ? __builtin_copysignf(1e-30f, rd.y) : rd.y;
    float dz = (g_abs(rd.z) < 1e-30f) ? __builtin_copysignf(1e-30f, rd.z) : rd.z;
    Slab s;
    s.ix = __builtin_amdgcn_rcpf(dx);
    s.iy = __builtin_amdgcn_rcpf(dy);
    s.iz = __builtin_amdgcn_rcpf(dz);
    s.ox = -(ro.x * s.ix);
    s.oy = -(ro.y * s.iy);
    s.oz = -(ro.z * s.iz);
    return s;
}

// Visit the internal node under the cursor: slab-test both children, descend into the
// nearer hit child, push the other.
template <typename Stack>
RT_DEV void trav_node_step(const float4* __restrict__ nodes, const Slab& sl, const Stack& stack, Trav& tv) {
    const float4* __restrict__ nd = table_row(nodes, tv.cur, 64u);
    const float4 q0 = nd[0];
    const float4 q1 = nd[1];
    const float4 q2 = nd[2];
    const float4 q3 = nd[3];
    // child 0: lo (q0.x q0.y q0.z) hi (q0.w q1.x q1.y); child 1: lo (q1.z q1.w q2.x) hi (q2.y q2.z q2.w)
    float a0 = __builtin_fmaf(q0.x, sl.ix, sl.ox), b0 = __builtin_fmaf(q0.w, sl.ix, sl.ox);
    float a1 = __builtin_fmaf(q0.y, sl.iy, sl.oy), b1 = __builtin_fmaf(q1.x, sl.iy, sl.oy);
    float a2 = __builtin_fmaf(q0.z, sl.iz, sl.oz), b2 = __builtin_fmaf(q1.y, sl.iz, sl.oz);
    float tmin0 = fmaxf(fmaxf(fminf(a0, b0), fminf(a1, b1)), fmaxf(fminf(a2, b2), 0.0f));
    float tmax0 = fminf(fminf(fmaxf(a0, b0), fmaxf(a1, b1)), fminf(fmaxf(a2, b2), tv.best_t));
    float c0 = __builtin_fmaf(q1.z, sl.ix, sl.ox), d0 = __builtin_fmaf(q2.y, sl.ix, sl.ox);
    float c1 = __builtin_fmaf(q1.w, sl.iy, sl.oy), d1 = __builtin_fmaf(q2.z, sl.iy, sl.oy);
    float c2 = __builtin_fmaf(q2.x, sl.iz, sl.oz), d2 = __builtin_fmaf(q2.w, sl.iz, sl.oz);
    float tmin1 = fmaxf(fmaxf(fminf(c0, d0), fminf(c1, d1)), fmaxf(fminf(c2, d2), 0.0f));
    float tmax1 = fminf(fminf(fmaxf(c0, d0), fmaxf(c1, d1)), fminf(fmaxf(c2, d2), tv.best_t));
    // q3 = the two children as cursor words (encode_child, done at upload) and their primitive
    // counts; an unused child slot (count -1, inverted box) must never be entered: the slab
    // arithmetic on its infinities would report a hit
    const bool hit0 = (tmin0 <= tmax0 * 1.000001f) && (__float_as_int(q3.z) >= 0);
    const bool hit1 = (tmin1 <= tmax1 * 1.000001f) && (__float_as_int(q3.w) >= 0);
    const int r0 = __float_as_int(q3.x);
    const int r1 = __float_as_int(q3.y);
    // selects around one push region and one pop region: nested branches cost more scalar
    // instructions per visit than the selects cost vector ones
    const bool both = hit0 && hit1;
    const bool second_nearer = both && (tmin1 < tmin0);
    const int near = (hit0 && !second_nearer) ? r0 : r1;
    if (both) {
        stack.put(tv.sp, (uint32_t)(second_nearer ? r0 : r1));
        tv.sp++;
    }
    if (hit0 || hit1) tv.cur = near;
    else trav_pop(tv, stack);
}

// The same for a four-wide node: slab-test the four children, descend into the nearest one hit, push the others,
// farthest first.  With centre and half extent the entry and exit distances need no min / max per axis:
// (c * inv + off) -+ h * |inv|.  An empty place has a negative half extent: its entry lies behind its exit.
// The boxes are padded on the host by 5e-6 of the scene's scale, five times what the rounding of these sums and of the
// reciprocal can amount to for rays that start inside the scene (host/bvh_builder.cpp).
// Two node formats (rt::WideNode, rt::WideNode16): HALF = false, 128 bytes - centre x, y, z and half extent x, y, z of the
// four children in binary32, their cursor words: seven 16-byte loads; HALF = true, 80 bytes back to back - a point of
// the node in binary32, centres relative to it and half extents in binary16, cursor words: five loads, three more
// multiply-adds (the node's point through the slab constants), the halves widened inside v_fma_mix_f32.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
RT_DEV float half_lo(float packed) { return (float)__builtin_bit_cast(half2_t, packed).x; }
RT_DEV float half_hi(float packed) { return (float)__builtin_bit_cast(half2_t, packed).y; }
struct Quad { float x, y, z, w; };
template <bool HALF, typename Stack>
RT_DEV void trav_wide_step(const float4* __restrict__ nodes, const Slab& sl, const Stack& stack, Trav& tv) {
    Quad cx, cy, cz, hx, hy, hz;
    float4 w;
    float ox, oy, oz;  // the slab offsets the centres are taken against
    if (HALF) {
        const float4* __restrict__ nd = table_row(nodes, tv.cur, 80u);
        const float4 org = nd[0], qa = nd[1], qb = nd[2], qc = nd[3];
        w = nd[4];
        ox = __builtin_fmaf(org.x, sl.ix, sl.ox); oy = __builtin_fmaf(org.y, sl.iy, sl.oy); oz = __builtin_fmaf(org.z, sl.iz, sl.oz);
        cx = {half_lo(qa.x), half_hi(qa.x), half_lo(qa.y), half_hi(qa.y)};
        cy = {half_lo(qa.z), half_hi(qa.z), half_lo(qa.w), half_hi(qa.w)};
        cz = {half_lo(qb.x), half_hi(qb.x), half_lo(qb.y), half_hi(qb.y)};
        hx = {half_lo(qb.z), half_hi(qb.z), half_lo(qb.w), half_hi(qb.w)};
        hy = {half_lo(qc.x), half_hi(qc.x), half_lo(qc.y), half_hi(qc.y)};
        hz = {half_lo(qc.z), half_hi(qc.z), half_lo(qc.w), half_hi(qc.w)};
    } else {
        const float4* __restrict__ nd = table_row(nodes, tv.cur, 128u);
        const float4 a = nd[0], b = nd[1], c = nd[2], d = nd[3], e = nd[4], f = nd[5];
        w = nd[6];
        cx = {a.x, a.y, a.z, a.w}; cy = {b.x, b.y, b.z, b.w}; cz = {c.x, c.y, c.z, c.w};
        hx = {d.x, d.y, d.z, d.w}; hy = {e.x, e.y, e.z, e.w}; hz = {f.x, f.y, f.z, f.w};
        ox = sl.ox; oy = sl.oy; oz = sl.oz;
    }
    float t0, t1, t2, t3;
    int c0 = __float_as_int(w.x), c1 = __float_as_int(w.y), c2 = __float_as_int(w.z), c3 = __float_as_int(w.w);
    const float ax = __builtin_fabsf(sl.ix), ay = __builtin_fabsf(sl.iy), az = __builtin_fabsf(sl.iz);
#define RT_WIDE_CHILD(T, F)                                                                                  \
    {                                                                                                         \
        const float bx = __builtin_fmaf(cx.F, sl.ix, ox), by = __builtin_fmaf(cy.F, sl.iy, oy);             \
        const float bz = __builtin_fmaf(cz.F, sl.iz, oz);                                                   \
        const float nx = __builtin_fmaf(-hx.F, ax, bx), ny = __builtin_fmaf(-hy.F, ay, by), nz = __builtin_fmaf(-hz.F, az, bz); \
        const float fx = __builtin_fmaf(hx.F, ax, bx), fy = __builtin_fmaf(hy.F, ay, by), fz = __builtin_fmaf(hz.F, az, bz);    \
        const float tmin = fmaxf(fmaxf(fmaxf(nx, ny), nz), 0.0f);                                           \
        const float tmax = fminf(fminf(fminf(fx, fy), fz), tv.best_t);                                      \
        T = (tmin <= tmax) ? tmin : kInf;                                                                    \
    }
    RT_WIDE_CHILD(t0, x)
    RT_WIDE_CHILD(t1, y)
    RT_WIDE_CHILD(t2, z)
    RT_WIDE_CHILD(t3, w)
#undef RT_WIDE_CHILD
    // nearest first (children not hit carry +inf and end up last: the k-th entry is a hit iff its distance is finite)
#define RT_WIDE_ORDER(TA, CA, TB, CB)                  \
    {                                                    \
        const bool swap = TB < TA;                       \
        const float tt = TA; const int cc = CA;          \
        TA = swap ? TB : TA; CA = swap ? CB : CA;        \
        TB = swap ? tt : TB; CB = swap ? cc : CB;        \
    }
    RT_WIDE_ORDER(t0, c0, t1, c1)
    RT_WIDE_ORDER(t2, c2, t3, c3)
    RT_WIDE_ORDER(t0, c0, t2, c2)
    RT_WIDE_ORDER(t1, c1, t3, c3)
    RT_WIDE_ORDER(t1, c1, t2, c2)
#undef RT_WIDE_ORDER
    if (t3 < kInf) { stack.put(tv.sp, (uint32_t)c3); tv.sp++; }
    if (t2 < kInf) { stack.put(tv.sp, (uint32_t)c2); tv.sp++; }
    if (t1 < kInf) { stack.put(tv.sp, (uint32_t)c1); tv.sp++; }
    if (t0 < kInf) tv.cur = c0;
    else trav_pop(tv, stack);
}

// Test the next primitive of the leaf under the cursor.
template <typename Stack>
RT_DEV void trav_leaf_step(const float4* __restrict__ prims, V3 ro, V3 rd, const Stack& stack, Trav& tv) {
    const int first = tv.cur & 0x0fffffff;
    const int left = (tv.cur >> 28) & 7;  // primitives after this one
    intersect_record(prims, first, ro, rd, tv.best_t, tv.best_prio, tv.best_idx);
    if (left == 0) trav_pop(tv, stack);
    else tv.cur = (int)(0x80000000u | ((uint32_t)(left - 1) << 28) | (uint32_t)(first + 1));
}

// Test the next two primitives of the leaf under the cursor (one, if it is the last): both records are
// fetched at once, so a leaf costs half the round trips to memory.
template <typename Stack>
RT_DEV void trav_leaf_step2(const float4* __restrict__ prims, V3 ro, V3 rd, const Stack& stack, Trav& tv) {
    const int first = tv.cur & 0x0fffffff;
    const int left = (tv.cur >> 28) & 7;  // primitives after this one
    const int second = first + (left > 0 ? 1 : 0);
    const float4* __restrict__ r0 = table_row(prims, first, 48u);
    const float4* __restrict__ r1 = table_row(prims, second, 48u);
    const float4 A0 = r0[0], B0 = r0[1], C0 = r0[2];
    const float4 A1 = r1[0], B1 = r1[1], C1 = r1[2];
    intersect_loaded(A0, B0, C0, first, ro, rd, tv.best_t, tv.best_prio, tv.best_idx);
    if (left > 0) intersect_loaded(A1, B1, C1, second, ro, rd, tv.best_t, tv.best_prio, tv.best_idx);
    if (left <= 1) trav_pop(tv, stack);
    else tv.cur = (int)(0x80000000u | ((uint32_t)(left - 2) << 28) | (uint32_t)(first + 2));
}

// Hit record of the winning primitive (raytracer.comp:354-358, 398-403, 454-461).
RT_DEV void make_hit(const DeviceScene& sc, const float4* __restrict__ recs, int idx, float t, V3 ro, V3 rd, Hit& h,
                     int& prim_id) {
    const float4* __restrict__ r = table_row(recs, idx, 48u);
    const float4 A = r[0];
    const float4 B = r[1];
    const float4 C = r[2];
    prim_id = __float_as_int(A.w);
    const int matbits = __float_as_int(C.w);
    h.t = t;
    h.p = ro + t * rd;
    V3 outward;
    if (matbits < 0) {
        h.mat = matbits & 0x7fffffff;
        outward = (h.p - v3(A.x, A.y, A.z)) / B.x;
    } else {
        h.mat = matbits;
        if (prim_id < sc.n_spheres + sc.n_triangles) {
            const float4 n = table_row(sc.triangles, prim_id - sc.n_spheres, 64u)[3];  // stored Triangle.normal
            outward = v3(n.x, n.y, n.z);
        } else {
            outward = normalize(cross(v3(B.x, B.y, B.z), v3(C.x, C.y, C.z)));
        }
    }
    h.front_face = dot(rd, outward) < 0.0f;  // set_face_normal, raytracer.comp:322-329
    h.normal = h.front_face ? outward : -outward;
}

RT_DEV uint32_t pack_unorm8(float x) {
    float c = g_clamp(x, 0.0f, 1.0f);  // NaN -> 0
    return (uint32_t)__builtin_rintf(c * 255.0f);
}

// raytracer.comp:984-995: accumulate one frame's colour, write acc/count as B,G,R,A bytes
RT_DEV uint32_t resolve_pixel(float4 a, int c) {
    float d = (float)((c > 1) ? c : 1);
    return pack_unorm8(a.z / d) | (pack_unorm8(a.y / d) << 8) | (pack_unorm8(a.x / d) << 16) | (pack_unorm8(a.w / d) << 24);
}

// The camera block (view_inv[16], cam_pos[3], tan_half_fov: the first 20 floats of FrameArgs) read from
// the kernel-argument segment where it is used, through a pointer the compiler cannot trace back: as a
// by-value argument it would sit in 20 scalar registers for the whole kernel, which the persistent
// kernels are short of (they spill scalars into vector lanes).  Only for kernels whose arguments begin
// with (DeviceScene, FrameArgs).
typedef const __attribute__((address_space(4))) float* KernargFloats;
static_assert(offsetof(FrameArgs, view_inv) == 0 && offsetof(FrameArgs, cam_pos) == 64 && offsetof(FrameArgs, tan_half_fov) == 76,
              "the camera block must open FrameArgs");
RT_DEV KernargFloats kernarg_camera() {
    const __attribute__((address_space(4))) char* p = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (KernargFloats)(p + ((sizeof(DeviceScene) + alignof(FrameArgs) - 1) & ~(alignof(FrameArgs) - 1)));
}
// The same for the whole of FrameArgs: a copy whose fields are loaded where this is called (only the
// fields the caller goes on to use), for the parts of a kernel that run once per pixel.
RT_DEV FrameArgs kernarg_frame() {
#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((address_space(4))) char* p = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const __attribute__((address_space(4))) FrameArgs*)(p + ((sizeof(DeviceScene) + alignof(FrameArgs) - 1) & ~(alignof(FrameArgs) - 1)));
#else
    return FrameArgs{};  // the host pass only parses device functions
#endif
}

// FrameArgs::frame_counts[frame] read from the kernel-argument segment (indexing a by-value copy of the block by a
// lane's frame would put the whole copy into scratch memory).  Same kernels as kernarg_camera.
RT_DEV uint32_t kernarg_frame_count(uint32_t frame) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((address_space(4))) char* p = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    p += ((sizeof(DeviceScene) + alignof(FrameArgs) - 1) & ~(alignof(FrameArgs) - 1)) + offsetof(FrameArgs, frame_counts);
    return ((const __attribute__((address_space(4))) uint32_t*)p)[frame];
#else
    return frame;
#endif
}

// get_ray, raytracer.comp:933-952: jittered primary ray through pixel (px, py); `cam` = the camera block.
template <typename CamPtr>
RT_DEV void generate_ray(const FrameArgs& fa, CamPtr cam, int px, int py, uint32_t& seed, V3& ro, V3& rd) {
    const float aspect = (float)fa.width / (float)fa.height;  // raytracer.comp:187
    float sx = rng_random(seed) - 0.5f;
    float sy = rng_random(seed) - 0.5f;
    float fx = (float)px + sx;
    float fy = (float)py + sy;
    float ndcX = 2.0f * fx / (float)fa.width - 1.0f;
    float ndcY = 1.0f - 2.0f * fy / (float)fa.height;
    const float tan_half_fov = cam[19];
    V3 dc = normalize(v3(ndcX * aspect * tan_half_fov, ndcY * tan_half_fov, -1.0f));
    const CamPtr m = cam;  // view_inv[16], cam_pos[3], tan_half_fov: the first 20 floats of FrameArgs
    V3 dw = v3(m[0] * dc.x + m[4] * dc.y + m[8] * dc.z + m[12] * 0.0f,
               m[1] * dc.x + m[5] * dc.y + m[9] * dc.z + m[13] * 0.0f,
               m[2] * dc.x + m[6] * dc.y + m[10] * dc.z + m[14] * 0.0f);
    rd = normalize(dw);
    ro = v3(cam[16], cam[17], cam[18]);
}

// Pixel slot -> image coordinates: slots walk 8x8 tiles row-major over the rows this
// context owns; with a strip partition the owned rows are every strip_count-th strip.
RT_DEV bool slot_to_pixel(const FrameArgs& fa, uint32_t slot, int& x, int& y) {
    uint32_t block = slot >> kBlockSlotsLog2;
    const uint32_t in = slot & ((1u << kBlockSlotsLog2) - 1u);
    if (fa.tile_order != nullptr) block = fa.tile_order[block];
    // inside a block: 8x8 tiles in row-major order, 64 consecutive slots per tile
    const uint32_t tile = in >> 6, p = in & 63u;
    const uint32_t tmask = (1u << kBlockTilesLog2) - 1u;
    x = (int)(((block % (uint32_t)fa.tiles_x) << kBlockPixLog2) + ((tile & tmask) << 3) + (p & 7u));
    const int r = (int)(((block / (uint32_t)fa.tiles_x) << kBlockPixLog2) + ((tile >> kBlockTilesLog2) << 3) + (p >> 3));
    if (x >= fa.width || r >= fa.owned_rows) return false;
    y = r;
    if (fa.strip_rows > 0) y = ((r / fa.strip_rows) * fa.strip_count + fa.strip_index) * fa.strip_rows + (r % fa.strip_rows);
    return true;
}

// The same for a launch of several frames (FrameArgs::n_frames): slot `slot` of XCD `xcd`, frame-major; the XCD's
// blocks are those at positions xcd, xcd + n_xcds, ... of the order.  `frame` and the image block are returned
// also for a slot outside the image (false), which the caller reports as finished.  `exhausted`: the XCD has
// no such slot.
RT_DEV bool batch_slot_to_pixel(const FrameArgs& fa, uint32_t xcd, uint32_t slot, int& x, int& y, uint32_t& frame, uint32_t& block,
                                bool& exhausted) {
    const uint32_t n_blocks = fa.total_slots >> kBlockSlotsLog2;
    const uint32_t nx = (uint32_t)fa.n_xcds;
    const uint32_t mine = n_blocks > xcd ? (n_blocks - xcd + nx - 1u) / nx : 0u;
    const uint32_t per_frame = mine << kBlockSlotsLog2;
    exhausted = per_frame == 0u || slot >= per_frame * (uint32_t)fa.n_frames;
    frame = 0; block = 0;
    if (exhausted) return false;
    frame = slot / per_frame;
    const uint32_t r = slot - frame * per_frame;
    const uint32_t pos = xcd + (r >> kBlockSlotsLog2) * nx;
    block = fa.tile_order != nullptr ? fa.tile_order[pos] : pos;
    const uint32_t in = r & ((1u << kBlockSlotsLog2) - 1u);
    const uint32_t tile = in >> 6, p = in & 63u;
    const uint32_t tmask = (1u << kBlockTilesLog2) - 1u;
    x = (int)(((block % (uint32_t)fa.tiles_x) << kBlockPixLog2) + ((tile & tmask) << 3) + (p & 7u));
    const int rr = (int)(((block / (uint32_t)fa.tiles_x) << kBlockPixLog2) + ((tile >> kBlockTilesLog2) << 3) + (p >> 3));
    if (x >= fa.width || rr >= fa.owned_rows) return false;
    y = rr;
    if (fa.strip_rows > 0) y = ((rr / fa.strip_rows) * fa.strip_count + fa.strip_index) * fa.strip_rows + (rr % fa.strip_rows);
    return true;
}
RT_DEV uint32_t xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }  // hwreg XCC_ID, bits 3:0
// Image block of pixel (x, y), as report_pixel_cost and FrameArgs::block_done index it, and how many of the block's
// pixel slots lie inside the image (the rows this context owns).
RT_DEV uint32_t block_of_pixel(const FrameArgs& fa, int x, int y, uint32_t& pixels_inside) {
    int r = y;
    if (fa.strip_rows > 0) r = ((y / fa.strip_rows) / fa.strip_count) * fa.strip_rows + (y % fa.strip_rows);
    const int side = 1 << kBlockPixLog2;
    const int bx = x >> kBlockPixLog2, by = r >> kBlockPixLog2;
    const int w = fa.width - bx * side, h = fa.owned_rows - by * side;
    pixels_inside = (uint32_t)((w < side ? w : side) * (h < side ? h : side));
    return (uint32_t)by * (uint32_t)fa.tiles_x + (uint32_t)bx;
}
// Accesses that are performed in the XCD's L2 (atomics never hit the L1; workgroup scope keeps them out of the
// memory-side path): how a pixel's frames, rendered by different CUs of one XCD, see each other's framebuffer values.
// (The load is an atomic add of a zero the compiler cannot see through: it turns an atomic OR / ADD of a literal 0 into a
// plain load, which may hit the L1 - for ever, when the value is being waited for.)
RT_DEV uint32_t l2_load(const uint32_t* p) {
    uint32_t zero = 0u;
    asm volatile("" : "+v"(zero));
    return __hip_atomic_fetch_add((uint32_t*)p, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
RT_DEV uint32_t l2_add(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// A finished pixel reports what it cost (path segments over all its samples) to its image block,
// which keeps the largest: a block is as urgent as its longest pixel, because the frame cannot
// end before a pixel that started late has walked its whole chain of samples (ordering blocks by
// the sum of their pixels' costs instead left C3 2 % and the 1 M-triangle scene 7 % slower).
RT_DEV void report_pixel_cost(const FrameArgs& fa, int x, int y, uint32_t cost) {
    if (fa.tile_cost == nullptr) return;
    int r = y;
    if (fa.strip_rows > 0) r = ((y / fa.strip_rows) / fa.strip_count) * fa.strip_rows + (y % fa.strip_rows);
    atomicMax(&fa.tile_cost[(uint32_t)(r >> kBlockPixLog2) * (uint32_t)fa.tiles_x + (uint32_t)(x >> kBlockPixLog2)], cost);
}

// The second half of sample_light (raytracer.comp:617-626, 636-644, 662-671): what arrives
// from light `ltype` (colour/strength lc) given the visibility search's result.
RT_DEV void light_arrival(int ltype, float4 lc, int hit_idx, float hit_t, float s_d, V3 s_L, V3 normal, V3& Le, V3& Ldir,
                          float& light_pdf) {
    Le = v3(0.0f);
    Ldir = normal;
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
}

// direct_light, raytracer.comp:870-881.  With no emission arriving (Le == 0: occluded,
// facing away, no lights) the term is clamp(0 * fr * cos / pdf) = +0 whatever fr is (a NaN
// clamps to 0 too), and the attenuation at bounce 0 is 1: nothing to add.
RT_DEV V3 direct_term(const Material& mat, const Hit& h, V3 V, V3 Le, V3 Ldir, float light_pdf, V3 att) {
    float cos_theta = g_max(0.0f, dot(h.normal, Ldir));
    float mat_pdf;
    V3 fr = eval_mat(mat, Ldir, V, h, mat_pdf);
    float pdf = power_heuristics(light_pdf, mat_pdf);
    return g_clamp(Le * fr * cos_theta / pdf, 0.0f, 1.0f) * att;
}

// The first half of sample_light (raytracer.comp:582-616, 633-635, 653-661): pick a light
// by strength, sample a point on it.  Returns the kind of query that follows:
//   0 = visibility search along s_L (distance s_d), 1 = no search, arrival known (AMBIENT:
//   Ldir in s_L), 2 = no search, nothing arrives (DIRECTIONAL facing away), 3 = no search,
//   the shader's undefined light types (Le = 1, L = 0, pdf = 0).
//   `to_light` receives the vector s_L and s_d derive from (s_d = length, s_L = normalize of it; for a
//   directional light s_L itself).
RT_DEV int light_pick(const DeviceScene& sc, const FrameArgs& fa, const Hit& h, uint32_t& seed, int& light, V3& s_L, float& s_d,
                      V3& to_light) {
    float rand_strength = rng_random(seed) * fa.lights_strength_sum;
    int low = 0, high = fa.total_lights - 1;
    while (low < high) {
        int mid = (low + high) / 2;
        if (rand_strength <= table_row(sc.lights, mid, 48u)[2].y) high = mid; else low = mid + 1;
    }
    light = low;
    const float4 lp = table_row(sc.lights, low, 48u)[0];
    const int ltype = __float_as_int(table_row(sc.lights, low, 48u)[2].x);
    s_d = 0.0f;
    s_L = v3(0.0f);
    to_light = v3(0.0f);
    if (ltype == 0 /* AMBIENT */) {
        s_L = random_vec_on_hemisphere(h.normal, seed);
        return 1;
    }
    if (ltype == 1 /* SPHERE */) {
        V3 center = v3(lp.x, lp.y, lp.z);
        V3 center_to_point = h.p - center;
        V3 sphere_point = random_vec_on_hemisphere(normalize(center_to_point), seed) * lp.w + center;
        V3 pts = sphere_point - h.p;
        s_d = length(pts);
        s_L = normalize(pts);
        to_light = pts;
        return 0;
    }
    if (ltype == 3 /* DIRECTIONAL */) {
        V3 light_dir = -v3(lp.x, lp.y, lp.z);
        if (dot(h.normal, light_dir) > 0.0f) {
            s_L = light_dir;
            to_light = light_dir;
            return 0;
        }
        return 2;
    }
    if (ltype == 6 /* TRIANGLE */) {
        const int ti = (int)lp.x;
        const float4* __restrict__ tri = table_row(sc.triangles, ti, 64u);
        const float4 t0 = tri[0];
        const float4 t1 = tri[1];
        const float4 t2 = tri[2];
        float e1 = g_sqrt(rng_random(seed));
        float e2 = rng_random(seed);
        V3 tri_point = (1.0f - e1) * v3(t0.x, t0.y, t0.z) + e1 * (1.0f - e2) * v3(t1.x, t1.y, t1.z) + e1 * e2 * v3(t2.x, t2.y, t2.z);
        V3 pts = tri_point - h.p;
        s_d = length(pts);
        s_L = normalize(pts);
        to_light = pts;
        return 0;
    }
    // POINT / CONE / AREA fall through the shader's switch (raytracer.comp:628-651) and
    // return vec3(1) with L and pdf never written; they are taken as zero.
    return 3;
}

}  // namespace rtd
