// rt_pool.hip — the pooled form of the path tracer for gfx950 (MI355X): one persistent
// launch per frame in which every workgroup owns a pool of kPool path slots (1344 for its 1024 lanes),
// keeps their hot state in LDS and regroups them continuously through queues in LDS.
//
// One frame = one execution of shaders/raytracer.comp::main for every pixel this context
// owns (raytracer.comp:956-996).  A slot carries one pixel: its samples and bounces are
// computed strictly one after the other with the pixel's own RNG state, by the same
// per-path functions as the other pipelines (rt_stages.h), so results do not depend on the
// scheduling below (the tests compare bit patterns).
//
// Why: with one path per lane (rt_kernels.hip) a wave's 64 paths disagree about what to do
// next, and most vector instructions run with most lanes masked off; with one launch per
// stage a 1080p frame has too few rays per launch to fill this chip.
// Here a workgroup keeps kPool paths in flight for its lanes.  Slots that wait for the
// same kind of work queue up in LDS (search, first hit, later hit, miss, the light's term,
// sample end, pixel end); a wave either
//   * shades: pops 64 slots of ONE kind and runs that stage for all of them, or
//   * searches: its lanes walk the tree for 64 rays in bursts of steps - each step the wave votes for a
//     visit of a (four-wide or binary) node or a test of the next two primitives of a leaf -, and a lane
//     whose ray has ended stores the result, queues the slot for its stage and pops the next waiting ray.
// Hot path state (RNG, pixel, counters, ray, attenuation, sample radiance, hash, hit: 72 B
// per slot) lives in LDS, structure of arrays; what is touched once per sample or per
// light query (pixel sum, light-query record: 64 B per slot) lives in device memory in a
// region private to the workgroup.  Hand-offs between the workgroup's waves need
// workgroup-scope ordering only: state stores, release fence, LDS queue entry; the popping
// wave reads the entry, acquire fence, state loads.  No wave ever waits for another wave
// except for the few instructions between a queue reservation and its entry store.
// Traversal stacks keep kShortStack entries per lane in LDS; deeper entries (rare) go to
// a lane-private overflow area in device memory.
#include "rt_stages.h"

namespace rtd {

namespace {

#ifndef RT_POOL_THREADS
#define RT_POOL_THREADS 1024
#endif
#ifndef RT_POOL_SLOTS
#define RT_POOL_SLOTS 1344
#endif
#ifndef RT_POOL_STACK
#define RT_POOL_STACK 8
#endif
#ifndef RT_POOL_WAVES_PER_SIMD
#define RT_POOL_WAVES_PER_SIMD 4
#endif
constexpr int kThreads = RT_POOL_THREADS;  // threads per workgroup: one workgroup of 16 waves per CU (128 VGPRs, 4 waves per SIMD)
constexpr int kWaves = kThreads / 64;
constexpr int kPool = RT_POOL_SLOTS;    // path slots per workgroup (multiple of kWaves, <= 4096)
constexpr int kRing = kPool <= 256 ? 256 : kPool <= 512 ? 512 : kPool <= 1024 ? 1024 : kPool <= 2048 ? 2048 : 4096;  // queue ring size
constexpr int kSearchRing = 2 * kRing;  // the search queue's ring: a slot can have two rays waiting (its path's and its light query's)
constexpr int kShortStack = RT_POOL_STACK;  // traversal stack entries per lane kept in LDS
constexpr int kOverflow = kStackDepth - kShortStack;  // entries per lane of the overflow area (a stack never holds more than kStackDepth)
enum { kLive = 7 };  // PoolLds::hc[kLive]: the live count and the abort bit
enum { Q_SEARCH = 0, Q_FIRST = 1, Q_NEXT = 2, Q_END = 3, Q_LIT = 4, Q_PIXEL = 5, Q_MISS = 6, kQueues = 7 };
constexpr int kJobPriority = 2;         // s_setprio level of a wave inside a shading job (0 elsewhere)
constexpr uint32_t kBacklog = 128;      // entries in a shading queue from which a searching wave helps the shading waves out
constexpr int kShaderHold = 512;        // polls for which a shading wave leaves the pixel-end queue alone after a pixel had to wait
constexpr uint32_t kChunk = 256;        // pixel slots a workgroup reserves per global atomic
constexpr uint16_t kEmpty = 0xffffu;
constexpr uint32_t kSpinLimit = 1u << 22;
constexpr uint32_t kGuardLimit = 1u << 22;  // consecutive outer iterations without any work

struct PoolLds {
    // hot path state, one array per field
    uint32_t seed[kPool], pixel[kPool], cnt[kPool];  // cnt = sample << 16 | bounce
    float ro[3][kPool], rd[3][kPool], att[3][kPool], col[3][kPool];
    uint32_t phash[kPool];
    float hit_t[kPool];
    int32_t hit_idx[kPool];
    uint32_t flags[(kPool + 3) / 4];  // one byte per slot: LF_* of the sample's light query (changed by LDS atomics on the word)
    // queues
    uint16_t items[(kQueues + 1) * kRing];  // rings (ring_entry: the search queue's first, twice the size); an entry is a slot index | kind << 12, kEmpty when free
    // What every wave polls once per round, in one 32-byte block (two 16-byte LDS reads):
    // hc[q] = next ring position to pop << 16 | entries published and not yet claimed (one word, so
    // that a pop claims entries and advances the position with a single compare-and-swap), for the kQueues queues;
    // hc[7] = slots that still carry a pixel | bit 31: a bound on waiting was exceeded (never in a correct run).
    alignas(16) uint32_t hc[8];
    uint32_t tail[kQueues];          // next ring position to push
    uint32_t res_lock, res_next, res_end;  // the workgroup's reserve of pixel slots of the frame (taken kChunk at a time)
};
static_assert(kQueues == 7 && kRing <= 65536 && 65536 % kRing == 0, "hc packs a 16-bit ring position and a 16-bit count");
static_assert(kPool % kWaves == 0 && kPool <= 4096 && kPool <= kRing && 2 * kPool <= kSearchRing,
              "a queue entry is a 12-bit slot index plus a kind, a ring holds every slot and the search queue's every ray there can be");
// Where position `pos` of queue q's ring lies in PoolLds::items.
RT_DEV uint32_t ring_entry(int q, uint32_t pos) {
    return q == Q_SEARCH ? (pos & (uint32_t)(kSearchRing - 1)) : (uint32_t)(q + 1) * (uint32_t)kRing + (pos & (uint32_t)(kRing - 1));
}
static_assert(kThreads % 64 == 0 && kThreads <= 1024, "whole waves, one workgroup");

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
struct PoolCtl {
    uint32_t count[kQueues], live, abort;
};

// The cold records are written once and read once, far apart in time: with RT_COLD_NT their accesses carry the
// non-temporal hint (1: stores, 2: stores and loads), so that they do not displace the scene in the L2.
#ifndef RT_COLD_NT
#define RT_COLD_NT 0
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
RT_DEV float4 cold_load(const float4* p) {
#if RT_COLD_NT >= 2
    const f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
RT_DEV void cold_store(float4* p, float4 v) {
#if RT_COLD_NT >= 1
    f32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, (f32x4*)p);
#else
    *p = v;
#endif
}

// Slot i's cold record in the workgroup's region (a 32-bit offset from a uniform base: see table_row).
RT_DEV float4* cold_of(float4* base, uint32_t i) { return (float4*)((char*)base + i * 64u); }

// Accessor of slot i: hot fields in LDS, cold record (pixel sum, pixel cost | the light's term: normal,
// light distance | incoming direction, packed bits | light direction, hit id; 64 bytes) in device memory.
struct PoolSlot {
    PoolLds& L;
    float4* cold;
    uint32_t i;
    RT_DEV void get_ctl(uint32_t& seed, uint32_t& pixel, uint32_t& sample, int& bounce) const {
        seed = L.seed[i]; pixel = L.pixel[i];
        const uint32_t c = L.cnt[i];
        sample = c >> 16; bounce = (int)(c & 0xffffu);
    }
    RT_DEV void put_ctl(uint32_t seed, uint32_t pixel, uint32_t sample, int bounce) const {
        L.seed[i] = seed; L.pixel[i] = pixel; L.cnt[i] = (sample << 16) | (uint32_t)bounce;
    }
    RT_DEV V3 get_att(uint32_t& phash) const { phash = L.phash[i]; return v3(L.att[0][i], L.att[1][i], L.att[2][i]); }
    RT_DEV void put_att(V3 a, uint32_t phash) const { L.att[0][i] = a.x; L.att[1][i] = a.y; L.att[2][i] = a.z; L.phash[i] = phash; }
    RT_DEV V3 get_ro() const { return v3(L.ro[0][i], L.ro[1][i], L.ro[2][i]); }
    RT_DEV V3 get_rd() const { return v3(L.rd[0][i], L.rd[1][i], L.rd[2][i]); }
    RT_DEV void put_ro(V3 v) const { L.ro[0][i] = v.x; L.ro[1][i] = v.y; L.ro[2][i] = v.z; }
    RT_DEV void put_rd(V3 v) const { L.rd[0][i] = v.x; L.rd[1][i] = v.y; L.rd[2][i] = v.z; }
    RT_DEV void get_hit(float& t, int& idx) const { t = L.hit_t[i]; idx = L.hit_idx[i]; }
    RT_DEV V3 get_col() const { return v3(L.col[0][i], L.col[1][i], L.col[2][i]); }
    RT_DEV void put_col(V3 v) const { L.col[0][i] = v.x; L.col[1][i] = v.y; L.col[2][i] = v.z; }
    RT_DEV KernargFloats camera(const FrameArgs&) const { return kernarg_camera(); }  // pool_kernel(DeviceScene, FrameArgs, ...)
    RT_DEV FrameArgs late_frame(const FrameArgs&) const { return kernarg_frame(); }
    RT_DEV uint32_t frame_count_of(uint32_t frame) const { return kernarg_frame_count(frame); }
    RT_DEV V3 get_pix(float& cost) const { const float4 v = cold_load(cold + 0); cost = v.w; return xyz(v); }
    RT_DEV void put_pix(V3 v, float cost) const { cold_store(cold + 0, f4(v, cost)); }
    RT_DEV void put_hash(uint32_t phash) const { L.phash[i] = phash; }
    // What the light's term of the sample needs at its end (LightRecord), 44 bytes of the cold record:
    // normal, light distance | incoming direction, front face << 31 | light << 16 | material | light
    // direction, the query's hit id (hash mode).  (The host offers this kernel only to scenes with at most
    // 32768 lights and 65536 materials.)
    RT_DEV void put_light_record(const LightRecord& q) const {
        cold_store(cold + 1, f4(q.normal, q.s_d));
        cold_store(cold + 2, f4(q.vdir, __uint_as_float((q.front_face ? 0x80000000u : 0u) | ((uint32_t)q.light << 16) | (uint32_t)q.mat)));
#if defined(RT_TAIL_DIAG)
        cold_store(cold + 3, f4(q.ldir, cold[3].w));  // (the word keeps the pixel's start time)
#else
        cold_store(cold + 3, f4(q.ldir, 0.0f));
#endif
    }
    RT_DEV void get_light_record(LightRecord& q) const {
        const float4 n = cold_load(cold + 1), v = cold_load(cold + 2), l = cold_load(cold + 3);
        const uint32_t bits = __float_as_uint(v.w);
        q.normal = xyz(n); q.s_d = n.w;
        q.vdir = xyz(v); q.light = (int)((bits >> 16) & 0x7fffu);
        q.mat = (int)(bits & 0xffffu); q.front_face = (bits >> 31) != 0u;
        q.ldir = xyz(l);
    }
#if defined(RT_TAIL_DIAG)
    RT_DEV void put_start(uint32_t t) const { ((uint32_t*)(cold + 3))[3] = t; }
    RT_DEV uint32_t get_start() const { return ((const uint32_t*)(cold + 3))[3]; }
#endif
    RT_DEV int get_light_hit_id() const { return __float_as_int(cold[3].w); }  // written when the per-pixel hash is on
    RT_DEV uint32_t get_flags() const { return (*(const volatile __attribute__((address_space(3))) uint32_t*)&L.flags[i >> 2] >> (8u * (i & 3u))) & 0xffu; }
    RT_DEV void put_flags(uint32_t f) const { *(volatile __attribute__((address_space(3))) uint8_t*)((uint8_t*)L.flags + i) = (uint8_t)f; }
    RT_DEV uint32_t or_flags(uint32_t bits) const { return (atomicOr(&L.flags[i >> 2], bits << (8u * (i & 3u))) >> (8u * (i & 3u))) & 0xffu; }
};

// Volatile accesses to the workgroup's LDS block go through LDS-typed pointers (a volatile
// access through a generic pointer compiles to a flat load).
typedef __attribute__((address_space(3))) uint16_t lds_u16;
RT_DEV uint32_t lds_peek(const uint32_t* p) { return *(const volatile lds_u32*)p; }
RT_DEV void lds_poke(uint32_t* p, uint32_t v) { *(volatile lds_u32*)p = v; }

// One coherent look at the queues' fill, the live count and the abort flag.
RT_DEV PoolCtl poll_ctl(const PoolLds& L) {
    const u32x4 a = *(const volatile lds_u32x4*)&L.hc[0];
    const u32x4 b = *(const volatile lds_u32x4*)&L.hc[4];
    PoolCtl c;
    c.count[0] = a.x & 0xffffu; c.count[1] = a.y & 0xffffu; c.count[2] = a.z & 0xffffu; c.count[3] = a.w & 0xffffu;
    c.count[4] = b.x & 0xffffu; c.count[5] = b.y & 0xffffu; c.count[6] = b.z & 0xffffu;
    c.live = b.w & 0x7fffffffu;
    c.abort = b.w >> 31;
    return c;
}

// Publish `item` for the lanes that ask.  Everything the wave stored before (path state in
// LDS and device memory) is visible to the workgroup before an entry can be read: state stores,
// release fence, entry store, count.
RT_DEV void queue_push(PoolLds& L, int q, bool want, uint32_t item) {
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return;
    const int leader = __ffsll((long long)m) - 1;
    const uint32_t n = (uint32_t)__popcll(m);
    uint32_t pos = 0;
    if (lane_id() == leader) pos = atomicAdd(&L.tail[q], n);
    pos = (uint32_t)__builtin_amdgcn_readlane((int)pos, leader);
    // Release BEFORE the entry store: entries are claimed by ring position, so another wave's count
    // may expose this position, and its popper takes the entry the moment it stops being kEmpty.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (want) *(volatile lds_u16*)&L.items[ring_entry(q, pos + (uint32_t)__popcll(m & lanes_below()))] = (uint16_t)item;
    if (lane_id() == leader) atomicAdd(&L.hc[q], n);  // LDS operations of one wave complete in order: after the entries
}

// The same for several queues at once: every lane names its queue (0..kQueues-1, or -1 for
// nothing) and its item; lane k reserves for queue k, so the wave pays one LDS atomic round trip
// however many queues receive entries.  MASK: the queues this call site can name (the others cost nothing).
template <unsigned MASK>
RT_DEV void queue_push_any(PoolLds& L, int q, uint32_t item) {
    unsigned long long m[kQueues];
    unsigned long long any = 0ull;
#pragma unroll
    for (int k = 0; k < kQueues; ++k) {
        m[k] = (MASK >> k) & 1u ? __ballot(q == k) : 0ull;
        any |= m[k];
    }
    if (any == 0ull) return;
    const int lane = lane_id();
    uint32_t n_mine = 0;
#pragma unroll
    for (int k = 0; k < kQueues; ++k)
        if ((MASK >> k) & 1u) n_mine = (lane == k) ? (uint32_t)__popcll(m[k]) : n_mine;
    uint32_t pos = 0;
    if (lane < kQueues && n_mine != 0u) pos = atomicAdd(&L.tail[lane], n_mine);
    uint32_t base = 0;
    unsigned long long mk = 0ull;
#pragma unroll
    for (int k = 0; k < kQueues; ++k) {
        if (!((MASK >> k) & 1u)) continue;
        const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)pos, k);
        if (q == k) { base = pk; mk = m[k]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // before the entry store, as in queue_push
    if (q >= 0) {
        const uint32_t at = base + (uint32_t)__popcll(mk & lanes_below());
        // (a call site that cannot name the search queue needs no look at which ring it is)
        const uint32_t idx = ((MASK >> Q_SEARCH) & 1u) ? ring_entry(q, at) : (uint32_t)(q + 1) * (uint32_t)kRing + (at & (uint32_t)(kRing - 1));
        *(volatile lds_u16*)&L.items[idx] = (uint16_t)item;
    }
    if (lane < kQueues && n_mine != 0u) atomicAdd(&L.hc[lane], n_mine);
}

// Claim up to one entry for each lane of `takers` (in lane order), but nothing if fewer than
// `at_least` are there.  Returns the number of entries claimed; `got` tells a lane whether
// `item` is valid.
RT_DEV uint32_t queue_pop(PoolLds& L, int q, unsigned long long takers, uint32_t at_least, uint32_t& item, bool& got) {
    got = false;
    const uint32_t want = (uint32_t)__popcll(takers);
    uint32_t n = 0, pos = 0;
    if (want != 0u && lane_id() == 0) {
        uint32_t w = lds_peek(&L.hc[q]);
        for (;;) {
            const uint32_t c = w & 0xffffu;
            if (c == 0u || c < at_least) break;
            const uint32_t take = c < want ? c : want;
            const uint32_t old = atomicCAS(&L.hc[q], w, (((w >> 16) + take) << 16) | (c - take));
            if (old == w) { n = take; pos = w >> 16; break; }
            w = old;
        }
    }
    n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
    pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
    if (n == 0u) return 0u;
    const uint32_t rank = (uint32_t)__popcll(takers & lanes_below());
    if (((takers >> lane_id()) & 1ull) && rank < n) {
        volatile lds_u16* e = (volatile lds_u16*)&L.items[ring_entry(q, pos + rank)];
        uint32_t v = *e, tries = 0;
        while (v == kEmpty && ++tries < kSpinLimit) v = *e;  // the pusher is between its reservation and its store
        if (v == kEmpty) atomicOr(&L.hc[kLive], 0x80000000u);
        *e = kEmpty;
        item = v;
        got = (v != kEmpty);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return n;
}

// Give the asking lanes' slots their next pixel (the caller queues them for the search).  Pixel slots
// of the frame come from the workgroup's reserve, refilled kChunk at a time from the frame's
// counter (one global atomic per kChunk pixels; the reserve is handed out under a small LDS
// lock, taken a few hundred times per frame).  A slot that finds the frame exhausted dies.
template <bool BATCH>
RT_DEV bool take_pixels(PoolLds& L, const FrameArgs& fa_hot, float4* cold_base, bool need, uint32_t local) {
    if (__ballot(need) == 0ull) return false;
    const FrameArgs fa = kernarg_frame();  // once per pixel: not worth scalar registers for the whole kernel
    // a launch of several frames hands out the slots of this XCD (frame-major), a single frame those of the image
    const bool batch = BATCH;
    const uint32_t xcd = batch ? xcc_id() : 0u;
    uint32_t* const counter = batch ? fa.xcd_next + 32u * xcd : fa.counters;
    bool started = false, dead = false;
    for (;;) {
        const unsigned long long m = __ballot(need);
        if (m == 0ull) break;
        const int leader = __ffsll((long long)m) - 1;
        const uint32_t n = (uint32_t)__popcll(m);
        uint32_t b1 = 0, c1 = 0, b2 = 0;
        if (lane_id() == leader) {
            while (atomicCAS(&L.res_lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint32_t next = lds_peek(&L.res_next), end = lds_peek(&L.res_end);
            const uint32_t avail = end - next;
            b1 = next;
            c1 = avail < n ? avail : n;
            next += c1;
            if (c1 < n) {
                b2 = atomicAdd(counter, kChunk);
                next = b2 + (n - c1);
                end = b2 + kChunk;
            }
            lds_poke(&L.res_next, next);
            lds_poke(&L.res_end, end);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            atomicExch(&L.res_lock, 0u);
        }
        b1 = (uint32_t)__builtin_amdgcn_readlane((int)b1, leader);
        c1 = (uint32_t)__builtin_amdgcn_readlane((int)c1, leader);
        b2 = (uint32_t)__builtin_amdgcn_readlane((int)b2, leader);
        if (need) {
            const uint32_t rank = (uint32_t)__popcll(m & lanes_below());
            const uint32_t mine = rank < c1 ? b1 + rank : b2 + (rank - c1);
            int x, y;
            if (batch) {
                uint32_t frame, block;
                bool exhausted;
                if (batch_slot_to_pixel(fa, xcd, mine, x, y, frame, block, exhausted)) {
                    start_pixel<BATCH>(fa, PoolSlot{L, cold_of(cold_base, local), local}, x, y, frame);
                    need = false;
                    started = true;
                } else if (exhausted) {
                    need = false;
                    dead = true;
                }
                // else: a slot of a ragged edge block outside the image; ask again
            } else if (mine >= fa.total_slots) {
                need = false;
                dead = true;
            } else if (slot_to_pixel(fa, mine, x, y)) {
                start_pixel<BATCH>(fa, PoolSlot{L, cold_of(cold_base, local), local}, x, y);
                need = false;
                started = true;
            }
            // else: a slot of a ragged edge tile outside the image; ask again
        }
    }
    const unsigned long long d = __ballot(dead);
    if (d != 0ull && lane_id() == __ffsll((long long)d) - 1) atomicSub(&L.hc[kLive], (uint32_t)__popcll(d));
    return started;  // the caller queues these slots for the search
}

// NSW: the number of shading waves as a compile-time constant (the instances the host's own choices run on: 1 % on a
// single frame of the teapot scene), or 0 for the kernel argument.
template <int ACCEL, bool STATS, bool BATCH, int NSW = 0>
__global__ void __launch_bounds__(kThreads, RT_POOL_WAVES_PER_SIMD) pool_kernel(const DeviceScene sc, const FrameArgs fa, const PathPool pp, int refill_min,
                                                         int burst, int job_min, int slots_cap, int shader_waves_arg) {
    const int shader_waves = NSW > 0 ? NSW : shader_waves_arg;
    __shared__ PoolLds L;
    extern __shared__ uint32_t s_stack[];  // [kShortStack][kThreads]
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    float4* const cold_base = pp.cold + (size_t)blockIdx.x * kPool * 4;
    const float4* __restrict__ recs = (ACCEL >= 1) ? sc.prims_bvh : sc.prims_flat;
    const ShortStack stack{(lds_u32*)(s_stack + tid), (glb_u32*)(pp.stack_overflow + ((size_t)blockIdx.x * kThreads + (size_t)tid) * pp.overflow_words), kShortStack, kThreads};

    for (int i = tid; i < (kQueues + 1) * kRing; i += kThreads) L.items[i] = kEmpty;
    if (tid < 8) L.hc[tid] = 0;
    if (tid < kQueues) L.tail[tid] = 0;
    // A frame with fewer pixels than the pools have slots is shared out evenly: every workgroup fills
    // only its share of slots (the rest stay dead), so that all CUs get work.
    // `slots_cap` (a multiple of the waves, at most kPool) limits the paths in flight further: a scene that
    // overflows the caches runs faster with fewer rays competing for them.
    const uint32_t share = (fa.total_slots * (uint32_t)(BATCH ? fa.n_frames : 1) + gridDim.x - 1u) / gridDim.x;
    const uint32_t per_wave = share >= (uint32_t)slots_cap ? (uint32_t)slots_cap / (uint32_t)kWaves : (share + kWaves - 1u) / kWaves;
    if (tid == 0) { L.hc[kLive] = per_wave * kWaves; L.res_lock = 0; L.res_next = 0; L.res_end = 0; }
    __syncthreads();
    for (uint32_t g = 0; g * 64u < per_wave; ++g) {
        const uint32_t mine = (uint32_t)wave * (uint32_t)(kPool / kWaves) + g * 64u + (uint32_t)lane;
        queue_push(L, Q_SEARCH, take_pixels<BATCH>(L, fa, cold_base, g * 64u + (uint32_t)lane < per_wave, mine), mine);
    }

    // the ray this lane walks the BVH for
    int slot = -1;   // slot index, -1: none
    int skind = 0;   // queue its result goes to
    V3 ro = v3(0.0f), rd = v3(0.0f, 0.0f, 1.0f);
    Slab sl = make_slab(ro, rd);
    Trav tv;
    trav_reset(tv, kTravDone);
    float s_d = -kInf;     // a light query's distance to the sampled light point (+inf: directional light; -inf: a path ray)
    const bool hash_mode = (fa.flags & 2u) != 0u;
    const float occl_bias = hash_mode ? kInf : 0.2f;
    bool search_first = false;  // skip the look for shading work once
    int pixel_hold = 0;         // rounds for which this wave leaves the pixel-end queue alone (see there)
    // Where a sample goes when path and light query have both ended: to the light's term if the light is
    // visible (`flags` as they were when the later of the two ended), else to the pixel's end after its last
    // sample, else to the next sample.
    const uint32_t last_sample = (uint32_t)kernarg_frame().spp - 1u;
    auto end_queue = [&](const PoolSlot& ps, uint32_t flags) -> int {
        if (flags & LF_LIT) return Q_LIT;
        return (L.cnt[ps.i] >> 16) >= last_sample ? Q_PIXEL : Q_END;
    };
    uint32_t n_rays = 0, n_nodes = 0, n_tests = 0, n_leaf_steps = 0, guard = 0;
    unsigned long long wave_iters = 0, busy_iters = 0;  // STATS: search steps of this wave; lanes that had a ray at those steps
    uint32_t search_peak = 0;  // STATS: the most rays this wave saw waiting in the search queue
    unsigned long long deep_pushes = 0;  // STATS: pushes onto a stack already holding >= 4 (<< 32), >= 6 (<< 16), >= 8 entries
    uint32_t overflow_steps = 0;         // STATS: node visits that left entries in the lane's overflow area in device memory
    unsigned long long job_stats[4] = {0, 0, 0, 0};  // STATS: per shading queue, items << 32 | jobs
    unsigned long long job_cycles[4] = {0, 0, 0, 0}, search_cycles = 0, step_cycles = 0;  // STATS: wave cycles (step_cycles: the part of search_cycles inside the burst of steps)
    const unsigned long long t_begin = STATS ? __builtin_readcyclecounter() : 0ull;

    for (;;) {
        const PoolCtl ctl = poll_ctl(L);
        if (ctl.abort != 0u) break;
        if (STATS) search_peak = ctl.count[Q_SEARCH] > search_peak ? ctl.count[Q_SEARCH] : search_peak;

        // ---- a full wave of shading work of one kind?  (or any at all, when this wave has nothing to search)
        int q = Q_NEXT;
        uint32_t best = ctl.count[Q_NEXT];
        if (ctl.count[Q_FIRST] > best) { best = ctl.count[Q_FIRST]; q = Q_FIRST; }
        if (ctl.count[Q_END] > best) { best = ctl.count[Q_END]; q = Q_END; }
        if (ctl.count[Q_LIT] > best) { best = ctl.count[Q_LIT]; q = Q_LIT; }
        if (ctl.count[Q_PIXEL] > best && (!BATCH || pixel_hold == 0)) { best = ctl.count[Q_PIXEL]; q = Q_PIXEL; }
        if (BATCH && pixel_hold > 0) pixel_hold--;
        if (ctl.count[Q_MISS] > best) { best = ctl.count[Q_MISS]; q = Q_MISS; }
        const unsigned long long busy = __ballot(slot >= 0);
        const bool nothing_to_search = (busy == 0ull) && ctl.count[Q_SEARCH] == 0u;
        // a job is worth a wave when it is full; when the pool runs thin (end of the frame, or fewer
        // pixels than slots) a sixteenth of the live paths must do, or they would wait for each other
        uint32_t job_thresh = ctl.live >> 4;
        job_thresh = job_thresh < 1u ? 1u : (job_thresh > (uint32_t)job_min ? (uint32_t)job_min : job_thresh);
        // The workgroup's first `shader_waves` waves only shade: they take every full job the moment it is there (a
        // searching wave comes by once per burst of steps, and while it shades the rays in its lanes stand still), hold
        // no rays, and sleep between looks.  The other waves search; they shade what piles up (a queue of kBacklog
        // entries) and, as always, whatever there is when they have nothing to search, so that jobs smaller than a
        // wave are still done by whoever runs out of rays.  How many: about the share of the wave cycles that is
        // shading (a quarter on the scenes that fit the caches, a tenth on the 1.07 M-triangle scene) times the
        // workgroup's waves; the host chooses (pool_shader_waves in rt_api.hip).
        const bool shader_wave = wave < shader_waves;
        const bool take_job = shader_wave ? best >= job_thresh
                                          : ((best >= job_thresh && (shader_waves == 0 || best >= kBacklog)) || (nothing_to_search && best != 0u));
        if (!search_first && take_job) {
            uint32_t item = 0;
            bool got;
            const unsigned long long t_job = STATS ? __builtin_readcyclecounter() : 0ull;
            // A shading job runs at raised wave priority: its 64 slots come back as rays as soon as it ends, and the
            // searching waves it shares the SIMD with are the ones waiting for rays (C3 7.02 -> 6.93 ms, C2 2.31 ->
            // 2.26, 4K frames 8.02 -> 7.91, 1.07 M triangles 11.72 -> 11.63; levels 1, 2 and 3 alike; raising the
            // hand-out of rays as well, or it alone, loses what this gains).
            __builtin_amdgcn_s_setprio(kJobPriority);
            const uint32_t n_items = queue_pop(L, q, ~0ull, best >= job_thresh ? job_thresh : 1u, item, got);
            if (n_items == 0u) {
                __builtin_amdgcn_s_setprio(0);
                continue;
            }
            guard = 0;
            const int jk = q == Q_FIRST ? 0 : q == Q_NEXT ? 1 : q == Q_LIT ? 2 : 3;  // STATS index: first-bounce, later-bounce, light's term, sample / pixel end / misses
            if (STATS) job_stats[jk] += ((unsigned long long)n_items << 32) | 1ull;
            const uint32_t i = item & 0xfffu;
            const PoolSlot ps{L, cold_of(cold_base, i), i};
            if (q == Q_PIXEL) {
                // (in a launch of several frames a pixel may have to wait for its previous frame's store: back into the queue)
                const bool ended = got && end_pixel_item<BATCH>(fa, ps);
                const bool started = take_pixels<BATCH>(L, fa, cold_base, ended, i);
                queue_push(L, Q_SEARCH, started, i);  // kind 0: the next pixel's first ray
                if (BATCH) {
                    queue_push(L, Q_PIXEL, got && !ended, i);
                    // the pixels that wait must not keep this wave from the other queues (the fullest queue is served
                    // first, and a queue of waiting pixels stays full): it looks away from this one for a while
                    // (a round of a shading wave is a poll and a short sleep, not a burst of search steps: it stays away for longer)
                    if (__ballot(got && !ended) != 0ull) { search_first = true; pixel_hold = shader_wave ? kShaderHold : 16; }
                }
            } else if (q == Q_END) {
                if (got) end_sample_item(fa, ps);
                queue_push(L, Q_SEARCH, got, i);  // kind 0: the next sample's first ray
            } else if (q == Q_LIT) {
                if (got) lit_item(sc, ps);
                queue_push_any<(1u << Q_END) | (1u << Q_PIXEL)>(L, got ? end_queue(ps, 0u) : -1, i);
            } else {
                bool ends = false, retry = false;
                int light_kind = 0;
                // (a ray that missed everything comes through its own queue, so that the hits' stages run full: its
                // entry says whether it was a sample's first ray)
                if (got) ends = shade_item(sc, fa, ps, recs, q == Q_MISS ? (int)(item >> 12) : q - Q_FIRST, light_kind, retry);
                // a first hit may send a light query on its way beside the path
                if (q == Q_FIRST) queue_push(L, Q_SEARCH, got && light_kind != 0, i | ((uint32_t)light_kind << 12));
                // the path: its next ray, or its end - where the sample waits for its light query if that is still
                // out (whoever of the two comes last queues the sample's end)
                int to = -1;
                uint32_t entry = i;
                if (got && retry) {
                    to = q; entry = item;  // back where it came from, as it came
                } else if (got && !ends) {
                    to = Q_SEARCH; entry = i | (1u << 12);
                } else if (got) {
                    uint32_t f = ps.get_flags();
                    bool last = true;
                    if (f & LF_PENDING) { f = ps.or_flags(LF_PATH_DONE); last = (f & LF_DONE) != 0u; }
                    if (last) to = end_queue(ps, f);
                }
                queue_push_any<(1u << Q_SEARCH) | (1u << Q_NEXT) | (1u << Q_MISS) | (1u << Q_END) | (1u << Q_LIT) | (1u << Q_PIXEL)>(L, to, entry);
                // items put back (they wait for their light query to be taken by a lane or, with the per-pixel hash,
                // to report its hit id) must not keep the waves from the searching that would end the wait
                search_first = __ballot(got && retry) != 0ull;
            }
            if (STATS) job_cycles[jk] += __builtin_readcyclecounter() - t_job;
            __builtin_amdgcn_s_setprio(0);
            continue;
        }
        search_first = false;
        if (shader_wave) {  // nothing to shade right now
            if ((lds_peek(&L.hc[kLive]) & 0x7fffffffu) == 0u) break;
            if (++guard > kGuardLimit) { atomicOr(&L.hc[kLive], 0x80000000u); break; }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }

        // ---- search: retire the lanes whose ray has ended, hand them the next waiting rays
        const unsigned long long t_search = STATS ? __builtin_readcyclecounter() : 0ull;
        const unsigned long long ended = __ballot(tv.cur == kTravDone && slot >= 0);
        if (ended != 0ull) {
            const bool mine = (tv.cur == kTravDone && slot >= 0);
            int to = -1;
            uint32_t entry = (uint32_t)slot;
            if (mine && skind < 2) {  // a path ray: its hit goes to the stage of its kind, a miss to the misses' stage
                L.hit_t[slot] = tv.best_t;
                L.hit_idx[slot] = tv.best_idx;
                to = tv.best_idx < 0 ? Q_MISS : Q_FIRST + skind;
                entry = (uint32_t)slot | (tv.best_idx < 0 ? (uint32_t)skind << 12 : 0u);
            }
            if (hash_mode && mine && skind >= 2)  // the query's hit id, for the per-pixel hash
                ((int*)(cold_of(cold_base, (uint32_t)slot) + 3))[3] = tv.best_idx >= 0 ? __float_as_int(table_row(recs, tv.best_idx, 48u)[0].w) : -1;
            if (hash_mode) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (mine && skind >= 2) {
                // a light query: the light is visible if the nearest hit lies within 0.1 of the sampled point
                // (raytracer.comp:617-626, 662-671), a directional one if nothing is hit (636-644).  The
                // sample's end is queued by whichever of query and path finishes last.
                const bool visible = skind == 3 ? tv.best_idx < 0 : (tv.best_idx >= 0 && g_abs(s_d - tv.best_t) < 0.1f);
                const PoolSlot ps{L, cold_of(cold_base, (uint32_t)slot), (uint32_t)slot};
                if (ps.or_flags(LF_DONE | (visible ? LF_LIT : 0u)) & LF_PATH_DONE) to = end_queue(ps, visible ? LF_LIT : 0u);
            }
            queue_push_any<(1u << Q_FIRST) | (1u << Q_NEXT) | (1u << Q_MISS) | (1u << Q_END) | (1u << Q_LIT) | (1u << Q_PIXEL)>(L, to, entry);
            if (mine) slot = -1;
        }
        const unsigned long long idle = __ballot(slot < 0);
        if (idle != 0ull && (__popcll(idle) >= refill_min || idle == ~0ull)) {
            uint32_t item = 0;
            bool got;
            if (queue_pop(L, Q_SEARCH, idle, 1u, item, got) != 0u && got) {
                slot = (int)(item & 0xfffu);
                skind = (int)(item >> 12);
                {
                    ro = v3(L.ro[0][slot], L.ro[1][slot], L.ro[2][slot]);
                    rd = v3(L.rd[0][slot], L.rd[1][slot], L.rd[2][slot]);
                    s_d = -kInf;
                }
                if (skind >= 2) {
                    // a light query: from the shading point (the slot's ro) along the vector to the sampled light
                    // point, which its first hit left in the slot's sample radiance; distance and direction are
                    // the length and the normalisation the shader takes of it (raytracer.comp:612-614, 657-659),
                    // a directional light's direction is used as it is (635)
                    const V3 to_light = v3(L.col[0][slot], L.col[1][slot], L.col[2][slot]);
                    s_d = skind == 3 ? kInf : length(to_light);
                    rd = skind == 3 ? to_light : normalize(to_light);
                    L.col[0][slot] = 0.0f; L.col[1][slot] = 0.0f; L.col[2][slot] = 0.0f;  // the sample's radiance again
                    (void)PoolSlot{L, cold_of(cold_base, (uint32_t)slot), (uint32_t)slot}.or_flags(LF_FETCHED);
                }
                n_rays++;
                if (ACCEL >= 1) {
                    sl = make_slab(ro, rd);
                    trav_reset(tv, 0);  // node 0 is always an internal node
                    // A light query only asks whether the nearest hit lies within 0.1 of the sampled point's
                    // distance s_d, or, for a directional light, whether anything is hit at all: its search is
                    // bounded to t < s_d + 0.2 and ends at the first hit nearer than s_d - 0.2 (any hit for a
                    // directional light).  Not when the per-pixel hash of nearest-hit ids is recorded, which
                    // needs the true nearest hit (occl_bias is infinite then).
                    if (skind == 2) tv.best_t = s_d + occl_bias;
                } else {
                    trav_reset(tv, kTravDone);
                    trace_flat(sc, ro, rd, tv.best_t, tv.best_idx);
                    if (STATS) n_tests += (uint32_t)sc.n_prims;
                }
            }
        }
        if (__ballot(slot >= 0) == 0ull) {
            if ((lds_peek(&L.hc[kLive]) & 0x7fffffffu) == 0u) break;
            if (++guard > kGuardLimit) { atomicOr(&L.hc[kLive], 0x80000000u); break; }
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        guard = 0;
        const unsigned long long t_steps = STATS ? __builtin_readcyclecounter() : 0ull;
        if (ACCEL >= 1) {
            for (int it = 0; it < burst; ++it) {
                const unsigned long long m_node = __ballot(tv.cur >= 0 && tv.cur != kTravDone);
                const unsigned long long m_leaf = __ballot(tv.cur < 0);
                if ((m_node | m_leaf) == 0ull) break;
                if (STATS) { wave_iters++; busy_iters += (unsigned long long)__popcll(m_node | m_leaf); }
                if ((int)__popcll(m_node) >= (int)__popcll(m_leaf)) {
                    if (tv.cur >= 0 && tv.cur != kTravDone) {
                        const int sp_before = tv.sp;
                        if (ACCEL == 3) trav_wide_step<true>(sc.nodes, sl, stack, tv);
                        else if (ACCEL == 2) trav_wide_step<false>(sc.nodes, sl, stack, tv);
                        else trav_node_step(sc.nodes, sl, stack, tv);
                        if (STATS) n_nodes++;
                        if (STATS && tv.sp > sp_before && tv.sp > kShortStack) overflow_steps++;
                        if (STATS && tv.sp > sp_before) deep_pushes += (sp_before >= 4 ? (1ull << 32) : 0ull) + (sp_before >= 8 ? 1ull : 0ull) + (sp_before >= 6 ? (1ull << 16) : 0ull);
                    }
                } else {
                    if (tv.cur < 0) {
                        if (STATS) { n_leaf_steps++; n_tests += ((tv.cur >> 28) & 7) > 0 ? 2u : 1u; }
                        trav_leaf_step2(sc.prims_bvh, ro, rd, stack, tv);
                        if (tv.best_t < s_d - occl_bias) tv.cur = kTravDone;  // a light query is settled (never true for path rays)
                    }
                }
            }
        }
        if (STATS) { const unsigned long long now = __builtin_readcyclecounter(); search_cycles += now - t_search; step_cycles += now - t_steps; }
    }

    // ---- counters: one atomic per wave
    unsigned long long r = n_rays, nn = n_nodes, nt = n_tests, nl = n_leaf_steps, ov = overflow_steps;
    unsigned long long dp4 = deep_pushes >> 32, dp6 = (deep_pushes >> 16) & 0xffffu, dp8 = deep_pushes & 0xffffu;
    for (int off = 32; off > 0; off >>= 1) {
        if (STATS) { dp4 += __shfl_down(dp4, off); dp6 += __shfl_down(dp6, off); dp8 += __shfl_down(dp8, off); }
        r += __shfl_down(r, off);
        if (STATS) {
            nn += __shfl_down(nn, off);
            nt += __shfl_down(nt, off);
            nl += __shfl_down(nl, off);
            ov += __shfl_down(ov, off);
        }
    }
    if (lane == 0) {
        atomicAdd((unsigned long long*)(fa.counters + 2), r);
        atomicAdd((unsigned long long*)(fa.counters + 8), r);
        if (STATS) {
            atomicAdd((unsigned long long*)(fa.counters + 4), nn);
            atomicAdd((unsigned long long*)(fa.counters + 6), nt);
            atomicAdd((unsigned long long*)(fa.counters + 16), nn + nl);
            atomicAdd((unsigned long long*)(fa.counters + 18), wave_iters);
            atomicAdd((unsigned long long*)(fa.counters + 40), busy_iters);
            for (int k = 0; k < 4; ++k) atomicAdd((unsigned long long*)(fa.counters + 20 + 2 * k), job_stats[k]);
            atomicAdd((unsigned long long*)(fa.counters + 28), (dp4 << 40) | (dp6 << 20) | dp8);
            const unsigned long long shade_cycles = job_cycles[0] + job_cycles[1] + job_cycles[2] + job_cycles[3];
            atomicAdd((unsigned long long*)(fa.counters + 10), __builtin_readcyclecounter() - t_begin - shade_cycles - search_cycles);
            atomicAdd((unsigned long long*)(fa.counters + 12), search_cycles);
            atomicAdd((unsigned long long*)(fa.counters + 42), step_cycles);
            atomicAdd((unsigned long long*)(fa.counters + 46), ov);
            atomicMax(fa.counters + 44, search_peak);
            fa.counters[45] = (uint32_t)kSearchRing;
            atomicAdd((unsigned long long*)(fa.counters + 14), shade_cycles);
            for (int k = 0; k < 4; ++k) atomicAdd((unsigned long long*)(fa.counters + 32 + 2 * k), job_cycles[k]);
        }
        if ((lds_peek(&L.hc[kLive]) >> 31) != 0u) *(volatile uint32_t*)kernarg_frame().abort_flag = 1u;
    }
}

typedef void (*PoolKernel)(DeviceScene, FrameArgs, PathPool, int, int, int, int, int);
template <bool BATCH>
PoolKernel pick_kernel_of(int accel, bool stats) {
    if (accel == 3) return stats ? pool_kernel<3, true, BATCH> : pool_kernel<3, false, BATCH>;
    if (accel == 2) return stats ? pool_kernel<2, true, BATCH> : pool_kernel<2, false, BATCH>;
    return accel == 1 ? (stats ? pool_kernel<1, true, BATCH> : pool_kernel<1, false, BATCH>) : (stats ? pool_kernel<0, true, BATCH> : pool_kernel<0, false, BATCH>);
}
PoolKernel pick_kernel(int accel, bool stats, bool batch = false, int shader_waves = 0) {
    if (accel == 3 && !stats) {  // the same for the 80-byte nodes
        if (!batch && shader_waves == 3) return pool_kernel<3, false, false, 3>;
        if (!batch && shader_waves == 2) return pool_kernel<3, false, false, 2>;
        if (batch && shader_waves == 4) return pool_kernel<3, false, true, 4>;
        if (batch && shader_waves == 2) return pool_kernel<3, false, true, 2>;
    }
    if (accel == 2 && !stats) {  // the instances with the shading waves compiled in
        if (!batch && shader_waves == 3) return pool_kernel<2, false, false, 3>;
        if (!batch && shader_waves == 2) return pool_kernel<2, false, false, 2>;
        if (batch && shader_waves == 4) return pool_kernel<2, false, true, 4>;
        if (batch && shader_waves == 2) return pool_kernel<2, false, true, 2>;
    }
    return batch ? pick_kernel_of<true>(accel, stats) : pick_kernel_of<false>(accel, stats);
}
constexpr size_t kStackLds = (size_t)kShortStack * kThreads * sizeof(uint32_t);

}  // namespace

int pool_slots_per_block() { return kPool; }
int pool_threads_per_block() { return kThreads; }
int pool_overflow_per_lane() { return kOverflow; }
int pool_overflow_for(int stack_need) { return stack_need + 1 - kShortStack > kOverflow ? stack_need + 1 - kShortStack : kOverflow; }

// Workgroups of the pooled launch: as many as are resident at once, fewer when the image
// is small.  The path pool must hold grid * pool_slots_per_block() cold records and
// grid * pool_threads_per_block() * pool_overflow_per_lane() stack words.
hipError_t pool_grid(const FrameArgs& fa, int accel, bool stats, bool batch, int shader_waves, int cu_count, int per_cu_cap, unsigned* grid) {
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pick_kernel(accel, stats, batch, shader_waves), kThreads, kStackLds);
    if (e != hipSuccess) return e;
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
    if (per_cu_cap >= 1 && per_cu_cap < per_cu) per_cu = per_cu_cap;
    // every CU takes part as long as a workgroup gets a few waves' worth of pixels
    const long long want = ((long long)fa.total_slots + kChunk - 1) / kChunk, cap = (long long)cu_count * per_cu;
    *grid = (unsigned)(want < cap ? want : cap);
    return hipSuccess;
}

hipError_t launch_pool(const DeviceScene& sc, const FrameArgs& fa, const PathPool& pp, int accel, bool stats, unsigned grid,
                       int refill_min, int burst, int job_min, int slots_cap, int shader_waves, hipStream_t stream) {
    if (grid == 0) return hipSuccess;
    shader_waves = shader_waves < 0 ? 0 : (shader_waves > kWaves / 2 ? kWaves / 2 : shader_waves);  // searching waves there must be
    slots_cap = (slots_cap <= 0 || slots_cap > kPool) ? kPool : (slots_cap < kWaves ? kWaves : slots_cap / kWaves * kWaves);
    hipLaunchKernelGGL(pick_kernel(accel, stats, fa.n_frames > 1, shader_waves), dim3(grid), dim3(kThreads), kStackLds, stream, sc, fa, pp, refill_min, burst, job_min, slots_cap, shader_waves);
    return hipGetLastError();
}

}  // namespace rtd
