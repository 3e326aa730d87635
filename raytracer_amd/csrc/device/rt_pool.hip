// rt_pool.hip — the pooled form of the path tracer for gfx950 (MI355X): one persistent
// launch per frame in which every workgroup owns a pool of kPool path slots (four per lane)
// and regroups them continuously through queues in LDS.
//
// One frame = one execution of shaders/raytracer.comp::main for every pixel this context
// owns (raytracer.comp:956-996).  A slot carries one pixel: its samples and bounces are
// computed strictly one after the other with the pixel's own RNG state, by the same
// per-path functions as the other pipelines (rt_stages.h), so results do not depend on the
// scheduling below (the tests compare bit patterns).
//
// Why: with one path per lane (rt_kernels.hip) a wave's 64 paths disagree about what to do
// next, and most vector instructions run with most lanes masked off; with one launch per
// stage (rt_wavefront.hip) a 1080p frame has too few rays per launch to fill this chip.
// Here a workgroup keeps 1024 paths in flight for its 256 lanes.  Slots that wait for the
// same kind of work queue up in LDS (search, first-bounce result, later-bounce result,
// light-query result, sample end); a wave either
//   * shades: pops 64 slots of ONE kind and runs that stage for all of them, or
//   * searches: its lanes walk the BVH for 64 rays, and a lane whose ray has ended stores
//     the result, queues the slot for its stage and pops the next waiting ray.
// Path state lives in device memory in a region private to the workgroup (PathPool,
// slot = blockIdx.x * kPool + local index), so hand-offs between the workgroup's waves need
// workgroup-scope ordering only: state stores, release fence, LDS queue entry; the popping
// wave reads the entry, acquire fence, state loads.  No wave ever waits for another wave
// except for the few instructions between a queue reservation and its entry store.
#include "rt_stages.h"

#include <cstdlib>

namespace rtd {

namespace {

#ifndef RT_POOL_SLOTS
#define RT_POOL_SLOTS 1024
#endif
constexpr int kPool = RT_POOL_SLOTS;  // path slots per workgroup
enum { Q_SEARCH = 0, Q_FIRST = 1, Q_NEXT = 2, Q_LIGHT = 3, Q_END = 4, kQueues = 5 };
constexpr uint16_t kEmpty = 0xffffu;
constexpr uint32_t kSpinLimit = 1u << 22;
constexpr uint32_t kGuardLimit = 1u << 22;  // consecutive outer iterations without any work

struct PoolLds {
    uint16_t items[kQueues][kPool];  // rings; an entry is a local slot index (| kind << 12 in Q_SEARCH), kEmpty when free
    uint32_t count[kQueues];         // entries published and not yet claimed
    uint32_t head[kQueues];          // next ring position to pop
    uint32_t tail[kQueues];          // next ring position to push
    uint32_t live;                   // slots that still carry a pixel
    uint32_t abort;                  // set when a bound on waiting was exceeded (never in a correct run)
};

RT_DEV uint32_t lds_peek(const uint32_t* p) { return *(const volatile uint32_t*)p; }

// Publish `item` for the lanes that ask.  Everything the wave stored before (path state in
// device memory) is visible to the workgroup before the entries can be claimed.
RT_DEV void queue_push(PoolLds& L, int q, bool want, uint32_t item) {
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return;
    const int leader = __ffsll((long long)m) - 1;
    const uint32_t n = (uint32_t)__popcll(m);
    uint32_t pos = 0;
    if (lane_id() == leader) pos = atomicAdd(&L.tail[q], n);
    pos = (uint32_t)__builtin_amdgcn_readlane((int)pos, leader);
    if (want) *(volatile uint16_t*)&L.items[q][(pos + (uint32_t)__popcll(m & lanes_below())) & (kPool - 1)] = (uint16_t)item;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane_id() == leader) atomicAdd(&L.count[q], n);
}

// Claim up to one entry for each lane of `takers` (in lane order).  Returns the number of
// entries claimed; `got` tells a lane whether `item` is valid.
RT_DEV uint32_t queue_pop(PoolLds& L, int q, unsigned long long takers, uint32_t& item, bool& got) {
    got = false;
    const uint32_t want = (uint32_t)__popcll(takers);
    uint32_t n = 0, pos = 0;
    if (want != 0u && lane_id() == 0) {
        uint32_t c = lds_peek(&L.count[q]);
        while (c != 0u) {
            const uint32_t take = c < want ? c : want;
            const uint32_t old = atomicCAS(&L.count[q], c, c - take);
            if (old == c) { n = take; break; }
            c = old;
        }
        if (n != 0u) pos = atomicAdd(&L.head[q], n);
    }
    n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
    pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
    if (n == 0u) return 0u;
    const uint32_t rank = (uint32_t)__popcll(takers & lanes_below());
    if (((takers >> lane_id()) & 1ull) && rank < n) {
        volatile uint16_t* e = &L.items[q][(pos + rank) & (kPool - 1)];
        uint32_t v = *e, tries = 0;
        while (v == kEmpty && ++tries < kSpinLimit) v = *e;  // the pusher is between its reservation and its store
        if (v == kEmpty) atomicExch(&L.abort, 1u);
        *e = kEmpty;
        item = v;
        got = (v != kEmpty);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return n;
}

// Give the asking lanes' slots their next pixel from the frame's pixel counter (seed:
// raytracer.comp:959-961; first primary ray: get_ray, raytracer.comp:933-952) and queue
// them for the search.  A slot that finds the frame exhausted dies.
RT_DEV void take_pixels(PoolLds& L, const FrameArgs& fa, const PathPool& pp, uint32_t slot_base, bool need, uint32_t local) {
    bool started = false, dead = false;
    for (;;) {
        const unsigned long long m = __ballot(need);
        if (m == 0ull) break;
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if (lane_id() == leader) base = atomicAdd(&fa.counters[0], (uint32_t)__popcll(m));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        if (need) {
            const uint32_t mine = base + (uint32_t)__popcll(m & lanes_below());
            int x, y;
            if (mine >= fa.total_slots) {
                need = false;
                dead = true;
            } else if (slot_to_pixel(fa, mine, x, y)) {
                const uint32_t s = slot_base + local;
                uint32_t seed = rng_hash(fa.time_u * 1920u) ^ rng_hash(fa.frame_count) ^ rng_hash((uint32_t)(x + y * 1920));
                V3 ro, rd;
                generate_ray(fa, x, y, seed, ro, rd);
                pp.ctl[s] = make_uint4(seed, (uint32_t)(y * fa.width + x), 0u, 0u);
                pp.ray_o[s] = f4(ro, 0.0f);
                pp.ray_d[s] = f4(rd, 0.0f);
                pp.att[s] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
                pp.col[s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                pp.pix[s] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                pp.stage[s] = WF_PATH_FIRST;
                need = false;
                started = true;
            }
            // else: a slot of a ragged edge tile outside the image; ask again
        }
    }
    queue_push(L, Q_SEARCH, started, local);
    const unsigned long long d = __ballot(dead);
    if (d != 0ull && lane_id() == __ffsll((long long)d) - 1) atomicSub(&L.live, (uint32_t)__popcll(d));
}

template <int ACCEL, bool STATS>
__global__ void __launch_bounds__(kBlock, 4) pool_kernel(const DeviceScene sc, const FrameArgs fa, const PathPool pp, int refill_min,
                                                         int burst) {
    __shared__ PoolLds L;
    extern __shared__ uint32_t s_stack[];  // [stack_depth][kBlock]
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t slot_base = blockIdx.x * (uint32_t)kPool;
    const float4* __restrict__ recs = (ACCEL == 1) ? sc.prims_bvh : sc.prims_flat;
    uint32_t* const stack = s_stack + tid;

    for (int i = tid; i < kQueues * kPool; i += kBlock) (&L.items[0][0])[i] = kEmpty;
    if (tid < kQueues) { L.count[tid] = 0; L.head[tid] = 0; L.tail[tid] = 0; }
    if (tid == 0) { L.live = kPool; L.abort = 0; }
    __syncthreads();
    for (int g = 0; g < kPool / kBlock; ++g) take_pixels(L, fa, pp, slot_base, true, (uint32_t)(wave * (kPool / 4) + g * 64 + lane));

    // the ray this lane walks the BVH for
    int slot = -1;   // local slot index, -1: none
    int skind = 0;   // queue its result goes to
    V3 ro = v3(0.0f), rd = v3(0.0f, 0.0f, 1.0f);
    Slab sl = make_slab(ro, rd);
    Trav tv;
    trav_reset(tv, kTravDone);
    uint32_t n_rays = 0, n_nodes = 0, n_tests = 0, guard = 0;
    unsigned long long wave_iters = 0;

    for (;;) {
        if (lds_peek(&L.abort) != 0u) break;

        // ---- a full wave of shading work of one kind?  (or any at all, when this wave has nothing to search)
        const uint32_t c_first = lds_peek(&L.count[Q_FIRST]), c_next = lds_peek(&L.count[Q_NEXT]);
        const uint32_t c_light = lds_peek(&L.count[Q_LIGHT]), c_end = lds_peek(&L.count[Q_END]);
        int q = Q_LIGHT;
        uint32_t best = c_light;
        if (c_next > best) { best = c_next; q = Q_NEXT; }
        if (c_first > best) { best = c_first; q = Q_FIRST; }
        if (c_end > best) { best = c_end; q = Q_END; }
        const unsigned long long busy = __ballot(tv.cur != kTravDone || slot >= 0);
        const bool nothing_to_search = (busy == 0ull) && lds_peek(&L.count[Q_SEARCH]) == 0u;
        if (best >= 64u || (nothing_to_search && best != 0u)) {
            uint32_t item = 0;
            bool got;
            if (queue_pop(L, q, ~0ull, item, got) == 0u) continue;
            guard = 0;
            if (q == Q_END) {
                bool alive = false;
                if (got) end_item(fa, pp, slot_base + item, alive);
                queue_push(L, Q_SEARCH, got && alive, item);  // WF_PATH_FIRST
                take_pixels(L, fa, pp, slot_base, got && !alive, item);
            } else {
                bool ends = false;
                uint32_t next_stage = WF_IDLE;
                if (got) ends = shade_item(sc, fa, pp, recs, q - Q_FIRST, slot_base + item, next_stage);
                const bool go = got && !ends;
                queue_push(L, Q_END, got && ends, item);
                queue_push(L, Q_SEARCH, go && next_stage == WF_PATH_NEXT, item | (1u << 12));
                queue_push(L, Q_SEARCH, go && next_stage == WF_LIGHT_SEARCH, item | (2u << 12));
                queue_push(L, Q_LIGHT, go && next_stage >= WF_LIGHT_AMBIENT && next_stage <= WF_LIGHT_UNDEF, item);
            }
            continue;
        }

        // ---- search: retire the lanes whose ray has ended, hand them the next waiting rays
        const unsigned long long ended = __ballot(tv.cur == kTravDone && slot >= 0);
        if (ended != 0ull) {
            const bool mine = (tv.cur == kTravDone && slot >= 0);
            if (mine) pp.hit[slot_base + (uint32_t)slot] = make_uint2(__float_as_uint(tv.best_t), (uint32_t)tv.best_idx);
            queue_push(L, Q_FIRST, mine && skind == 0, (uint32_t)slot);
            queue_push(L, Q_NEXT, mine && skind == 1, (uint32_t)slot);
            queue_push(L, Q_LIGHT, mine && skind == 2, (uint32_t)slot);
            if (mine) slot = -1;
        }
        const unsigned long long idle = __ballot(slot < 0);
        if (idle != 0ull && (__popcll(idle) >= refill_min || idle == ~0ull)) {
            uint32_t item = 0;
            bool got;
            if (queue_pop(L, Q_SEARCH, idle, item, got) != 0u && got) {
                slot = (int)(item & 0xfffu);
                skind = (int)(item >> 12);
                const uint32_t s = slot_base + (uint32_t)slot;
                ro = xyz(pp.ray_o[s]);
                rd = xyz(pp.ray_d[s]);
                n_rays++;
                if (ACCEL == 1) {
                    sl = make_slab(ro, rd);
                    trav_reset(tv, 0);  // node 0 is always an internal node
                } else {
                    trav_reset(tv, kTravDone);
                    trace_flat(sc, ro, rd, tv.best_t, tv.best_idx);
                    if (STATS) n_tests += (uint32_t)sc.n_prims;
                }
            }
        }
        if (__ballot(slot >= 0) == 0ull) {
            if (lds_peek(&L.live) == 0u) break;
            if (++guard > kGuardLimit) { atomicExch(&L.abort, 1u); break; }
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        guard = 0;
        if (ACCEL == 1) {
            for (int it = 0; it < burst; ++it) {
                const unsigned long long m_node = __ballot(tv.cur >= 0 && tv.cur != kTravDone);
                const unsigned long long m_leaf = __ballot(tv.cur < 0);
                if ((m_node | m_leaf) == 0ull) break;
                if (STATS) wave_iters++;
                if (__popcll(m_node) >= __popcll(m_leaf)) {
                    if (tv.cur >= 0 && tv.cur != kTravDone) {
                        trav_node_step(sc.nodes, sl, stack, tv);
                        if (STATS) n_nodes++;
                    }
                } else {
                    if (tv.cur < 0) {
                        trav_leaf_step(sc.prims_bvh, ro, rd, stack, tv);
                        if (STATS) n_tests++;
                    }
                }
            }
        }
    }

    // ---- counters: one atomic per wave
    unsigned long long r = n_rays, nn = n_nodes, nt = n_tests;
    for (int off = 32; off > 0; off >>= 1) {
        r += __shfl_down(r, off);
        if (STATS) {
            nn += __shfl_down(nn, off);
            nt += __shfl_down(nt, off);
        }
    }
    if (lane == 0) {
        atomicAdd((unsigned long long*)(fa.counters + 2), r);
        atomicAdd((unsigned long long*)(fa.counters + 8), r);
        if (STATS) {
            atomicAdd((unsigned long long*)(fa.counters + 4), nn);
            atomicAdd((unsigned long long*)(fa.counters + 6), nt);
            atomicAdd((unsigned long long*)(fa.counters + 16), nn + (ACCEL == 1 ? nt : 0ull));
            atomicAdd((unsigned long long*)(fa.counters + 18), wave_iters);
        }
        if (lds_peek(&L.abort) != 0u) atomicExch(&fa.counters[30], 1u);
    }
}

}  // namespace

int pool_slots_per_block() { return kPool; }

// Workgroups of the pooled launch: as many as are resident at once, fewer when the image
// is small.  The path pool must hold grid * pool_slots_per_block() slots.
hipError_t pool_grid(const FrameArgs& fa, int accel, bool stats, int cu_count, unsigned* grid) {
    void (*kernel)(DeviceScene, FrameArgs, PathPool, int, int) =
        accel == 1 ? (stats ? pool_kernel<1, true> : pool_kernel<1, false>) : (stats ? pool_kernel<0, true> : pool_kernel<0, false>);
    const size_t lds = (size_t)fa.stack_depth * kBlock * sizeof(uint32_t);
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, lds);
    if (e != hipSuccess) return e;
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
    if (const char* q = getenv("RT_POOL_PER_CU")) {  // tuning aid
        int v = atoi(q);
        if (v >= 1 && v < per_cu) per_cu = v;
    }
    const long long want = ((long long)fa.total_slots + kPool - 1) / kPool, cap = (long long)cu_count * per_cu;
    *grid = (unsigned)(want < cap ? want : cap);
    return hipSuccess;
}

hipError_t launch_pool(const DeviceScene& sc, const FrameArgs& fa, const PathPool& pp, int accel, bool stats, unsigned grid,
                       int refill_min, int burst, hipStream_t stream) {
    if (grid == 0) return hipSuccess;
    void (*kernel)(DeviceScene, FrameArgs, PathPool, int, int) =
        accel == 1 ? (stats ? pool_kernel<1, true> : pool_kernel<1, false>) : (stats ? pool_kernel<0, true> : pool_kernel<0, false>);
    const size_t lds = (size_t)fa.stack_depth * kBlock * sizeof(uint32_t);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds, stream, sc, fa, pp, refill_min, burst);
    return hipGetLastError();
}

}  // namespace rtd
