// rt_wavefront.hip — the staged ("wavefront") form of the path tracer for gfx950 (MI355X).
//
// One frame = one execution of shaders/raytracer.comp::main for every pixel this context
// owns (raytracer.comp:956-996), as in rt_kernels.hip, but the per-pixel program is cut at
// its nearest-hit searches (hit_scene, raytracer.comp:469-508).  Every pixel in flight owns
// one slot of path state in device memory (PathPool); a frame is
//     generate, then repeat { search launch ; shading launch }
// and every repetition advances each live path by exactly one ray.  A pixel's samples and
// bounces are still computed strictly one after the other with the pixel's own RNG state,
// operation for operation as the shader orders them, so the result is independent of how
// the work is scheduled (the tests compare bit patterns against the single-kernel form and
// the CPU restatement).
//
// Why: inside one wave the single-kernel form runs 64 pixels whose paths disagree about
// what to do next (search, light sampling, BSDF evaluation, sample end), so most vector
// instructions execute with most lanes masked off.  Here each launch regroups the paths:
//   * the search kernel's waves stream over 64-slot groups, compact the slots that wait
//     for a search (ballot + popcount, through a wave-private LDS list) and hand a new ray
//     to a lane the moment its previous search ends;
//   * the shading kernel's workgroups read the stage words of a range of slots, compact
//     them by kind of work into LDS lists, and give every wave 64 items of ONE kind.
// Nothing is queued in global memory and no global atomic sits on the data path: a slot
// is owned by one workgroup per launch, so all compaction is workgroup-local.
#include "rt_stages.h"

#include <cstdlib>

namespace rtd {

namespace {

constexpr int kMaxRange = 4096;  // most slots one workgroup compacts per pass (LDS lists of uint16 offsets)


// ------------------------------------------------------------------------------------ generate
// Seed every pixel (raytracer.comp:959-961) and write its first primary ray.
__global__ void __launch_bounds__(kBlock) wf_generate_kernel(const FrameArgs fa, const PathPool pp) {
    const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
    if (s >= pp.n_slots) return;
    int x, y;
    uint32_t st = WF_IDLE;
    if (slot_to_pixel(fa, s, x, y)) {
        start_pixel(fa, GlobalSlot{pp, s}, x, y);
        return;
    }
    pp.stage[s] = st;
}

// ------------------------------------------------------------------------------------ search
// Persistent waves; wave w takes the 64-slot groups w, w + W, w + 2W, ...  Within a group
// the slots that wait for a search are compacted into a wave-private list; a lane whose
// search has ended stores its result and takes the next listed slot.  The BVH walk is the
// voted node/leaf step of the single-kernel form: each iteration the wave executes the kind
// of step most of its lanes wait for.
template <int ACCEL, bool STATS>
__global__ void __launch_bounds__(kBlock) wf_search_kernel(const DeviceScene sc, const FrameArgs fa, const PathPool pp, int iter,
                                                           int refill_min) {
    extern __shared__ uint32_t s_mem[];  // [stack_depth][kBlock] traversal stacks, then [kBlock] slot lists
    if (iter > 0 && pp.active[iter] == 0u) return;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const LdsStack stack{s_mem + tid};
    uint32_t* const list = s_mem + (size_t)fa.stack_depth * kBlock + (tid & ~63);

    const uint32_t n_groups = (pp.n_slots + 63u) / 64u;
    const uint32_t n_waves = gridDim.x * (kBlock / 64);
    uint32_t group = blockIdx.x * (kBlock / 64) + (uint32_t)(tid >> 6);
    uint32_t list_n = 0, list_head = 0;

    int slot = -1;
    V3 ro = v3(0.0f), rd = v3(0.0f, 0.0f, 1.0f);
    Slab sl = make_slab(ro, rd);
    Trav tv;
    trav_reset(tv, kTravDone);
    uint32_t n_nodes = 0, n_tests = 0;
    unsigned long long wave_iters = 0;

    for (;;) {
        const unsigned long long idle = __ballot(tv.cur == kTravDone);
        if (idle != 0ull) {
            if (tv.cur == kTravDone && slot >= 0) {
                pp.hit[slot] = make_uint2(__float_as_uint(tv.best_t), (uint32_t)tv.best_idx);
                slot = -1;
            }
            if (__popcll(idle) >= refill_min || idle == ~0ull) {
                while (list_head == list_n && group < n_groups) {
                    const uint32_t s = group * 64u + (uint32_t)lane;
                    const uint32_t st = (s < pp.n_slots) ? pp.stage[s] : (uint32_t)WF_IDLE;
                    const bool need = (st == WF_PATH_FIRST || st == WF_PATH_NEXT || st == WF_LIGHT_SEARCH);
                    const unsigned long long m = __ballot(need);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (need) list[__popcll(m & lanes_below())] = s;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    list_n = (uint32_t)__popcll(m);
                    list_head = 0;
                    group += n_waves;
                }
                if (list_head < list_n) {
                    const uint32_t rank = (uint32_t)__popcll(idle & lanes_below());
                    if (tv.cur == kTravDone && list_head + rank < list_n) {
                        slot = (int)list[list_head + rank];
                        ro = xyz(pp.ray_o[slot]);
                        rd = xyz(pp.ray_d[slot]);
                        if (ACCEL == 1) {
                            sl = make_slab(ro, rd);
                            trav_reset(tv, 0);  // node 0 is always an internal node
                        } else {
                            trav_reset(tv, kTravDone);
                            trace_flat(sc, ro, rd, tv.best_t, tv.best_idx);
                            if (STATS) n_tests += (uint32_t)sc.n_prims;
                        }
                    }
                    const uint32_t want = (uint32_t)__popcll(idle), have = list_n - list_head;
                    list_head += want < have ? want : have;
                }
            }
        }
        const unsigned long long m_node = __ballot(tv.cur >= 0 && tv.cur != kTravDone);
        const unsigned long long m_leaf = __ballot(tv.cur < 0);
        if ((m_node | m_leaf) == 0ull) {
            if (__ballot(slot >= 0) != 0ull) continue;  // flat search: results wait to be stored
            if (list_head == list_n && group >= n_groups) break;
            continue;
        }
        if (STATS) wave_iters++;
        if ((int)__popcll(m_node) >= (int)__popcll(m_leaf)) {
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

    if (STATS) {
        unsigned long long nn = n_nodes, nt = n_tests;
        for (int off = 32; off > 0; off >>= 1) {
            nn += __shfl_down(nn, off);
            nt += __shfl_down(nt, off);
        }
        if (lane == 0) {
            atomicAdd((unsigned long long*)(fa.counters + 4), nn);
            atomicAdd((unsigned long long*)(fa.counters + 6), nt);
            atomicAdd((unsigned long long*)(fa.counters + 16), nn + nt);
            atomicAdd((unsigned long long*)(fa.counters + 18), wave_iters);
        }
    }
}

// ------------------------------------------------------------------------------------ shading
struct ShadeLds {
    uint16_t list[4][kMaxRange];  // slot offsets by kind of work: 0 first-bounce results, 1 later-bounce results,
                                  // 2 light-query results, 3 sample ends
    uint32_t n[4];
    uint32_t cursor[2];           // next 64-item chunk of phase 1 / phase 2
    uint32_t rays;                // searches this range consumed
    uint32_t alive;               // slots of this range still in flight afterwards
};

// Append `value` to list `k` for the lanes that ask; one LDS atomic per wave.
RT_DEV void list_push(ShadeLds& L, int k, bool want, uint32_t value) {
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane_id() == leader) base = atomicAdd(&L.n[k], (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    if (want) L.list[k][base + (uint32_t)__popcll(m & lanes_below())] = (uint16_t)value;
}

RT_DEV void count_lanes(uint32_t* counter, bool pred) {
    const unsigned long long m = __ballot(pred);
    if (m != 0ull && lane_id() == __ffsll((long long)m) - 1) atomicAdd(counter, (uint32_t)__popcll(m));
}

__global__ void __launch_bounds__(kBlock, 4) wf_shade_kernel(const DeviceScene sc, const FrameArgs fa, const PathPool pp, int iter,
                                                             int range_len) {
    __shared__ ShadeLds L;
    if (iter > 0 && pp.active[iter] == 0u) return;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const float4* __restrict__ recs = (fa.accel == 1) ? sc.prims_bvh : sc.prims_flat;
    uint32_t rays_total = 0, alive_total = 0;  // thread 0 only

    for (uint32_t base = blockIdx.x * (uint32_t)range_len; base < pp.n_slots; base += gridDim.x * (uint32_t)range_len) {
        if (tid < 4) L.n[tid] = 0;
        if (tid < 2) L.cursor[tid] = 0;
        if (tid == 0) { L.rays = 0; L.alive = 0; }
        __syncthreads();
        // ---- compact the range's slots by the kind of work they wait for
        for (int o = tid; o < range_len; o += kBlock) {
            const uint32_t s = base + (uint32_t)o;
            const uint32_t st = (s < pp.n_slots) ? pp.stage[s] : (uint32_t)WF_IDLE;
            list_push(L, 0, st == WF_PATH_FIRST, (uint32_t)o);
            list_push(L, 1, st == WF_PATH_NEXT, (uint32_t)o);
            list_push(L, 2, st >= WF_LIGHT_SEARCH && st <= WF_LIGHT_UNDEF, (uint32_t)o);
            count_lanes(&L.rays, st == WF_PATH_FIRST || st == WF_PATH_NEXT || st == WF_LIGHT_SEARCH);
        }
        __syncthreads();
        const uint32_t n0 = L.n[0], n1 = L.n[1], n2 = L.n[2];
        if (n0 + n1 + n2 != 0u) {
            // ---- phase 1: results of searches; 64 items of one kind per wave, dearest kind first
            const uint32_t c2 = (n2 + 63u) / 64u, c1 = (n1 + 63u) / 64u, c0 = (n0 + 63u) / 64u;
            for (;;) {
                uint32_t chunk = 0;
                if (lane == 0) chunk = atomicAdd(&L.cursor[0], 1u);
                chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk);
                if (chunk >= c0 + c1 + c2) break;
                int kind;
                uint32_t idx, n;
                if (chunk < c2) { kind = 2; idx = chunk * 64u; n = n2; }
                else if (chunk < c2 + c1) { kind = 1; idx = (chunk - c2) * 64u; n = n1; }
                else { kind = 0; idx = (chunk - c2 - c1) * 64u; n = n0; }
                idx += (uint32_t)lane;
                const bool valid = idx < n;
                bool ends = false, alive = false;
                uint32_t o = 0;
                if (valid) {
                    uint32_t next_stage;
                    o = L.list[kind][idx];
                    ends = shade_item(sc, fa, GlobalSlot{pp, base + o}, recs, kind, pp.stage[base + o], next_stage);
                    alive = !ends;
                }
                list_push(L, 3, ends, o);
                count_lanes(&L.alive, alive);
            }
        }
        __syncthreads();
        // ---- phase 2: sample ends
        const uint32_t n3 = L.n[3];
        if (n3 != 0u) {
            const uint32_t c3 = (n3 + 63u) / 64u;
            for (;;) {
                uint32_t chunk = 0;
                if (lane == 0) chunk = atomicAdd(&L.cursor[1], 1u);
                chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk);
                if (chunk >= c3) break;
                const uint32_t idx = chunk * 64u + (uint32_t)lane;
                bool alive = false;
                if (idx < n3) end_item(fa, GlobalSlot{pp, base + L.list[3][idx]}, alive);
                count_lanes(&L.alive, alive);
            }
        }
        __syncthreads();
        if (tid == 0) {
            rays_total += L.rays;
            alive_total += L.alive;
        }
    }
    if (tid == 0) {
        if (rays_total != 0u) {
            atomicAdd((unsigned long long*)(fa.counters + 2), (unsigned long long)rays_total);
            atomicAdd((unsigned long long*)(fa.counters + 8), (unsigned long long)rays_total);
        }
        if (alive_total != 0u) atomicAdd(&pp.active[iter + 1], alive_total);
    }
}

}  // namespace

hipError_t launch_wavefront(const DeviceScene& sc, const FrameArgs& fa, const PathPool& pp, int accel, bool stats, int iterations,
                            int refill_min, int cu_count, hipStream_t stream) {
    if (pp.n_slots == 0) return hipSuccess;
    hipLaunchKernelGGL(wf_generate_kernel, dim3((pp.n_slots + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, fa, pp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;

    void (*search)(DeviceScene, FrameArgs, PathPool, int, int) =
        accel == 1 ? (stats ? wf_search_kernel<1, true> : wf_search_kernel<1, false>)
                   : (stats ? wf_search_kernel<0, true> : wf_search_kernel<0, false>);
    const size_t lds = ((size_t)fa.stack_depth + 1) * kBlock * sizeof(uint32_t);
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, search, kBlock, lds);
    if (e != hipSuccess) return e;
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
    if (const char* q = getenv("RT_WF_SEARCH_PER_CU")) {  // tuning aid
        int v = atoi(q);
        if (v >= 1 && v < per_cu) per_cu = v;
    }
    const long long groups = ((long long)pp.n_slots + 63) / 64;
    long long want = (groups + 3) / 4, cap = (long long)cu_count * per_cu;
    const unsigned grid_search = (unsigned)(want < cap ? want : cap);

    // Range of slots one workgroup compacts per pass: large enough that each kind of work
    // fills whole waves, small enough that the launch has a few workgroups per CU.
    long long range = ((long long)pp.n_slots + (long long)cu_count * 4 - 1) / ((long long)cu_count * 4);
    range = (range + kBlock - 1) / kBlock * kBlock;
    if (range < 1024) range = 1024;
    if (const char* q = getenv("RT_WF_SHADE_RANGE")) {  // tuning aid
        int v = atoi(q);
        if (v >= 256) range = v / kBlock * kBlock;
    }
    if (range > kMaxRange) range = kMaxRange;
    want = ((long long)pp.n_slots + range - 1) / range;
    cap = (long long)cu_count * 4;
    const unsigned grid_shade = (unsigned)(want < cap ? want : cap);

    for (int it = 0; it < iterations; ++it) {
        hipLaunchKernelGGL(search, dim3(grid_search), dim3(kBlock), lds, stream, sc, fa, pp, it, refill_min);
        hipLaunchKernelGGL(wf_shade_kernel, dim3(grid_shade), dim3(kBlock), 0, stream, sc, fa, pp, it, (int)range);
    }
    return hipGetLastError();
}

}  // namespace rtd
