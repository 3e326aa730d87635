// rt_api.hip — device half of the C ABI in include/rt_abi.h: the context that owns the
// hipMalloc'd scene / framebuffer memory and launches the kernels.  It stands where the
// reference's Vulkan buffer, descriptor and dispatch code stands (src/main.cpp:1222-1306,
// 1424-1606); each entry point cites its counterpart in the header.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "device/rt_kernels.h"
#include "host/bvh_builder.h"
#include "host/guarded.h"
#include "host/scene_handle.h"
#include "rt_context.h"

namespace {
thread_local std::string g_create_error;  // what rt_last_error(NULL) returns: the calling thread's last failed rt_create
}

namespace rtapi {
int fail(RtContext* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}
int check_abandoned(RtContext* c) {
    if (c->h_abort && *(volatile uint32_t*)c->h_abort != 0u) {
        *(volatile uint32_t*)c->h_abort = 0u;
        return fail(c, RT_ERR_HIP, "the pooled pipeline gave up waiting for work inside a workgroup; a frame since the last synchronisation is incomplete");
    }
    return RT_OK;
}
}  // namespace rtapi
using rtapi::fail;

// ---------------------------------------------------------------------------------------------
// Scratch memory of rt_render_frame_noalloc: what a frame needs beyond its inputs and outputs (the
// tile order fed back from call to call and the pooled kernel's slot records and stack overflow), in a
// file-scope device array instead of hipMalloc'd memory.  Every GPU the library runs on holds its
// own copy at its own address until the process ends; the array's name is only a host-side stand-in
// for it, so host code uses the address hipGetSymbolAddress returns for the current device
// (g_scratch_ptr, taken once per device in rt_create) wherever the memory is meant.
namespace {
constexpr size_t kScratchTiles = 128u * 1024u;      // 8x8 tiles of the largest frame (8.39 M pixels, e.g. 3840 x 2184)
constexpr size_t kScratchStampBytes = 256;          // 8 words: what the stored tile order was made for
constexpr size_t kScratchAuxBytes = 192u << 10;     // work area of the tile order kernels (rtd::tile_order_aux_bytes() <= this)
constexpr size_t kScratchPoolBytes = 46u << 20;     // 256 pooled workgroups x (1344 x 64 B records + 1024 x 24 stack words) = 45.0 MiB, rounded up (the size dates from a pool of 1392 slots)
constexpr size_t kScratchBytes = kScratchStampBytes + kScratchAuxBytes + 2 * 4 * kScratchTiles + kScratchPoolBytes;  // 47 MiB + 192 KiB + 256 B
constexpr int kMaxDevices = 64;
}  // namespace
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
namespace {
unsigned char* g_scratch_ptr[kMaxDevices] = {nullptr};  // per device: the address of its copy of `scratch`

// Device memory a frame needs beyond the scene and the framebuffer.
struct FrameScratch {
    uint32_t* counters = nullptr;    // 48 words (FrameArgs::counters)
    uint32_t* tile_cost = nullptr;   // per 8x8 tile, this frame's cost
    uint32_t* tile_order = nullptr;  // the order to hand the tiles out in (null: image order)
    void* pool = nullptr;            // cold slot records + stack overflow of the pooled kernel
    size_t pool_bytes = 0;
};

void free_scene(RtContext* c) {
    void** ptrs[] = {&c->d_prims_bvh, &c->d_prims_flat, &c->d_nodes, &c->d_nodes_wide, &c->d_nodes_wide16, &c->d_materials, &c->d_lights, &c->d_triangles};
    for (void** p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    c->has_scene = false;
}

void free_framebuffer(RtContext* c) {
    if (c->owns_fb) {
        if (c->d_accum) (void)hipFree(c->d_accum);
        if (c->d_counts) (void)hipFree(c->d_counts);
        if (c->d_bgra) (void)hipFree(c->d_bgra);
    }
    if (c->d_primary_ids) (void)hipFree(c->d_primary_ids);
    if (c->d_path_hash) (void)hipFree(c->d_path_hash);
    c->d_accum = c->d_counts = c->d_bgra = nullptr;
    c->d_primary_ids = nullptr;
    c->d_path_hash = nullptr;
    c->owns_fb = false;
    c->width = c->height = 0;
}

// hipMalloc max(bytes, min_bytes), zero it, copy `bytes` from src.
int upload_array(RtContext* c, void** dst, const void* src, size_t bytes, size_t min_bytes) {
    size_t alloc = bytes > min_bytes ? bytes : min_bytes;
    RT_HIP(c, hipMalloc(dst, alloc));
    RT_HIP(c, hipMemsetAsync(*dst, 0, alloc, c->stream));
    if (bytes) RT_HIP(c, hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    return RT_OK;
}

void free_pool(RtContext* c) {
    if (c->d_pool) (void)hipFree(c->d_pool);
    c->d_pool = nullptr;
    c->pool_bytes = 0;
}

// Device memory of the pooled pipeline: one allocation, grown when needed.
int ensure_pool_bytes(RtContext* c, size_t bytes) {
    if (bytes > c->pool_bytes) {
        RT_HIP(c, hipStreamSynchronize(c->stream));
        if (c->d_pool) (void)hipFree(c->d_pool);
        c->d_pool = nullptr;
        c->pool_bytes = 0;
        size_t cap = (bytes + 4095) / 4096 * 4096;
        RT_HIP(c, hipMalloc(&c->d_pool, cap));
        c->pool_bytes = cap;
    }
    return RT_OK;
}

// Pooled pipeline: per workgroup 64 B of cold state per slot and a stack overflow area per lane.
size_t pooled_pool_bytes(unsigned grid, int overflow_words) {
    return (size_t)grid * ((size_t)rtd::pool_slots_per_block() * 64 + (size_t)rtd::pool_threads_per_block() * (size_t)overflow_words * 4);
}
void carve_pooled_pool(void* base, unsigned grid, int overflow_words, rtd::PathPool& pp) {
    const size_t cold = (size_t)grid * (size_t)rtd::pool_slots_per_block() * 64;
    pp.cold = (float4*)base;
    pp.stack_overflow = (uint32_t*)((char*)base + cold);
    pp.n_slots = grid * (unsigned)rtd::pool_slots_per_block();
    pp.overflow_words = (uint32_t)overflow_words;
}

int alloc_debug_buffers(RtContext* c) {
    size_t n = (size_t)c->width * (size_t)c->height;
    if (!c->d_primary_ids) {
        RT_HIP(c, hipMalloc((void**)&c->d_primary_ids, n * 4));
        RT_HIP(c, hipMemsetAsync(c->d_primary_ids, 0xff, n * 4, c->stream));
    }
    if (!c->d_path_hash) {
        RT_HIP(c, hipMalloc((void**)&c->d_path_hash, n * 4));
        RT_HIP(c, hipMemsetAsync(c->d_path_hash, 0, n * 4, c->stream));
    }
    return RT_OK;
}

}  // namespace

namespace rtapi {

// Everything rt_upload_scene checks and builds on the host: validated indices, the flat-order records, the binary
// tree in its device form, its four-wide collapse.  materials / lights / triangles stay borrowed from the caller.
int build_scene(const RtOptions& opt, const RtSphere* spheres, int32_t n_spheres, const RtMaterial* materials, int32_t n_materials,
                const RtLight* lights, int32_t n_lights, const RtTriangle* triangles, int32_t n_triangles, const RtVertex* vertices,
                int32_t n_vertices, const uint32_t* indices, int32_t n_indices, const RtMeshInfo* meshes, int32_t n_meshes, BuiltScene& out,
                std::string& err) {
    if (n_spheres < 0 || n_materials < 0 || n_lights < 0 || n_triangles < 0 || n_vertices < 0 || n_indices < 0 || n_meshes < 0) {
        err = "negative element count";
        return RT_ERR_INVALID_ARGUMENT;
    }
    if ((n_spheres && !spheres) || (n_materials && !materials) || (n_lights && !lights) || (n_triangles && !triangles) ||
        (n_vertices && !vertices) || (n_indices && !indices) || (n_meshes && !meshes)) {
        err = "null array with non-zero count";
        return RT_ERR_INVALID_ARGUMENT;
    }
    // The kernels index materials / triangles / vertices with values taken from the scene:
    // check every such index on the host so that no launch can read out of bounds.
    for (int32_t i = 0; i < n_spheres; ++i)
        if (spheres[i].mat < 0 || spheres[i].mat >= n_materials) { err = "sphere material index out of range"; return RT_ERR_INVALID_ARGUMENT; }
    for (int32_t i = 0; i < n_triangles; ++i)
        if (triangles[i].mat < 0 || triangles[i].mat >= n_materials) { err = "triangle material index out of range"; return RT_ERR_INVALID_ARGUMENT; }
    for (int32_t i = 0; i < n_meshes; ++i)
        if (meshes[i].material < 0 || meshes[i].material >= n_materials) { err = "mesh material index out of range"; return RT_ERR_INVALID_ARGUMENT; }
    for (int32_t i = 0; i < n_lights; ++i) {
        if (lights[i].type == RT_LIGHT_TRIANGLE) {
            float fx = lights[i].pos_angle_aux[0];
            if (!(fx >= 0.0f) || !(fx < (float)(n_triangles > 0 ? n_triangles : 1))) {
                err = "TRIANGLE light refers to a triangle that does not exist";
                return RT_ERR_INVALID_ARGUMENT;
            }
        }
    }
    rt::SceneArrays a;
    a.spheres = spheres;     a.n_spheres = n_spheres;
    a.triangles = triangles; a.n_triangles = n_triangles;
    a.vertices = vertices;   a.n_vertices = n_vertices;
    a.indices = indices;     a.n_indices = n_indices;
    a.meshes = meshes;       a.n_meshes = n_meshes;
    if (!rt::gather_primitives(a, out.flat, &err)) return RT_ERR_INVALID_ARGUMENT;
    // (the kernels address records and nodes by 32-bit byte offsets: 2^26 records of 48 bytes are 3.2 GB, and their
    // tree's nodes - at most a third as many four-wide nodes of 128 bytes, as many binary ones of 64 - stay below 4 GB too)
    if (out.flat.size() >= (size_t)(1u << 26)) { err = "more than 2^26 primitives"; return RT_ERR_TOO_LARGE; }
    if (!rt::build_bvh(a, out.bvh, &err, opt.bvh_max_leaf)) return RT_ERR_INVALID_ARGUMENT;
    if (out.bvh.depth > rt::kMaxBvhDepth || out.bvh.max_leaf > 8) { err = "BVH exceeds traversal limits"; return RT_ERR_TOO_LARGE; }
    rt::layout_treelets(out.bvh, opt.bvh_treelet);
    // device form of a node: each child as the traversal's cursor word (an internal node's index, or
    // bit 31 | (count - 1) << 28 | first record for a leaf), the count of an unused slot -1
    out.dev_nodes = out.bvh.nodes;
    for (rt::BvhNode& n : out.dev_nodes) {
        if (n.child0 < 0) n.count0 = -1; else if (n.count0 > 0) n.child0 = (int32_t)(0x80000000u | ((uint32_t)(n.count0 - 1) << 28) | (uint32_t)n.child0);
        if (n.child1 < 0) n.count1 = -1; else if (n.count1 > 0) n.child1 = (int32_t)(0x80000000u | ((uint32_t)(n.count1 - 1) << 28) | (uint32_t)n.child1);
    }
    rt::collapse_wide(out.bvh, out.wide);
    out.materials = materials; out.n_materials = n_materials;
    out.lights = lights;       out.n_lights = n_lights;
    out.triangles = triangles; out.n_triangles = n_triangles;
    out.n_spheres = n_spheres;
    out.n_meshes = n_meshes;
    return RT_OK;
}

// The device half of rt_upload_scene: copies of a built scene's arrays in this context's device memory.
int upload_built(RtContext* c, const BuiltScene& b) {
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    int rc;
    // (a failed upload still waits for the copies already queued: they read the host vectors)
    auto bail = [&](int code) { (void)hipStreamSynchronize(c->stream); return code; };
    if ((rc = upload_array(c, &c->d_prims_bvh, b.bvh.prims.data(), b.bvh.prims.size() * sizeof(rt::PrimRecord), 48)) != RT_OK) return bail(rc);
    if ((rc = upload_array(c, &c->d_prims_flat, b.flat.data(), b.flat.size() * sizeof(rt::PrimRecord), 48)) != RT_OK) return bail(rc);
    if ((rc = upload_array(c, &c->d_nodes, b.dev_nodes.data(), b.dev_nodes.size() * sizeof(rt::BvhNode), 64)) != RT_OK) return bail(rc);
    if ((rc = upload_array(c, &c->d_nodes_wide, b.wide.nodes.data(), b.wide.nodes.size() * sizeof(rt::WideNode), 128)) != RT_OK) return bail(rc);
    if (b.wide.half_ok && (rc = upload_array(c, &c->d_nodes_wide16, b.wide.nodes16.data(), b.wide.nodes16.size() * sizeof(rt::WideNode16) + 48, 128)) != RT_OK)
        return bail(rc);  // (+ 48: the last node's fifth 16-byte load ends inside the allocation whatever its alignment)
    if ((rc = upload_array(c, &c->d_materials, b.materials, (size_t)b.n_materials * sizeof(RtMaterial), sizeof(RtMaterial))) != RT_OK) return bail(rc);
    if ((rc = upload_array(c, &c->d_lights, b.lights, (size_t)b.n_lights * sizeof(RtLight), sizeof(RtLight))) != RT_OK) return bail(rc);
    if ((rc = upload_array(c, &c->d_triangles, b.triangles, (size_t)b.n_triangles * sizeof(RtTriangle), sizeof(RtTriangle))) != RT_OK) return bail(rc);
    RT_HIP(c, hipStreamSynchronize(c->stream));  // the caller's vectors may go out of scope
    c->n_prims = (int32_t)b.flat.size();
    c->n_spheres = b.n_spheres;
    c->n_triangles = b.n_triangles;
    c->n_meshes = b.n_meshes;
    c->n_lights = b.n_lights;
    c->n_materials = b.n_materials;
    c->bvh_nodes = (int32_t)b.bvh.nodes.size();
    c->bvh_depth = b.bvh.depth;
    c->wide_nodes = (int32_t)b.wide.nodes.size();
    c->wide_stack_need = b.wide.stack_need;
    c->has_scene = true;
    c->scene_generation++;
    return RT_OK;
}

}  // namespace rtapi

extern "C" {

int rt_create(int device_ordinal, RtContext** out_ctx) {
    if (!out_ctx) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "out_ctx is null");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, RT_ERR_HIP, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device_ordinal < 0 || device_ordinal >= n) return fail(nullptr, RT_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    RT_HIP(nullptr, hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    RT_HIP(nullptr, hipGetDeviceProperties(&prop, device_ordinal));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, RT_ERR_UNSUPPORTED,
                    std::string("device is ") + prop.gcnArchName + "; this library carries gfx950 (MI355X) code only");
    RtContext* c = new (std::nothrow) RtContext();
    if (!c) return fail(nullptr, RT_ERR_TOO_LARGE, "out of host memory");
    c->device = device_ordinal;
    c->cu_count = prop.multiProcessorCount;
    hipError_t e1 = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    hipError_t e2 = hipEventCreate(&c->ev_begin);
    hipError_t e3 = hipEventCreate(&c->ev_end);
    for (int k = 0; k < 4 && e3 == hipSuccess; ++k) e3 = hipEventCreate(&c->ev_trial[k >> 1][k & 1]);
    hipError_t e4 = hipMalloc((void**)&c->d_counters, 192);
    if (e4 == hipSuccess) e4 = hipMemset(c->d_counters, 0, 192);
    if (e4 == hipSuccess) e4 = hipHostMalloc((void**)&c->h_abort, 64, hipHostMallocMapped);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
        rt_destroy(c);
        return fail(nullptr, RT_ERR_HIP, "stream/event/counter creation failed");
    }
    *c->h_abort = 0u;
    // how many XCDs this device spreads its workgroups over (for launches of several frames); 0 if it cannot be told
    {
        uint32_t* probe = nullptr;
        if (hipMalloc((void**)&probe, (size_t)c->cu_count * 4) == hipSuccess) {
            (void)rtd::probe_xcds(c->cu_count, probe, &c->n_xcds, c->stream);
            (void)hipFree(probe);
        }
    }
    // this device's copy of the no-allocation scratch array (the device is current since hipSetDevice above)
    if (device_ordinal < kMaxDevices && g_scratch_ptr[device_ordinal] == nullptr) {
        void* sp = nullptr;
        hipError_t es = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));
        if (es != hipSuccess || sp == nullptr) {
            rt_destroy(c);
            return fail(nullptr, RT_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(es));
        }
        g_scratch_ptr[device_ordinal] = (unsigned char*)sp;
    }
    // tuning aids: the environment may override an option's default, read once here
    struct { const char* name; int option; } env[] = {
        {"RT_EXIT_QUOTA_PCT", RT_OPT_EXIT_QUOTA_PCT}, {"RT_REFILL_MIN", RT_OPT_REFILL_MIN}, {"RT_POOL_BURST", RT_OPT_POOL_BURST},
        {"RT_POOL_JOB_MIN", RT_OPT_POOL_JOB_MIN}, {"RT_TILE_CLASSES", RT_OPT_TILE_CLASSES}, {"RT_POOL_PER_CU", RT_OPT_POOL_PER_CU},
        {"RT_MAX_LEAF", RT_OPT_BVH_MAX_LEAF}, {"RT_POOL_SLOTS", RT_OPT_POOL_SLOTS}, {"RT_BVH_TREELET", RT_OPT_BVH_TREELET},
        {"RT_BVH_WIDTH", RT_OPT_BVH_WIDTH}, {"RT_POOL_SHADER_WAVES", RT_OPT_POOL_SHADER_WAVES}, {"RT_BVH_NODE_FORMAT", RT_OPT_BVH_NODE_FORMAT}};
    for (const auto& e : env)
        if (const char* v = getenv(e.name)) (void)rt_set_option(c, e.option, atoi(v));  // out-of-range values are ignored
    *out_ctx = c;
    return RT_OK;
}

int rt_destroy(RtContext* c) {
    if (!c) return RT_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_scene(c);
    free_framebuffer(c);
    free_pool(c);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->h_abort) (void)hipHostFree(c->h_abort);
    for (int k = 0; k < 4; ++k)
        if (c->ev_trial[k >> 1][k & 1]) (void)hipEventDestroy(c->ev_trial[k >> 1][k & 1]);
    if (c->d_tile_order) (void)hipFree(c->d_tile_order);
    if (c->d_tile_cost) (void)hipFree(c->d_tile_cost);
    if (c->d_tile_aux) (void)hipFree(c->d_tile_aux);
    if (c->d_batch) (void)hipFree(c->d_batch);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RT_OK;
}

int rt_upload_scene(RtContext* c, const RtSphere* spheres, int32_t n_spheres, const RtMaterial* materials,
                    int32_t n_materials, const RtLight* lights, int32_t n_lights, const RtTriangle* triangles,
                    int32_t n_triangles, const RtVertex* vertices, int32_t n_vertices, const uint32_t* indices,
                    int32_t n_indices, const RtMeshInfo* meshes, int32_t n_meshes) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    return rt::guarded([&]() -> int {
        rtapi::BuiltScene built;
        std::string err;
        int rc = rtapi::build_scene(c->opt, spheres, n_spheres, materials, n_materials, lights, n_lights, triangles, n_triangles, vertices,
                                    n_vertices, indices, n_indices, meshes, n_meshes, built, err);
        if (rc != RT_OK) return fail(c, rc, err);
        return rtapi::upload_built(c, built);
    });
}

int rt_upload_scene_from(RtContext* c, const RtScene* s) {
    if (!c || !s) return RT_ERR_INVALID_ARGUMENT;
    const rt::Scene& sc = s->scene;
    return rt_upload_scene(c, sc.sphereVec.data(), sc.total_spheres, sc.materialVec.data(), (int32_t)sc.materialVec.size(),
                           sc.lightsVec.data(), sc.total_lights, sc.triangleVec.data(), sc.total_triangles,
                           sc.vertexVec.data(), (int32_t)sc.vertexVec.size(), sc.indexVec.data(),
                           (int32_t)sc.indexVec.size(), sc.meshVec.data(), sc.total_meshes);
}

int rt_resize(RtContext* c, int32_t width, int32_t height) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 27))
        return fail(c, RT_ERR_INVALID_ARGUMENT, "bad image size");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    free_framebuffer(c);
    size_t n = (size_t)width * (size_t)height;
    c->owns_fb = true;
    RT_HIP(c, hipMalloc(&c->d_accum, n * 16));
    RT_HIP(c, hipMalloc(&c->d_counts, n * 4));
    RT_HIP(c, hipMalloc(&c->d_bgra, n * 4));
    RT_HIP(c, hipMemsetAsync(c->d_accum, 0, n * 16, c->stream));
    RT_HIP(c, hipMemsetAsync(c->d_counts, 0, n * 4, c->stream));
    RT_HIP(c, hipMemsetAsync(c->d_bgra, 0, n * 4, c->stream));
    c->width = width;
    c->height = height;
    return RT_OK;
}

int rt_bind_framebuffer(RtContext* c, int32_t width, int32_t height, void* d_accum, void* d_counts, void* d_bgra) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 27) || !d_accum || !d_counts || !d_bgra)
        return fail(c, RT_ERR_INVALID_ARGUMENT, "bad framebuffer binding");
    if (((uintptr_t)d_accum & 15u) || ((uintptr_t)d_counts & 3u) || ((uintptr_t)d_bgra & 3u))
        return fail(c, RT_ERR_INVALID_ARGUMENT, "framebuffer pointers must be 16/4/4-byte aligned");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    free_framebuffer(c);
    c->owns_fb = false;
    c->d_accum = d_accum;
    c->d_counts = d_counts;
    c->d_bgra = d_bgra;
    c->width = width;
    c->height = height;
    return RT_OK;
}

// Everything rt_render_frame checks and derives before it touches the device: the kernels' argument
// blocks for this frame (output pointers, scratch and tile order still unset).
static int prepare_frame(RtContext* c, const RtCamera* camera, const RtPushConstants* pc, const RtParams* params, int32_t width,
                         int32_t height, RtParams& p, rtd::DeviceScene& sc, rtd::FrameArgs& fa) {
    if (!camera || !pc) return fail(c, RT_ERR_INVALID_ARGUMENT, "camera / push constants are null");
    if (!c->has_scene) return fail(c, RT_ERR_NO_SCENE, "rt_upload_scene has not been called");
    if (params) p = *params; else rt_default_params(&p);
    if (p.spp <= 0 || p.max_bounces < 0 || p.spp > 65536 || p.max_bounces > 65536)
        return fail(c, RT_ERR_INVALID_ARGUMENT, "spp / max_bounces out of range");
    if (p.accel != RT_ACCEL_FLAT && p.accel != RT_ACCEL_BVH) return fail(c, RT_ERR_INVALID_ARGUMENT, "unknown accel");
    if (p.skybox < RT_SKY_GREY || p.skybox > RT_SKY_BLACK) return fail(c, RT_ERR_INVALID_ARGUMENT, "unknown skybox");
    if (p.flags & RT_FLAG_RESERVED_16) return fail(c, RT_ERR_UNSUPPORTED, "flag 16 selected a pipeline that no longer exists");
    if (p.strip_rows > 0 && (p.strip_count < 1 || p.strip_index < 0 || p.strip_index >= p.strip_count))
        return fail(c, RT_ERR_INVALID_ARGUMENT, "bad strip partition");
    // The acceleration structure was built for the uploaded element counts; the loop bounds
    // the shader would take from the push constants must be those counts.
    if (pc->total_spheres != c->n_spheres || pc->total_triangles != c->n_triangles || pc->total_meshes != c->n_meshes)
        return fail(c, RT_ERR_INVALID_ARGUMENT, "push-constant primitive counts differ from the uploaded scene");
    if (pc->total_lights < 0 || pc->total_lights > c->n_lights)
        return fail(c, RT_ERR_INVALID_ARGUMENT, "push-constant light count exceeds the uploaded lights");

    sc.prims_bvh = (const float4*)c->d_prims_bvh;
    sc.prims_flat = (const float4*)c->d_prims_flat;
    sc.nodes = (const float4*)c->d_nodes;
    sc.materials = (const float4*)c->d_materials;
    sc.lights = (const float4*)c->d_lights;
    sc.triangles = (const float4*)c->d_triangles;
    sc.n_prims = c->n_prims;
    sc.n_spheres = c->n_spheres;
    sc.n_triangles = c->n_triangles;

    memcpy(fa.view_inv, camera->viewInv, 64);
    memcpy(fa.cam_pos, camera->position, 12);
    fa.tan_half_fov = camera->tanHalfFOV;
    // uint(pc.time): GLSL leaves negative / out-of-range conversions undefined; they map to 0 here
    fa.time_u = (pc->time >= 0.0f && pc->time < 4294967296.0f) ? (uint32_t)pc->time : 0u;
    fa.frame_count = pc->frameCount;
    fa.total_lights = pc->total_lights;
    fa.lights_strength_sum = pc->lights_strength_sum;
    fa.reset = pc->reset_frame_accumulation ? 1u : 0u;
    fa.width = width;
    fa.height = height;
    fa.spp = p.spp;
    fa.max_bounces = p.max_bounces;
    fa.strip_rows = p.strip_rows > 0 ? p.strip_rows : 0;
    fa.strip_count = p.strip_rows > 0 ? p.strip_count : 1;
    fa.strip_index = p.strip_rows > 0 ? p.strip_index : 0;
    int owned = height;
    if (p.strip_rows > 0) {
        owned = 0;
        int n_strips = (height + p.strip_rows - 1) / p.strip_rows;
        for (int s = p.strip_index; s < n_strips; s += p.strip_count) {
            int rows = height - s * p.strip_rows;
            owned += rows < p.strip_rows ? rows : p.strip_rows;
        }
    }
    fa.owned_rows = owned;
    const int block_px = 1 << rtd::kBlockPixLog2;
    fa.tiles_x = (width + block_px - 1) / block_px;
    fa.total_slots = ((uint32_t)fa.tiles_x * (uint32_t)((owned + block_px - 1) / block_px)) << rtd::kBlockSlotsLog2;
    fa.flags = p.flags;
    fa.exit_quota_pct = c->opt.exit_quota_pct;
    fa.accel = p.accel;
    fa.skybox = p.skybox;
    fa.accum = nullptr;
    fa.counts = nullptr;
    fa.bgra = nullptr;
    fa.primary_ids = nullptr;
    fa.path_hash = nullptr;
    fa.counters = c->d_counters;
    fa.abort_flag = c->h_abort;
    fa.tile_order = nullptr;
    fa.tile_cost = nullptr;
    fa.n_frames = 1;
    fa.n_xcds = 1;
    memset(fa.frame_counts, 0, sizeof fa.frame_counts);
    fa.xcd_next = nullptr;
    fa.block_done = nullptr;
    return RT_OK;
}

// 16-bit sample / bounce counters in LDS; light and material index share a word of the slot record
static bool pooled_fits(const RtContext* c, const RtParams& p) {
    return p.spp <= 65535 && p.max_bounces <= 65534 && c->n_lights <= 32768 && c->n_materials <= 65536;
}

// Paths in flight per pooled workgroup.  A single frame of a scene far beyond the L2 caches (the 84 MB of the
// 1.07 M-triangle scene against 8 x 4 MB) runs faster with fewer (measured while a pool held 1392 slots: 1392 slots 12.35 ms, 1152 11.9, 896 11.75):
// its long rays make the last generation of paths expensive, and fewer slots make that generation smaller.  A launch
// of several frames has one such generation for all of them and wants every slot (4 frames: 896 slots 10.84 ms per
// frame, 1152 10.25, 1392 10.10), as do the scenes that fit the caches.
static int pool_slots(const RtContext* c, int n_frames) {
    if (c->opt.pool_slots > 0) return c->opt.pool_slots;
    const size_t scene_bytes = (size_t)c->n_prims * 48u + (size_t)c->bvh_nodes * 64u;
    return (scene_bytes > ((size_t)16 << 20) && n_frames == 1) ? 896 : 0;
}

// Children per node of the tree the pooled kernel walks.  The four-wide tree takes about half the node visits
// (each a whole cache line) for about twice the arithmetic per visit; measured faster on every workload (1080p:
// small scene 8.53 -> 8.09 ms, 1.07 M triangles 14.95 -> 13.2 ms), so it is used whenever its stack fits the
// overflow area on offer.
static int pooled_bvh_width(const RtContext* c, int overflow_words) {
    if (c->wide_nodes <= 0 || rtd::pool_overflow_for(c->wide_stack_need) > overflow_words) return 2;
    return c->opt.bvh_width != 0 ? c->opt.bvh_width : 4;
}
// Node format of the four-wide tree for the pooled kernel.  The 80-byte nodes (binary16 boxes relative to a point of
// the node) take five 16-byte loads per visit instead of seven and three more multiply-adds: where the tree does not
// fit the caches the fetch path is what the vector pipes wait for (1.07 M triangles, 16 frames per launch: L1 / texture
// path 71 % busy, 80 % of the issue slots used -> 60 % / 87 %, 9.11 -> 8.44 ms per frame; a frame alone 10.96 -> 10.56), where
// it does fit they are pure cost.  Fields of n x n teapots at 1080p (tools/field_time.py, ms per frame, 128- / 80-byte
// nodes): 0.5 MB of scene 3.54 / 3.61, 1.9 MB 4.03 / 4.02, 4.3 MB 4.49 / 4.44, 7.7 MB 4.94 / 4.83, 12 MB 5.49 / 5.34, 17 MB 6.66 / 6.46,
// 31 MB 7.78 / 7.45: they pay from about the size of one XCD's L2 (4 MB).  Same bits either way.
static bool pooled_half_nodes(const RtContext* c) {
    if (!c->d_nodes_wide16) return false;
    if (c->opt.bvh_node_format != 0) return c->opt.bvh_node_format == 16;
    const size_t scene_bytes = (size_t)c->n_prims * 48u + (size_t)c->bvh_nodes * 64u;
    return scene_bytes > ((size_t)4 << 20);
}
// Search steps of a pooled wave between two looks at the queues: about as many as a ray takes, so that few
// lanes sit finished through the rest of a burst.  Four-wide tree, 1080p: the small scenes' rays take 6 lane steps
// in 7-8 wave steps (burst 6: 7.08 ms, 7: 7.00, 8: 7.01, 9: 7.07, 10: 7.11, 12: 7.28 on the teapot scene; 8 also
// best at 4K and on the star scene), the 1.07 M-triangle scene's 16 (burst 12: 11.70 ms, 16: 11.70, 20: 11.69).
static int pool_burst(const RtContext* c, int width) {
    if (c->opt.pool_burst > 0) return c->opt.pool_burst;
    const size_t scene_bytes = (size_t)c->n_prims * 48u + (size_t)c->bvh_nodes * 64u;
    return (width == 4 && scene_bytes <= ((size_t)16 << 20)) ? 8 : 16;
}
// Waves of a pooled workgroup (of 16) that only shade.  About the share of the wave cycles that is shading: the scenes
// that fit the caches spend a quarter of them there (1080p, per frame: teapot scene 6.94 -> 6.77 ms with three such waves,
// star scene 2.26 -> 2.22, an eighth of the teapot frame 3.33 -> 3.13; fused launches 6.74 -> 6.68 and 1.95 -> 1.93 with
// four, an eighth 1.33 -> 1.14), the 1.07 M-triangle scene a tenth (two waves: 11.61 -> 11.28, fused 9.83 -> 9.49; four
// cost it 3 %).
static int pool_shader_waves(const RtContext* c, int n_frames) {
    if (c->opt.pool_shader_waves > 0) return c->opt.pool_shader_waves >= 9 ? 0 : c->opt.pool_shader_waves;
    const size_t scene_bytes = (size_t)c->n_prims * 48u + (size_t)c->bvh_nodes * 64u;
    if (scene_bytes > ((size_t)16 << 20)) return 2;
    return n_frames > 1 ? 4 : 3;
}
// Queue entries that make a shading job.  A full wave's worth was best while every wave both searched and shaded (a
// job of 48 cost the teapot frame 1 %, one of 40 3 %); a wave that only shades loses nothing by starting early: with
// shading waves 48 is worth 0.3 % on the scenes that fit the caches (32 / 40 / 48 / 64: 6.76 / 6.76 / 6.77 / 6.78 ms, 4K frames
// 7.82 / 7.82 / 7.82 / 7.85), and the 1.07 M-triangle scene, whose searches are long and whose queues fill slowly, takes
// 11.19 / 11.26 / 11.30 / 11.33 ms.
static int pool_job_min(const RtContext* c, int n_frames) {
    if (c->opt.pool_job_min > 0) return c->opt.pool_job_min;
    if (pool_shader_waves(c, n_frames) == 0) return 64;
    const size_t scene_bytes = (size_t)c->n_prims * 48u + (size_t)c->bvh_nodes * 64u;
    return scene_bytes > ((size_t)16 << 20) ? 32 : 48;
}
// Stack overflow words per lane the pooled kernel is given by rt_render_frame (which may allocate).
static int pooled_overflow_words(const RtContext* c) {
    if (c->opt.bvh_width == 2 || c->wide_nodes <= 0) return rtd::pool_overflow_per_lane();
    return rtd::pool_overflow_for(c->wide_stack_need);
}

// Clears the frame's counters and launches its kernel between the context's two timing events.
// `pool` must hold pooled_pool_bytes(grid, overflow_words) bytes when the pooled kernel runs.
static int launch_frame(RtContext* c, const rtd::DeviceScene& sc_in, const rtd::FrameArgs& fa, const RtParams& p, int pipeline, unsigned grid,
                        void* pool, int overflow_words) {
    RT_HIP(c, hipMemsetAsync(c->d_counters, 0, 32, c->stream));
    RT_HIP(c, hipMemsetAsync(c->d_counters + 10, 0, 152, c->stream));
    RT_HIP(c, hipEventRecord(c->ev_begin, c->stream));
    if (fa.total_slots > 0) {
        const bool stats = (p.flags & RT_FLAG_TRAVERSAL_STATS) != 0;
        if (pipeline == RT_PIPELINE_POOLED) {
            rtd::PathPool pp;
            carve_pooled_pool(pool, grid, overflow_words, pp);
            rtd::DeviceScene sc = sc_in;
            int accel = p.accel;
            const int width = pooled_bvh_width(c, overflow_words);
            c->last_node_bytes = p.accel == RT_ACCEL_BVH ? 64 : 0;
            if (accel == RT_ACCEL_BVH && width == 4) {
                const bool half = pooled_half_nodes(c);
                accel = half ? 3 : 2;
                sc.nodes = (const float4*)(half ? c->d_nodes_wide16 : c->d_nodes_wide);
                c->last_node_bytes = half ? 80 : 128;
            }
            c->last_width = p.accel == RT_ACCEL_BVH ? width : 0;
            RT_HIP(c, rtd::launch_pool(sc, fa, pp, accel, stats, grid, c->opt.refill_min, pool_burst(c, width), pool_job_min(c, fa.n_frames), pool_slots(c, fa.n_frames),
                                       pool_shader_waves(c, fa.n_frames), c->stream));
        } else {
            // the single kernel keeps a stack of rtd::kStackDepth entries per lane in LDS: the wide tree where that holds it
            rtd::DeviceScene sc = sc_in;
            int accel = p.accel;
            if (accel == RT_ACCEL_BVH && c->wide_nodes > 0 && c->wide_stack_need < (int)rtd::kStackDepth && c->opt.bvh_width != 2) {
                accel = 2;
                sc.nodes = (const float4*)c->d_nodes_wide;
            }
            c->last_width = p.accel == RT_ACCEL_BVH ? (accel == 2 ? 4 : 2) : 0;
            c->last_node_bytes = p.accel == RT_ACCEL_BVH ? (accel == 2 ? 128 : 64) : 0;
            RT_HIP(c, rtd::launch_render(sc, fa, accel, stats, c->cu_count, c->stream));
        }
    }
    RT_HIP(c, hipEventRecord(c->ev_end, c->stream));
    c->timed = true;
    c->last_pipeline = pipeline;
    c->last_frames = fa.n_frames;
    return RT_OK;
}

// rt_render_frame (n_frames == 1) and the fused form of rt_render_frames (2..16 frames of one camera in one launch of
// the pooled kernel; the caller has checked that the frames differ in frameCount only and that the pooled kernel applies).
static int render_impl(RtContext* c, const RtCamera* camera, const RtPushConstants* pc, int n_frames, const RtParams* params) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!c->d_accum && c->has_scene && camera && pc) return fail(c, RT_ERR_NO_FRAMEBUFFER, "rt_resize / rt_bind_framebuffer has not been called");
    RtParams p;
    rtd::DeviceScene sc;
    rtd::FrameArgs fa;
    int rc = prepare_frame(c, camera, pc, params, c->width, c->height, p, sc, fa);
    if (rc != RT_OK) return rc;
    fa.accum = (float4*)c->d_accum;
    fa.counts = (int32_t*)c->d_counts;
    fa.bgra = (uint32_t*)c->d_bgra;

    RT_HIP(c, hipSetDevice(c->device));
    // Tile order: each frame records what its tiles cost, and the next frame of the same size
    // and partition hands the dear tiles out first (same pixels, same results, shorter tail).
    const uint32_t n_tiles = fa.total_slots >> rtd::kBlockSlotsLog2;  // blocks, the unit of the order
    const bool feedback = !(p.flags & RT_FLAG_FIXED_TILE_ORDER) && n_tiles > 0 && n_tiles <= rtd::tile_order_max_tiles();
    if (feedback) {
        const int32_t key[6] = {fa.width, fa.height, fa.strip_rows, fa.strip_count, fa.strip_index, (int32_t)n_tiles};
        if (n_tiles > c->tile_capacity) {
            RT_HIP(c, hipStreamSynchronize(c->stream));
            if (c->d_tile_order) (void)hipFree(c->d_tile_order);
            if (c->d_tile_cost) (void)hipFree(c->d_tile_cost);
            c->d_tile_order = c->d_tile_cost = nullptr;
            c->tile_capacity = 0;
            c->order_ready = false;
            RT_HIP(c, hipMalloc((void**)&c->d_tile_order, (size_t)n_tiles * 4));
            RT_HIP(c, hipMalloc((void**)&c->d_tile_cost, (size_t)n_tiles * 4));
            if (!c->d_tile_aux) {
                RT_HIP(c, hipMalloc((void**)&c->d_tile_aux, rtd::tile_order_aux_bytes()));
                RT_HIP(c, hipMemsetAsync(c->d_tile_aux, 0, 256, c->stream));
            }
            c->tile_capacity = n_tiles;
        }
        if (!c->order_ready || memcmp(key, c->order_key, sizeof key) != 0) {
            c->order_ready = false;
            memcpy(c->order_key, key, sizeof key);
            RT_HIP(c, hipMemsetAsync(c->d_tile_cost, 0, (size_t)n_tiles * 4, c->stream));
        }
        fa.tile_cost = c->d_tile_cost;
        if (c->order_ready) fa.tile_order = c->d_tile_order;
    }
    if (p.flags & (RT_FLAG_PRIMARY_IDS | RT_FLAG_PATH_HASH)) {
        rc = alloc_debug_buffers(c);
        if (rc != RT_OK) return rc;
        if (p.flags & RT_FLAG_PRIMARY_IDS) fa.primary_ids = c->d_primary_ids;
        if (p.flags & RT_FLAG_PATH_HASH) fa.path_hash = c->d_path_hash;
    }

    // which schedule of the per-path program runs this frame (see rt_abi.h)
    int pipeline = RT_PIPELINE_SINGLE;
    int trial = -1;  // 0 / 1: this frame is the timed pooled / single trial of its configuration
    if (n_frames > 1) {
        pipeline = RT_PIPELINE_POOLED;
        const size_t words = (size_t)rtd::kXcds * 32u + (size_t)n_tiles;
        if (words > c->batch_words) {
            RT_HIP(c, hipStreamSynchronize(c->stream));
            if (c->d_batch) (void)hipFree(c->d_batch);
            c->d_batch = nullptr;
            c->batch_words = 0;
            RT_HIP(c, hipMalloc((void**)&c->d_batch, words * 4));
            c->batch_words = words;
        }
        RT_HIP(c, hipMemsetAsync(c->d_batch, 0, words * 4, c->stream));
        fa.n_frames = n_frames;
        fa.n_xcds = c->n_xcds;
        for (int k = 0; k < n_frames; ++k) fa.frame_counts[k] = pc[k].frameCount;
        fa.xcd_next = c->d_batch;
        fa.block_done = c->d_batch + (size_t)rtd::kXcds * 32u;
    } else if (p.flags & RT_FLAG_POOLED) {
        if (!pooled_fits(c, p))
            return fail(c, RT_ERR_TOO_LARGE, "the pooled pipeline holds at most 65535 samples, 65534 bounces, 32768 lights and 65536 materials");
        pipeline = RT_PIPELINE_POOLED;
    } else if (!(p.flags & RT_FLAG_SINGLE_KERNEL) && p.accel == RT_ACCEL_BVH && pooled_fits(c, p) && fa.total_slots >= 16384u) {
        // Neither kernel wins everywhere (the pooled one on most frames, the single one on big scenes
        // with few samples): of a configuration's first frames, 0-2 run pooled (the third one timed,
        // once the tile order and the clocks have settled) and 3-4 single (the second one timed); later
        // frames run the faster of the two as soon as both timings have arrived - the events are only
        // queried, never waited for - and pooled until then.  Frames of a few thousand pixels are not worth it.
        const int32_t key[9] = {c->scene_generation, fa.width, fa.height, fa.strip_rows, fa.strip_count, fa.strip_index,
                                p.spp, p.max_bounces, feedback ? 1 : 0};
        if (memcmp(key, c->tune_key, sizeof key) != 0) {
            memcpy(c->tune_key, key, sizeof key);
            c->tune_stage = 0;
            c->tune_decided = false;
        }
        if (c->tune_stage <= 2) pipeline = RT_PIPELINE_POOLED;
        else if (c->tune_stage <= 4) pipeline = RT_PIPELINE_SINGLE;
        else {
            if (!c->tune_decided && hipEventQuery(c->ev_trial[0][1]) == hipSuccess && hipEventQuery(c->ev_trial[1][1]) == hipSuccess) {
                float pooled_ms = 0.0f, single_ms = 0.0f;
                if (hipEventElapsedTime(&pooled_ms, c->ev_trial[0][0], c->ev_trial[0][1]) == hipSuccess &&
                    hipEventElapsedTime(&single_ms, c->ev_trial[1][0], c->ev_trial[1][1]) == hipSuccess) {
                    c->tune_choice = (single_ms < pooled_ms) ? RT_PIPELINE_SINGLE : RT_PIPELINE_POOLED;
                    c->tune_decided = true;
                }
            }
            pipeline = c->tune_decided ? c->tune_choice : RT_PIPELINE_POOLED;
        }
        if (c->tune_stage == 2) trial = 0;
        if (c->tune_stage == 4) trial = 1;
        if (c->tune_stage < 5) c->tune_stage++;
    }
    unsigned grid = 0;
    if (pipeline == RT_PIPELINE_POOLED && fa.total_slots > 0) {
        RT_HIP(c, rtd::pool_grid(fa, p.accel == RT_ACCEL_BVH ? (pooled_bvh_width(c, pooled_overflow_words(c)) == 4 ? (pooled_half_nodes(c) ? 3 : 2) : 1) : 0, (p.flags & RT_FLAG_TRAVERSAL_STATS) != 0,
                                 n_frames > 1, pool_shader_waves(c, n_frames), c->cu_count, c->opt.pool_per_cu, &grid));
        rc = ensure_pool_bytes(c, pooled_pool_bytes(grid, pooled_overflow_words(c)));
        if (rc != RT_OK) return rc;
    }
    if (trial >= 0) RT_HIP(c, hipEventRecord(c->ev_trial[trial][0], c->stream));
    rc = launch_frame(c, sc, fa, p, pipeline, grid, c->d_pool, pooled_overflow_words(c));
    if (rc != RT_OK) return rc;
    if (trial >= 0) RT_HIP(c, hipEventRecord(c->ev_trial[trial][1], c->stream));
    if (n_frames > 1 && fa.total_slots > 0) RT_HIP(c, rtd::launch_batch_check(fa, c->stream));
    if (feedback && fa.total_slots > 0) {
        RT_HIP(c, rtd::launch_tile_order(c->d_tile_cost, c->d_tile_order, n_tiles, c->opt.tile_classes, c->d_tile_aux, nullptr, nullptr, c->stream));
        c->order_ready = true;
    }
    return RT_OK;
}

int rt_render_frame(RtContext* c, const RtCamera* camera, const RtPushConstants* pc, const RtParams* params) {
    return render_impl(c, camera, pc, 1, params);
}

int rt_render_frames(RtContext* c, const RtCamera* camera, const RtPushConstants* pcs, int32_t n_frames, const RtParams* params) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!camera || !pcs || n_frames < 1) return fail(c, RT_ERR_INVALID_ARGUMENT, "camera / push constants are null or no frames asked for");
    RtParams p;
    if (params) p = *params; else rt_default_params(&p);
    // One launch where the frames differ in frameCount only, none after the first resets the accumulation, the
    // pooled kernel applies and the image is neither small nor beyond 4 M pixels (measured: 1080p frames gain 4 % at
    // 16 spp x 4 bounces, 15 % at 4 spp x 20 bounces on the small scene, 8 % on the 1.07 M-triangle scene; 4K frames
    // of 4 spp lose 2.5 %: their drain is short against the frame, and a pixel's end costs six L2 atomics more);
    // else frame by frame, with the same result.
    bool fused = n_frames >= 2 && c->n_xcds >= 1 && p.accel == RT_ACCEL_BVH && !(p.flags & (RT_FLAG_SINGLE_KERNEL | RT_FLAG_PRIMARY_IDS | RT_FLAG_PATH_HASH)) &&
                 c->has_scene && c->d_accum && pooled_fits(c, p) && (long long)c->width * c->height <= (1ll << 22);
    for (int k = 1; k < n_frames && fused; ++k) {
        RtPushConstants a = pcs[0], b = pcs[k];
        a.frameCount = b.frameCount = 0;
        a.reset_frame_accumulation = 0;  // (the first frame may reset, the others must not)
        fused = memcmp(&a, &b, sizeof a) == 0;
    }
    if (fused) {
        const int block_px = 1 << rtd::kBlockPixLog2;
        int owned = c->height;
        if (p.strip_rows > 0 && p.strip_count >= 1) owned = c->height / p.strip_count;  // a lower bound is enough here
        const long long blocks = (long long)((c->width + block_px - 1) / block_px) * ((owned + block_px - 1) / block_px);
        fused = blocks >= 64;  // (measured down to 64 x 64 pixels: 2.27 -> 0.66 ms per frame of 16 spp; below that some XCDs would own nothing)
    }
    if (!fused) {
        for (int k = 0; k < n_frames; ++k) {
            const int rc = render_impl(c, camera, pcs + k, 1, &p);
            if (rc != RT_OK) return rc;
        }
        return RT_OK;
    }
    for (int k = 0; k < n_frames; k += rtd::kMaxBatchFrames) {
        const int n = n_frames - k < rtd::kMaxBatchFrames ? n_frames - k : rtd::kMaxBatchFrames;
        const int rc = render_impl(c, camera, pcs + k, n, &p);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}

int rt_render_frame_noalloc(RtContext* c, const RtCamera* camera, const RtPushConstants* pc, const RtParams* params, void* d_accum,
                            void* d_counts, void* d_bgra, int32_t width, int32_t height) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 27) || !d_accum || !d_counts || !d_bgra)
        return fail(c, RT_ERR_INVALID_ARGUMENT, "bad framebuffer");
    if (((uintptr_t)d_accum & 15u) || ((uintptr_t)d_counts & 3u) || ((uintptr_t)d_bgra & 3u))
        return fail(c, RT_ERR_INVALID_ARGUMENT, "framebuffer pointers must be 16/4/4-byte aligned");
    RtParams p;
    rtd::DeviceScene sc;
    rtd::FrameArgs fa;
    int rc = prepare_frame(c, camera, pc, params, width, height, p, sc, fa);
    if (rc != RT_OK) return rc;
    if (p.flags & (RT_FLAG_PRIMARY_IDS | RT_FLAG_PATH_HASH))
        return fail(c, RT_ERR_UNSUPPORTED, "the debug outputs live in buffers this call may not allocate; use rt_render_frame");
    const uint32_t n_tiles = fa.total_slots >> rtd::kBlockSlotsLog2;
    if (n_tiles > kScratchTiles)
        return fail(c, RT_ERR_TOO_LARGE, "rt_render_frame_noalloc renders at most 131072 tiles of 8x8 pixels (8.39 M pixels) per call");
    if (c->device >= kMaxDevices || g_scratch_ptr[c->device] == nullptr) return fail(c, RT_ERR_HIP, "no scratch memory on this device");
    fa.accum = (float4*)d_accum;
    fa.counts = (int32_t*)d_counts;
    fa.bgra = (uint32_t*)d_bgra;

    RT_HIP(c, hipSetDevice(c->device));
    unsigned char* const base = g_scratch_ptr[c->device];
    uint32_t* const stamp = (uint32_t*)base;
    uint32_t* const tile_aux = (uint32_t*)(base + kScratchStampBytes);  // its first word: zero at load, and after every call
    uint32_t* const tile_cost = (uint32_t*)(base + kScratchStampBytes + kScratchAuxBytes);
    uint32_t* const tile_order = tile_cost + kScratchTiles;
    void* const pool = base + kScratchStampBytes + kScratchAuxBytes + 2 * 4 * kScratchTiles;
    if (rtd::tile_order_aux_bytes() > kScratchAuxBytes) return fail(c, RT_ERR_HIP, "scratch layout out of date");

    // The tile order of the previous call is kept in the scratch array together with a stamp of what it
    // was made for; a kernel compares the stamp with this call's key and falls back to image order, so
    // that no host state is involved and a call of another size (or another context) cannot misuse it.
    const bool feedback = !(p.flags & RT_FLAG_FIXED_TILE_ORDER) && n_tiles > 0;
    const uint32_t key[8] = {0x7231746fu, (uint32_t)fa.width, (uint32_t)fa.height, (uint32_t)fa.strip_rows, (uint32_t)fa.strip_count,
                             (uint32_t)fa.strip_index, n_tiles, (uint32_t)c->opt.tile_classes};
    if (feedback) {
        RT_HIP(c, hipMemsetAsync(tile_cost, 0, (size_t)n_tiles * 4, c->stream));  // whatever the last call left
        RT_HIP(c, rtd::launch_tile_order_check(tile_order, n_tiles, stamp, key, c->stream));
        fa.tile_cost = tile_cost;
        fa.tile_order = tile_order;
    }
    int pipeline = RT_PIPELINE_SINGLE;
    if (p.flags & RT_FLAG_POOLED) {
        if (!pooled_fits(c, p))
            return fail(c, RT_ERR_TOO_LARGE, "the pooled pipeline holds at most 65535 samples, 65534 bounces, 32768 lights and 65536 materials");
        pipeline = RT_PIPELINE_POOLED;
    } else if (!(p.flags & RT_FLAG_SINGLE_KERNEL) && p.accel == RT_ACCEL_BVH && pooled_fits(c, p) && fa.total_slots >= 16384u) {
        pipeline = RT_PIPELINE_POOLED;  // no timed trial here: the pooled kernel wins on most frames
    }
    unsigned grid = 0;
    if (pipeline == RT_PIPELINE_POOLED && fa.total_slots > 0) {
        RT_HIP(c, rtd::pool_grid(fa, p.accel == RT_ACCEL_BVH ? (pooled_bvh_width(c, rtd::pool_overflow_per_lane()) == 4 ? (pooled_half_nodes(c) ? 3 : 2) : 1) : 0,
                                 (p.flags & RT_FLAG_TRAVERSAL_STATS) != 0, false, pool_shader_waves(c, 1), c->cu_count, c->opt.pool_per_cu, &grid));
        const unsigned fits = (unsigned)(kScratchPoolBytes / pooled_pool_bytes(1, rtd::pool_overflow_per_lane()));
        grid = grid > fits ? fits : grid;  // fewer workgroups than CUs still render the whole frame
    }
    // (the scratch array's stack overflow area is the smallest one: a four-wide tree that needs more is not used here)
    rc = launch_frame(c, sc, fa, p, pipeline, grid, pool, rtd::pool_overflow_per_lane());
    if (rc != RT_OK) return rc;
    if (feedback && fa.total_slots > 0)
        RT_HIP(c, rtd::launch_tile_order(tile_cost, tile_order, n_tiles, c->opt.tile_classes, tile_aux, stamp, key, c->stream));
    return RT_OK;
}

int rt_set_option(RtContext* c, int option, int value) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    struct { int option; int* field; int lo, hi; } table[] = {
        {RT_OPT_EXIT_QUOTA_PCT, &c->opt.exit_quota_pct, 1, 100}, {RT_OPT_REFILL_MIN, &c->opt.refill_min, 1, 64},
        {RT_OPT_POOL_BURST, &c->opt.pool_burst, 0, 1024},        {RT_OPT_POOL_JOB_MIN, &c->opt.pool_job_min, 0, 64},
        {RT_OPT_TILE_CLASSES, &c->opt.tile_classes, 1, 64},      {RT_OPT_POOL_PER_CU, &c->opt.pool_per_cu, 0, 8},
        {RT_OPT_BVH_MAX_LEAF, &c->opt.bvh_max_leaf, 1, 8},       {RT_OPT_POOL_SLOTS, &c->opt.pool_slots, 0, rtd::pool_slots_per_block()},
        {RT_OPT_BVH_TREELET, &c->opt.bvh_treelet, 0, 16},        {RT_OPT_BVH_WIDTH, &c->opt.bvh_width, 0, 4},
        {RT_OPT_POOL_SHADER_WAVES, &c->opt.pool_shader_waves, 0, 9}, {RT_OPT_BVH_NODE_FORMAT, &c->opt.bvh_node_format, 0, 32}};
    for (const auto& t : table) {
        if (t.option != option) continue;
        if (value < t.lo || value > t.hi || (option == RT_OPT_BVH_WIDTH && (value == 1 || value == 3)) ||
            (option == RT_OPT_BVH_NODE_FORMAT && value != 0 && value != 16 && value != 32))
            return fail(c, RT_ERR_INVALID_ARGUMENT, "option value out of range");
        *t.field = value;
        return RT_OK;
    }
    return fail(c, RT_ERR_INVALID_ARGUMENT, "unknown option");
}

int rt_resolve(RtContext* c) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!c->d_accum) return fail(c, RT_ERR_NO_FRAMEBUFFER, "no framebuffer");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, rtd::launch_resolve((float4*)c->d_accum, (int32_t*)c->d_counts, (uint32_t*)c->d_bgra, c->width * c->height,
                                  c->stream));
    return RT_OK;
}

int rt_read_output_bgra8(RtContext* c, void* dst) {
    if (!c || !dst) return RT_ERR_INVALID_ARGUMENT;
    if (!c->d_bgra) return fail(c, RT_ERR_NO_FRAMEBUFFER, "no framebuffer");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpy(dst, c->d_bgra, (size_t)c->width * c->height * 4, hipMemcpyDeviceToHost));
    return rtapi::check_abandoned(c);
}

int rt_read_accum(RtContext* c, float* accum, int32_t* counts) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!c->d_accum) return fail(c, RT_ERR_NO_FRAMEBUFFER, "no framebuffer");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    size_t n = (size_t)c->width * c->height;
    if (accum) RT_HIP(c, hipMemcpy(accum, c->d_accum, n * 16, hipMemcpyDeviceToHost));
    if (counts) RT_HIP(c, hipMemcpy(counts, c->d_counts, n * 4, hipMemcpyDeviceToHost));
    return rtapi::check_abandoned(c);
}

int rt_read_debug(RtContext* c, int32_t* primary_ids, uint32_t* path_hash) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if ((primary_ids && !c->d_primary_ids) || (path_hash && !c->d_path_hash))
        return fail(c, RT_ERR_INVALID_ARGUMENT, "no frame was rendered with RT_FLAG_PRIMARY_IDS / RT_FLAG_PATH_HASH");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    size_t n = (size_t)c->width * c->height;
    if (primary_ids) RT_HIP(c, hipMemcpy(primary_ids, c->d_primary_ids, n * 4, hipMemcpyDeviceToHost));
    if (path_hash) RT_HIP(c, hipMemcpy(path_hash, c->d_path_hash, n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_get_stats(RtContext* c, RtStats* out) {
    if (!c || !out) return RT_ERR_INVALID_ARGUMENT;
    memset(out, 0, sizeof *out);
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    uint32_t h[48] = {0};
    RT_HIP(c, hipMemcpy(h, c->d_counters, 192, hipMemcpyDeviceToHost));
    memcpy(&out->rays, &h[2], 8);
    memcpy(&out->nodes_visited, &h[4], 8);
    memcpy(&out->prims_tested, &h[6], 8);
    memcpy(&out->total_rays, &h[8], 8);
    memcpy(&out->cycles_raygen, &h[10], 8);
    memcpy(&out->cycles_search, &h[12], 8);
    memcpy(&out->cycles_shade, &h[14], 8);
    memcpy(&out->lane_iterations, &h[16], 8);
    memcpy(&out->wave_iterations, &h[18], 8);
    memcpy(out->cycles_shade_parts, &h[20], 40);
    memcpy(out->pool_job_cycles, &h[32], 32);
    memcpy(&out->busy_lane_iterations, &h[40], 8);
    memcpy(&out->cycles_steps, &h[42], 8);
    out->search_queue_peak = (int32_t)h[44];
    out->search_queue_size = (int32_t)h[45];
    memcpy(&out->stack_overflow_steps, &h[46], 8);
    if (c->timed) RT_HIP(c, hipEventElapsedTime(&out->kernel_ms, c->ev_begin, c->ev_end));
    out->bvh_nodes = c->bvh_nodes;
    out->bvh_depth = c->bvh_depth;
    out->total_prims = c->n_prims;
    out->pipeline = c->last_pipeline;
    out->bvh_width = c->last_width;
    out->bvh_node_bytes = c->last_node_bytes;
    out->frames = c->last_frames;
    return rtapi::check_abandoned(c);
}

int rt_get_device_buffers(RtContext* c, void** d_accum, void** d_counts, void** d_bgra8) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!c->d_accum) return fail(c, RT_ERR_NO_FRAMEBUFFER, "no framebuffer");
    if (d_accum) *d_accum = c->d_accum;
    if (d_counts) *d_counts = c->d_counts;
    if (d_bgra8) *d_bgra8 = c->d_bgra;
    return RT_OK;
}

int rt_debug_math(RtContext* c, int fn, const float* x, const float* y, float* out, int32_t n) {
    if (!c || !x || !y || !out || n < 0) return RT_ERR_INVALID_ARGUMENT;
    if (n == 0) return RT_OK;
    RT_HIP(c, hipSetDevice(c->device));
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    size_t bytes = (size_t)n * 4;
    hipError_t e = hipMalloc((void**)&dx, bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&dy, bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&dout, bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(dx, x, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dy, y, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = rtd::launch_math_probe(fn, dx, dy, dout, n, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (dx) (void)hipFree(dx);
    if (dy) (void)hipFree(dy);
    if (dout) (void)hipFree(dout);
    if (e != hipSuccess) return fail(c, RT_ERR_HIP, std::string("rt_debug_math: ") + hipGetErrorString(e));
    return RT_OK;
}

void* rt_get_stream(RtContext* c) { return c ? (void*)c->stream : nullptr; }

int rt_sync(RtContext* c) {
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return rtapi::check_abandoned(c);
}

const char* rt_last_error(RtContext* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

}  // extern "C"
