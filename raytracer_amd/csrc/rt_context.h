// rt_context.h — the RtContext behind the C ABI of include/rt_abi.h, shared by rt_api.hip (one
// device) and rt_group.hip (several devices combined over RCCL).  Internal: not installed.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "host/bvh_builder.h"

// Tuning knobs (rt_set_option; defaults may be overridden once, at rt_create, by the environment
// variables named in rt_api.hip).
struct RtOptions {
    int exit_quota_pct = 75;  // single kernel: the search loop returns once this share of the busy lanes has a result
    int refill_min = 16;      // pooled kernel: idle lanes of a searching wave before it fetches new rays
    int pool_burst = 0;       // pooled kernel: search steps between two looks at the queues (0: the library's choice by tree and scene size)
    int pool_job_min = 0;     // pooled kernel: queue entries that make a shading job (0: the library's choice by scene size)
    int pool_slots = 0;       // pooled kernel: paths in flight per workgroup (0: the library's choice by scene size)
    int pool_shader_waves = 0;  // pooled kernel: waves per workgroup that only shade (0: the library's choice by scene size; 9: none)
    int tile_classes = 16;    // cost classes of the frame-to-frame tile order
    int pool_per_cu = 0;      // pooled kernel: cap on workgroups per CU (0: what fits)
    int bvh_max_leaf = 4;     // builder: most primitives per leaf (takes effect at the next rt_upload_scene)
    int bvh_width = 0;        // pooled kernel: children per node of the tree it walks (2, 4; 0: four when its stack fits)
    int bvh_treelet = 0;      // node layout: levels per treelet (0 / 1: depth-first order; next rt_upload_scene)
    int bvh_node_format = 0;  // pooled kernel, four-wide tree: 32 = 128-byte nodes, 16 = 80-byte nodes with binary16 boxes; 0: by scene size
};

struct RtContext {
    int device = 0;
    int cu_count = 0;
    int n_xcds = 0;  // XCDs the device's workgroups are spread over (0: unknown, several frames are never fused into one launch)
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool timed = false;
    int32_t last_pipeline = 0;
    int32_t last_frames = 1;  // frames rendered by the last launch (rt_render_frames)
    int32_t last_width = 0;  // children per node of the tree the last frame's kernel walked (0: the flat loops)
    int32_t last_node_bytes = 0;  // bytes per node of that tree
    RtOptions opt;

    // scene
    bool has_scene = false;
    void* d_prims_bvh = nullptr;
    void* d_prims_flat = nullptr;
    void* d_nodes = nullptr;
    void* d_nodes_wide = nullptr;  // the four-wide form of the same tree (pooled kernel)
    void* d_nodes_wide16 = nullptr;  // ... in 80-byte nodes (null if the scene's coordinates do not fit binary16 offsets)
    void* d_materials = nullptr;
    void* d_lights = nullptr;
    void* d_triangles = nullptr;
    int32_t n_prims = 0, n_spheres = 0, n_triangles = 0, n_meshes = 0, n_lights = 0, n_materials = 0;
    int32_t bvh_nodes = 0, bvh_depth = 0;
    int32_t wide_nodes = 0, wide_stack_need = 0;

    // framebuffer
    int32_t width = 0, height = 0;
    bool owns_fb = false;
    void* d_accum = nullptr;
    void* d_counts = nullptr;
    void* d_bgra = nullptr;
    int32_t* d_primary_ids = nullptr;
    uint32_t* d_path_hash = nullptr;
    uint32_t* d_counters = nullptr;  // 48 x u32, see FrameArgs::counters
    uint32_t* h_abort = nullptr;     // pinned host word the pooled kernel sets when it gave up (FrameArgs::abort_flag)
    // tile order fed back from frame to frame (see FrameArgs::tile_order)
    uint32_t* d_tile_order = nullptr;
    uint32_t* d_tile_cost = nullptr;
    uint32_t* d_tile_aux = nullptr;  // work area of the tile order kernels
    uint32_t* d_batch = nullptr;     // several frames in one launch: per-XCD slot counters and per-block finished counts
    size_t batch_words = 0;
    uint32_t tile_capacity = 0;
    bool order_ready = false;                   // d_tile_order holds the order made from the last frame
    int32_t order_key[6] = {0, 0, 0, 0, 0, 0};  // width, height, strip_rows, strip_count, strip_index, tiles it was made for
    // choice between the pooled and the single kernel by measurement (see rt_render_frame)
    int32_t scene_generation = 0;
    int32_t tune_key[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int tune_stage = 0;  // frames of this configuration so far (saturating): 0-2 pooled, 3-4 single, then the choice
    bool tune_decided = false;
    hipEvent_t ev_trial[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [pooled, single][begin, end] of the timed trial frames
    int tune_choice = RT_PIPELINE_POOLED;

    // device memory of the pooled pipeline (allocated on first use, grown when needed)
    void* d_pool = nullptr;
    size_t pool_bytes = 0;

    // row-strip exchange (rt_strip_pack / rt_strip_unpack): nothing owned here, see rt_group.hip

    std::string err;
};

namespace rtapi {
// A scene as rt_upload_scene prepares it on the host, ready to be copied to any number of devices (the tree build is
// the expensive part: seconds for a million triangles).  materials / lights / triangles are borrowed from the caller.
struct BuiltScene {
    std::vector<rt::PrimRecord> flat;    // the reference's loop order
    rt::HostBvh bvh;                     // binary tree; bvh.prims = the records in leaf order
    std::vector<rt::BvhNode> dev_nodes;  // bvh.nodes with children as cursor words
    rt::HostWideBvh wide;                // four-wide collapse of the same tree
    const RtMaterial* materials = nullptr; int32_t n_materials = 0;
    const RtLight* lights = nullptr;       int32_t n_lights = 0;
    const RtTriangle* triangles = nullptr; int32_t n_triangles = 0;
    int32_t n_spheres = 0, n_meshes = 0;
};
int build_scene(const RtOptions& opt, const RtSphere* spheres, int32_t n_spheres, const RtMaterial* materials, int32_t n_materials,
                const RtLight* lights, int32_t n_lights, const RtTriangle* triangles, int32_t n_triangles, const RtVertex* vertices,
                int32_t n_vertices, const uint32_t* indices, int32_t n_indices, const RtMeshInfo* meshes, int32_t n_meshes, BuiltScene& out,
                std::string& err);
int upload_built(RtContext* c, const BuiltScene& b);
int fail(RtContext* ctx, int code, const std::string& msg);
// after a stream synchronise: RT_ERR_HIP (once) if a pooled frame was abandoned since the last report
int check_abandoned(RtContext* c);
}  // namespace rtapi

#define RT_HIP(ctx, call)                                                                                   \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return rtapi::fail((ctx), RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));       \
    } while (0)
