// rt_kernels.h — argument blocks and launch entry points shared by rt_kernels.hip (device
// code) and rt_api.hip (host API).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtd {

// Read-only scene on the device.  Records and nodes are addressed as float4 (16-byte loads).
struct DeviceScene {
    const float4* prims_bvh;   // 3 float4 per primitive, BVH leaf order
    const float4* prims_flat;  // 3 float4 per primitive, the reference's loop order
    const float4* nodes;       // 4 float4 per BVH node, node 0 = root (four-wide search: 8 float4 per node, or 5 in the 80-byte form)
    const float4* materials;   // 5 float4 per material (raytracer.comp:54-82)
    const float4* lights;      // 3 float4 per light    (raytracer.comp:47-52)
    const float4* triangles;   // 4 float4 per triangle (raytracer.comp:90-96), for stored normals and light sampling
    int32_t n_prims;
    int32_t n_spheres;
    int32_t n_triangles;
};

struct FrameArgs {
    // camera (raytracer.comp:27-35): only the fields the shader reads
    float view_inv[16];
    float cam_pos[3];
    float tan_half_fov;
    // push constants (raytracer.comp:127-137)
    uint32_t time_u;  // uint(pc.time)
    uint32_t frame_count;
    int32_t total_lights;
    float lights_strength_sum;
    uint32_t reset;
    // image and per-frame parameters
    int32_t width, height;
    int32_t spp, max_bounces;
    int32_t strip_rows, strip_count, strip_index;  // strip_rows <= 0: all rows
    int32_t owned_rows;                             // rows this context renders
    int32_t tiles_x;                                // blocks per row: ceil(width / (8 << kBlockTilesLog2))
    uint32_t total_slots;                           // blocks * pixels per block (blocks cover the owned rows)
    uint32_t flags;
    int32_t exit_quota_pct;  // the search loop returns once this share of the wave's busy lanes has finished
    int32_t accel;           // RT_ACCEL_*: which record array the search results index
    int32_t skybox;          // RT_SKY_*
    // outputs
    float4* accum;
    int32_t* counts;
    uint32_t* bgra;
    int32_t* primary_ids;  // may be null
    uint32_t* path_hash;   // may be null
    // device counters: [0] next pixel slot, [2..3] rays (u64), [4..5] nodes visited, [6..7] prims tested
    // (cleared per frame), [8..9] rays of all frames (never cleared); instrumented builds only (cleared
    // per frame): [10..15] wave cycles in refill+raygen / search / shading, [16..17] per-lane traversal
    // iterations, [18..19] per-wave traversal iterations, [20..29] wave cycles of the five parts of
    // the continuation (pooled pipeline: [20..27] items << 32 | jobs per shading queue, [28..29] deep
    // stack pushes, [32..39] wave cycles per shading queue, [40..43] lanes holding a ray / cycles inside the bursts,
    // [44] the search queue's peak, [45] its size, [46..47] node visits that pushed into the device-memory stack overflow)
    uint32_t* counters;
    // Set to 1 (never cleared by a kernel) when the pooled kernel gave up waiting inside a workgroup and the
    // frame is incomplete; host-visible memory, so that the host sees it after a plain stream synchronise.
    uint32_t* abort_flag;
    // Order in which the image's pixel blocks are handed out: slot block k renders image block
    // tile_order[k] (null: k itself).  tile_cost (may be null) receives, per image block, the path
    // segments traced for it this frame; the next frame's order is made from it.
    const uint32_t* tile_order;
    uint32_t* tile_cost;
    // Several frames of one camera in one launch of the pooled kernel (n_frames > 1; up to kMaxBatchFrames): frame k
    // is seeded with frame_counts[k], `reset` applies to frame 0.  The image blocks are shared out among the
    // XCDs by their position in the order (position mod 8), the same in every frame of the launch, so that
    // a pixel's frames follow each other on ONE XCD, whose L2 keeps them coherent (n_xcds: how many there are, probed
    // once per device); xcd_next[32 * x] is XCD
    // x's next slot (frame-major), block_done[b] counts the pixels of image block b finished so far
    // (a frame's pixel ends after all the block's pixels of the frame before).  Both zeroed before the launch.
    int32_t n_frames;
    int32_t n_xcds;  // XCDs the launch's workgroups are spread over (their XCC_ID = 0 .. n_xcds - 1), <= kXcds
    uint32_t frame_counts[32];
    uint32_t* xcd_next;
    uint32_t* block_done;
};
enum { kMaxBatchFrames = 32, kXcds = 8 };

enum { kBlock = 256, kStackDepth = 32 };
// Pixels are handed out in blocks of (1 << kBlockTilesLog2)^2 tiles of 8x8 pixels; the block is
// the unit of the frame-to-frame order, a tile (64 consecutive slots) stays one wave's worth.
#ifndef RT_BLOCK_TILES_LOG2
#define RT_BLOCK_TILES_LOG2 0
#endif
enum { kBlockTilesLog2 = RT_BLOCK_TILES_LOG2, kBlockPixLog2 = 3 + kBlockTilesLog2, kBlockSlotsLog2 = 2 * kBlockPixLog2 };

// What a slot of the pooled pipeline waits for (the kind bits of a queue entry are derived from it).
enum : uint32_t {
    WF_IDLE = 0,          // no path (pixel finished, or slot outside the image)
    WF_PATH_FIRST = 1,    // path ray at bounce 0 awaits the nearest-hit search
    WF_LIGHT_SEARCH = 2,  // light-visibility ray awaits the search
    WF_LIGHT_AMBIENT = 3, // light picked, no search needed: arrival known (AMBIENT)
    WF_LIGHT_NONE = 4,    // light picked, no search needed: nothing arrives (DIRECTIONAL facing away)
    WF_LIGHT_UNDEF = 5,   // light picked, no search needed: the shader's undefined light types
    WF_PATH_NEXT = 6      // path ray at bounce > 0 awaits the search
};
// Device memory of the pooled pipeline (its hot path state lives in LDS): per slot 4 float4 of cold
// state (pixel sum | normal, light distance | path direction, light index | material bits), per lane an
// overflow area for traversal stack entries beyond those kept in LDS.
struct PathPool {
    float4* cold;
    uint32_t* stack_overflow;
    uint32_t n_slots;
    uint32_t overflow_words;  // per lane
};

hipError_t launch_render(const DeviceScene& sc, const FrameArgs& fa, int accel, bool stats, int cu_count,
                         hipStream_t stream);
// The pooled pipeline: one persistent launch, path slots regrouped through LDS queues.
int pool_slots_per_block();
int pool_overflow_per_lane();  // the least a lane's overflow area holds; a four-wide tree may need more (pool_overflow_for)
int pool_overflow_for(int stack_need);  // words per lane for a tree whose traversal holds up to stack_need entries
int pool_threads_per_block();
// accel: 0 the flat loops, 1 the binary tree, 2 the four-wide tree in 128-byte nodes, 3 in 80-byte nodes (sc.nodes must
// then be the nodes of that form)
// per_cu_cap > 0 limits the workgroups per CU below what fits (tuning aid); accel / stats / batch / shader_waves name
// the kernel instance launch_pool will pick (batch: a launch of several frames), whose occupancy is asked for
hipError_t pool_grid(const FrameArgs& fa, int accel, bool stats, bool batch, int shader_waves, int cu_count, int per_cu_cap, unsigned* grid);
hipError_t launch_pool(const DeviceScene& sc, const FrameArgs& fa, const PathPool& pp, int accel, bool stats, unsigned grid,
                       int refill_min, int burst, int job_min, int slots_cap, int shader_waves, hipStream_t stream);
// How many XCDs a grid of one workgroup per CU is spread over (the hardware's XCC_ID of every workgroup, read
// back): their ids must be 0 .. n - 1, each with at least one workgroup; *n_xcds = 0 if that does not hold or
// n > kXcds.  Synchronous; `scratch`: cu_count words of device memory.
hipError_t probe_xcds(int cu_count, uint32_t* scratch, int* n_xcds, hipStream_t stream);
// After a launch of several frames: sets *abort_flag (FrameArgs::abort_flag) unless every XCD handed out all its slots.
hipError_t launch_batch_check(const FrameArgs& fa, hipStream_t stream);
hipError_t launch_math_probe(int fn, const float* x, const float* y, float* out, int n, hipStream_t stream);
// Turns the per-tile costs of the frame just rendered into the next frame's tile order (most
// expensive class first, image order within a class) and clears the costs.  At most
// tile_order_max_tiles() tiles.  `aux`: tile_order_aux_bytes() bytes of work area whose first word is zero
// before the first call (every call leaves it zero again).  `stamp` (may be null): 8 words that receive
// `key` once the order is written, so that a later launch_tile_order_check can tell what the order was made for.
uint32_t tile_order_max_tiles();
size_t tile_order_aux_bytes();
hipError_t launch_tile_order(uint32_t* tile_cost, uint32_t* tile_order, uint32_t n_tiles, int classes, uint32_t* aux, uint32_t* stamp,
                             const uint32_t key[8], hipStream_t stream);
// Replaces tile_order by the identity unless `stamp` holds `key` (device-side check: no host state).
hipError_t launch_tile_order_check(uint32_t* tile_order, uint32_t n_tiles, const uint32_t* stamp, const uint32_t key[8],
                                   hipStream_t stream);
hipError_t launch_resolve(float4* accum, int32_t* counts, uint32_t* bgra, int n_pixels, hipStream_t stream);

}  // namespace rtd
