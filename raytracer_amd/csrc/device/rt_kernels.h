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
    const float4* nodes;       // 4 float4 per BVH node, node 0 = root
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
    int32_t tiles_x;                                // ceil(width / 8)
    uint32_t total_slots;                           // tiles_x * ceil(owned_rows / 8) * 64
    uint32_t flags;
    int32_t exit_quota_pct;  // the search loop returns once this share of the wave's busy lanes has finished
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
    // the continuation
    uint32_t* counters;
};

enum { kBlock = 256, kStackDepth = 32 };

hipError_t launch_render(const DeviceScene& sc, const FrameArgs& fa, int accel, bool stats, int cu_count,
                         hipStream_t stream);
hipError_t launch_math_probe(int fn, const float* x, const float* y, float* out, int n, hipStream_t stream);
hipError_t launch_resolve(float4* accum, int32_t* counts, uint32_t* bgra, int n_pixels, hipStream_t stream);

}  // namespace rtd
