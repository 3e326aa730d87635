/*
 * rt_abi.h — C ABI of libraytracer_amd.so, the MI355X (gfx950) replacement for the
 * Vulkan compute pipeline of VictorOrrios/Raytracer.
 *
 * The reference has no plugin/FFI layer: its "interface" is the Vulkan binding contract
 * of shaders/raytracer.comp as driven by RaytracingApp in src/main.cpp.  Every entry
 * point below cites the reference code it replaces (paths relative to the reference
 * repository root).  All functions return RT_OK (0) or a negative RtStatus; nothing
 * throws across this boundary.  One context owns one HIP stream on one device; calls on
 * a context are not thread-safe, different contexts may be driven from different threads.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status codes */
typedef enum RtStatus {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1,
    RT_ERR_HIP = -2,            /* a HIP runtime call failed; see rt_last_error()            */
    RT_ERR_NO_SCENE = -3,       /* render requested before rt_upload_scene                    */
    RT_ERR_NO_FRAMEBUFFER = -4, /* render requested before rt_resize / rt_bind_framebuffer    */
    RT_ERR_UNSUPPORTED = -5,    /* not a gfx950 device; a BVH the traversal cannot walk; a retired flag */
    RT_ERR_IO = -6,             /* asset file missing / malformed                             */
    RT_ERR_TOO_LARGE = -7
} RtStatus;

/* ------------------------------------------------------------------ scene element layouts
 * std430 layouts of the shader's SSBO element types (shaders/raytracer.comp:47-106) and
 * of their host mirrors (src/definitions.hpp:18-87).  Sizes/offsets are static_asserted
 * in raytracer_amd/csrc/rt_layout_check.cpp. */

/* light types: shaders/raytracer.comp:16-22, src/definitions.hpp:7-15 */
enum { RT_LIGHT_AMBIENT = 0, RT_LIGHT_SPHERE = 1, RT_LIGHT_POINT = 2, RT_LIGHT_DIRECTIONAL = 3,
       RT_LIGHT_CONE = 4, RT_LIGHT_AREA = 5, RT_LIGHT_TRIANGLE = 6 };

typedef struct RtSphere {   /* 32 B: raytracer.comp:84-88, definitions.hpp:18-22 */
    float   pos[3];
    float   r;
    int32_t mat;
    int32_t _pad[3];
} RtSphere;

typedef struct RtTriangle { /* 64 B: raytracer.comp:90-96, definitions.hpp:24-30 */
    float   v0[3];  float _p0;
    float   v1[3];  float _p1;
    float   v2[3];  float _p2;
    float   normal[3];
    int32_t mat;
} RtTriangle;

typedef struct RtVertex {   /* 16 B: raytracer.comp:104-106, definitions.hpp:40-42 */
    float pos[3];
    float _pad;
} RtVertex;

typedef struct RtMaterial { /* 80 B: raytracer.comp:54-82, definitions.hpp:45-73 */
    float albedo[4];
    float subsurface[4];
    float specular_tint[4];
    float emission_color[4];
    float roughness;
    float metallic;
    float ior;
    float trs_weight;
} RtMaterial;

typedef struct RtLight {    /* 48 B: raytracer.comp:47-52, definitions.hpp:76-81 */
    float   pos_angle_aux[4];
    float   color_str[4];
    int32_t type;
    float   accumulated_str;
    int32_t _pad[2];
} RtLight;

typedef struct RtMeshInfo { /* 12 B: raytracer.comp:98-102, definitions.hpp:83-87 */
    uint32_t index_start;
    uint32_t index_end;
    int32_t  material;
} RtMeshInfo;

/* Camera uniform block, std140, 336 B (raytracer.comp:27-35; host main.cpp:141-150).
 * Matrices are column-major.  The kernels read viewInv, position and tanHalfFOV only. */
typedef struct RtCamera {
    float view[16];
    float viewInv[16];
    float proj[16];
    float projInv[16];
    float viewproj[16];
    float position[3];
    float tanHalfFOV;
} RtCamera;

/* Push-constant block, 44 B (raytracer.comp:127-137; host main.cpp:158-169).  The host
 * struct's 1-byte bool is widened to the 32-bit bool the shader actually reads. */
typedef struct RtPushConstants {
    float    time;
    uint32_t frameCount;
    int32_t  total_lights;
    float    lights_strength_sum;
    float    world_up[3];
    uint32_t reset_frame_accumulation;
    int32_t  total_spheres;
    int32_t  total_triangles;
    int32_t  total_meshes;
} RtPushConstants;

/* Nearest-hit search strategy.  Both give the reference's result (lowest t; exact-t ties
 * resolved as the flat loops of raytracer.comp:469-508 / 408-464 resolve them). */
enum { RT_ACCEL_FLAT = 0,   /* the reference's three flat loops, every ray x every primitive */
       RT_ACCEL_BVH  = 1 }; /* binary BVH over all primitives, same winner                    */

/* Per-frame parameters that the reference hard-codes in the shader
 * (rays_per_pixel = 5, max_bounces = 20: raytracer.comp:117-118) plus the row-strip
 * partition used to split one frame over several GPUs.  rt_default_params() fills the
 * reference's values. */
typedef struct RtParams {
    int32_t spp;          /* rays_per_pixel                                                   */
    int32_t max_bounces;  /* path loop runs bounce = 0..max_bounces inclusive                 */
    int32_t accel;        /* RT_ACCEL_*                                                        */
    int32_t strip_rows;   /* height of one row strip; <= 0: this context renders every row    */
    int32_t strip_count;  /* number of contexts sharing the frame (>= 1)                       */
    int32_t strip_index;  /* this context renders rows y with (y/strip_rows)%strip_count==idx  */
    uint32_t flags;       /* RT_FLAG_*                                                         */
    int32_t skybox;       /* RT_SKY_*: which of the shader's skybox functions a missed ray sees */
} RtParams;

/* The shader's skybox variants (raytracer.comp:520-577); the reference compiles in the grey one
 * (575-577), the others are a one-line edit there and a parameter here. */
enum { RT_SKY_GREY = 0, RT_SKY_DAY = 1, RT_SKY_NIGHT = 2, RT_SKY_WHITE = 3, RT_SKY_BLACK = 4 };

enum { RT_FLAG_PRIMARY_IDS = 1u,   /* record sample-0 primary-hit primitive id per pixel       */
       RT_FLAG_PATH_HASH   = 2u,   /* record a hash of every nearest-hit id along the paths    */
       RT_FLAG_COUNT_RAYS  = 4u,   /* accepted, no effect: rays are always counted             */
       RT_FLAG_TRAVERSAL_STATS = 8u, /* count BVH nodes visited / primitives tested            */
       RT_FLAG_RESERVED_16 = 16u,  /* selected a pipeline of one launch per stage that was measured,
                                      lost everywhere and was removed; refused with RT_ERR_UNSUPPORTED */
       RT_FLAG_POOLED      = 32u,  /* pooled pipeline: one persistent launch whose workgroups keep
                                      two paths per lane, their hot state in LDS, and regroup them
                                      by kind of work through queues in LDS; same results         */
       RT_FLAG_SINGLE_KERNEL = 64u, /* the single persistent kernel, one path per lane             */
       RT_FLAG_FIXED_TILE_ORDER = 128u /* hand the image's 8x8 tiles out in image order instead of
                                      dearest-first as measured on the previous frame of the same
                                      size; same results, only the schedule differs              */ };
/* With neither RT_FLAG_POOLED nor RT_FLAG_SINGLE_KERNEL the library chooses: with the BVH and at
 * least 16384 pixels it times the pooled kernel (third frame of a configuration) against the single
 * kernel (fifth frame) and runs the faster from the sixth frame on (pooled while a timing is still
 * outstanding: the events are polled, never waited for); otherwise the single kernel.
 * Both pipelines give the same bits. */
enum { RT_PIPELINE_SINGLE = 0, RT_PIPELINE_POOLED = 2 };

typedef struct RtStats {
    uint64_t rays;          /* nearest-hit searches in the last frame (always counted)         */
    uint64_t nodes_visited; /* BVH nodes whose children were box-tested (needs TRAVERSAL_STATS; see bvh_width) */
    uint64_t prims_tested;  /* primitive intersection tests (needs TRAVERSAL_STATS)            */
    uint64_t total_rays;    /* nearest-hit searches of ALL frames since rt_create              */
    /* the next five need TRAVERSAL_STATS (instrumented kernel, never the timed one):           */
    uint64_t cycles_raygen; /* wave cycles spent in pixel refill + ray generation              */
    uint64_t cycles_search; /* wave cycles spent in the nearest-hit search                     */
    uint64_t cycles_shade;  /* wave cycles spent in the continuation (lights, BSDF, store)     */
    uint64_t lane_iterations; /* traversal loop iterations summed over lanes                   */
    uint64_t wave_iterations; /* traversal loop iterations summed over waves                   */
    uint64_t cycles_shade_parts[5]; /* cycles_shade split: hit record + light pick, direct BSDF,
                                       direction sampling, indirect BSDF, sample/pixel end      */
    float    kernel_ms;     /* HIP-event time of the last frame's render kernel                */
    int32_t  bvh_nodes;     /* nodes in the device BVH                                         */
    int32_t  bvh_depth;     /* maximum depth of the device BVH                                 */
    int32_t  total_prims;   /* spheres + triangles + mesh triangles                            */
    int32_t  pipeline;      /* RT_PIPELINE_* that rendered the last frame                      */
    uint64_t pool_job_cycles[4]; /* pooled pipeline with TRAVERSAL_STATS: wave cycles spent in
                                    shading jobs of each kind (first-bounce result, later-bounce
                                    result, light-query result, sample end); cycles_search then
                                    holds the search section, cycles_shade the sum of the four and
                                    cycles_raygen everything else (queue polls, waiting, start)  */
    uint64_t busy_lane_iterations; /* pooled pipeline with TRAVERSAL_STATS: lanes that held a ray, summed over
                                      the search steps of all waves (lane_iterations counts those that took the
                                      step: the rest waited for the other kind of step)                      */
    uint64_t cycles_steps;  /* pooled pipeline with TRAVERSAL_STATS: the part of cycles_search spent in the bursts of
                               search steps (the rest: handing results in, taking new rays)                      */
    int32_t  bvh_width;     /* children per node of the tree the last frame's kernel walked: 2, or 4 (pooled pipeline,
                               RT_OPT_BVH_WIDTH; nodes_visited then counts four-wide nodes); 0 for the flat loops   */
    int32_t  frames;        /* frames the last launch rendered (rt_render_frames fuses up to 32; else 1); rays, the
                               traversal counts and kernel_ms are that launch's                                    */
    int32_t  search_queue_peak; /* pooled pipeline with TRAVERSAL_STATS: the most rays any workgroup's waves saw waiting
                                   for a lane at once ...                                                            */
    int32_t  search_queue_size; /* ... and what that queue holds (a ray pushed beyond it would be lost, and the frame
                                   would fail with RT_ERR_HIP instead of finishing)                                  */
    int32_t  bvh_node_bytes; /* bytes per node of the tree the last frame's kernel walked: 64 (binary), 128 or 80 (four-wide,
                                see RT_OPT_BVH_NODE_FORMAT); 0 for the flat loops                                     */
    int32_t  reserved0;
    uint64_t stack_overflow_steps; /* pooled pipeline with TRAVERSAL_STATS: node visits that pushed onto a traversal stack
                                      whose part in LDS was already full, i.e. whose entries went to the lane's overflow
                                      area in device memory                                                          */
} RtStats;

typedef struct RtContext RtContext;
typedef struct RtScene RtScene;   /* host scene, see "host scene surface" below */

/* ------------------------------------------------------------------ device path */

/* Replaces device/queue selection + compute pipeline creation
 * (main.cpp:799-946 pickPhysicalDevice/createLogicalDevice, 1127-1186 createComputePipeline).
 * Fails with RT_ERR_HIP when no gfx950 device / HIP runtime is usable. */
int rt_create(int device_ordinal, RtContext** out_ctx);

/* Replaces cleanup() of the scene/accumulation/output buffers (main.cpp:311-382). */
int rt_destroy(RtContext* ctx);

/* Replaces createShaderStorageBuffers / createSSBOVector (main.cpp:1555-1593): the seven
 * scene vectors in binding order 1..7 (raytracer.comp:145-171).  Host arrays stay owned
 * by the caller and are copied; counts may be 0 with a non-null pointer (the reference
 * keeps one placeholder element per vector, scene.cpp:26-32).  Builds the device-side
 * SoA triangle arrays and the BVH. */
int rt_upload_scene(RtContext* ctx,
                    const RtSphere* spheres, int32_t n_spheres,
                    const RtMaterial* materials, int32_t n_materials,
                    const RtLight* lights, int32_t n_lights,
                    const RtTriangle* triangles, int32_t n_triangles,
                    const RtVertex* vertices, int32_t n_vertices,
                    const uint32_t* indices, int32_t n_indices,
                    const RtMeshInfo* meshes, int32_t n_meshes);

/* Replaces createImageBuffer + createFrameAccumulationBuffers and their re-creation on
 * resize (main.cpp:1501-1552, 1596-1606, 1396-1420): allocates ZEROED accumulation
 * (float4 per pixel), sample-count (int32 per pixel) and BGRA8 output buffers.  An image holds at most 2^27 pixels
 * (RT_ERR_INVALID_ARGUMENT beyond; the same bound applies to rt_bind_framebuffer and rt_render_frame_noalloc). */
int rt_resize(RtContext* ctx, int32_t width, int32_t height);

/* Same role as rt_resize but the caller provides (and keeps owning) the device memory,
 * e.g. torch tensors that a collective library reads directly.  Nothing is zeroed. */
int rt_bind_framebuffer(RtContext* ctx, int32_t width, int32_t height,
                        void* d_accum_float4, void* d_counts_int32, void* d_bgra8);

/* One reference frame: updateUniformBuffer + push constants + vkCmdDispatch
 * (main.cpp:1466-1497, 1424-1438, 1234-1241) = one execution of raytracer.comp::main per
 * pixel (raytracer.comp:956-996), including reset_frame_accumulation and the accumulate /
 * BGRA8 store.  Asynchronous on the context's stream: it enqueues and returns, except that the
 * first frame of a new size / partition / pipeline may allocate scratch memory (which waits for the
 * stream).  params == NULL: reference defaults.  A frame the pooled kernel had to abandon (a bound
 * on waiting inside a workgroup was exceeded; never seen in a correct run) is reported by the next
 * rt_sync / rt_read_* / rt_get_stats as RT_ERR_HIP. */
int rt_render_frame(RtContext* ctx, const RtCamera* camera, const RtPushConstants* pc,
                    const RtParams* params);

/* n_frames consecutive rt_render_frame calls with one camera and pcs[0 .. n_frames - 1], with the same result bit for
 * bit - the reference's frame loop with a still camera (main.cpp:2160-2171: frameCount differs from frame to frame,
 * the accumulation is reset at most by the first).  Where the frames differ in frameCount only and the pooled kernel
 * applies (BVH search, no debug outputs, an image of 64 tiles to 4 M pixels), up to 32 frames at a time share ONE launch:
 * a pixel's frames follow each other on one XCD, so the drain of one frame's last paths is hidden behind the next
 * frame's first.  Otherwise the frames are rendered one by one.  RtStats then describe the last launch (rays: all its
 * frames). */
int rt_render_frames(RtContext* ctx, const RtCamera* camera, const RtPushConstants* pcs, int32_t n_frames,
                     const RtParams* params);

/* rt_render_frame for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same frame, same bits,
 * but the framebuffer comes as arguments (device memory, width*height float4 / int32 / BGRA8, aligned
 * 16/4/4, contents kept as rt_render_frame keeps them) and everything else the frame needs (tile order,
 * the pooled kernel's slot records and stack overflow) lives in a fixed 47.2 MiB device array inside the
 * library instead of memory allocated on demand: the call makes no hipMalloc / hipFree and waits for
 * nothing.  Limits: at most 131072 tiles of 8x8 pixels (8.39 M pixels) -> RT_ERR_TOO_LARGE;
 * RT_FLAG_PRIMARY_IDS / RT_FLAG_PATH_HASH -> RT_ERR_UNSUPPORTED (their buffers are allocated on
 * demand); no timed trial between the kernels (BVH frames of >= 16384 pixels run pooled, others
 * single, the flags force either).  All calls on one GPU share that GPU's array: calls that overlap
 * in time on one GPU (two contexts, two threads) race and must be serialised by the caller. */
int rt_render_frame_noalloc(RtContext* ctx, const RtCamera* camera, const RtPushConstants* pc,
                            const RtParams* params, void* d_accum_float4, void* d_counts_int32,
                            void* d_bgra8, int32_t width, int32_t height);

/* Recompute the BGRA8 image from the accumulation buffers (raytracer.comp:992-995) after
 * they were combined across GPUs. */
int rt_resolve(RtContext* ctx);

/* Replaces the output-image copy (main.cpp:1275 vkCmdCopyImage): w*h*4 bytes, B,G,R,A. */
int rt_read_output_bgra8(RtContext* ctx, void* dst);
/* Read back accumulated_colors (w*h float4) and sample_counts (w*h int32); either may be NULL. */
int rt_read_accum(RtContext* ctx, float* accum_float4, int32_t* counts);
/* Debug outputs with no counterpart in the reference (its Hit struct has no id field):
 * primitive id of sample 0's primary hit (-1 = miss; spheres [0,S), triangles [S,S+T),
 * mesh triangles S+T+index/3) and the per-pixel path hash.  Either may be NULL. */
int rt_read_debug(RtContext* ctx, int32_t* primary_ids, uint32_t* path_hash);

/* Self-test hook with no counterpart in the reference: evaluates one of the device's
 * arithmetic primitives element-wise (fn: 0 sin, 1 cos, 2 tan, 3 acos, 4 pow(x,y), 5 log2,
 * 6 exp2, 7 asin, 8 x/y, 9 sqrt, 10 random() from state bits x, 11 x*y+x, 12 1/sqrt(x*x+y*y)).
 * x, y, out are HOST arrays of n floats. */
int rt_debug_math(RtContext* ctx, int fn, const float* x, const float* y, float* out, int32_t n);

int rt_get_stats(RtContext* ctx, RtStats* out);

/* Tuning knobs with no counterpart in the reference.  Results never depend on them, only the
 * schedule (and for RT_OPT_BVH_MAX_LEAF the shape of the tree built by the NEXT rt_upload_scene). */
enum { RT_OPT_EXIT_QUOTA_PCT = 1, /* single kernel: leave the search loop once this % of busy lanes have a result (1..100, 75) */
       RT_OPT_REFILL_MIN = 2,     /* pooled kernel: idle lanes of a searching wave before it fetches new rays (1..64, 16)      */
       RT_OPT_POOL_BURST = 3,     /* pooled kernel: search steps between two looks at the queues (0 = by scene size, ..1024)   */
       RT_OPT_POOL_JOB_MIN = 4,   /* pooled kernel: queue entries that make a shading job (1..64; 0 = by scene size)         */
       RT_OPT_TILE_CLASSES = 5,   /* cost classes of the frame-to-frame tile order (1..64, 16)                               */
       RT_OPT_POOL_PER_CU = 6,    /* pooled kernel: cap on workgroups per CU (0 = what fits, ..8)                            */
       RT_OPT_BVH_MAX_LEAF = 7,   /* most primitives per BVH leaf (1..8, 4)                                                  */
       RT_OPT_POOL_SLOTS = 8,     /* pooled kernel: paths in flight per workgroup (0 = chosen by scene size, ..1344 = all)   */
       RT_OPT_BVH_TREELET = 9,    /* BVH node layout: levels per treelet (0 / 1 = depth-first order, ..16); next upload      */
       RT_OPT_BVH_WIDTH = 10,     /* pooled kernel: children per node of the tree it walks (2, 4; 0 = 4 when its stack fits) */
       RT_OPT_POOL_SHADER_WAVES = 11, /* pooled kernel: waves of a workgroup's 16 that only shade (1..8; 9 = none; 0 = by scene) */
       RT_OPT_BVH_NODE_FORMAT = 12 }; /* pooled kernel, four-wide tree: 32 = 128-byte nodes (binary32 boxes), 16 = 80-byte nodes
                                       (binary16 boxes relative to a point of the node, rounded outwards: same results, fewer
                                       bytes fetched per visit); 0 = 16 for scenes beyond 4 MB (one XCD's L2), else 32          */
int rt_set_option(RtContext* ctx, int option, int value);
/* Device pointers of the live framebuffer (for zero-copy use by a collective library). */
int rt_get_device_buffers(RtContext* ctx, void** d_accum, void** d_counts, void** d_bgra8);
/* The context's hipStream_t. */
void* rt_get_stream(RtContext* ctx);
/* Replaces vkWaitForFences / vkQueueWaitIdle (main.cpp:1310-1312). */
int rt_sync(RtContext* ctx);

const char* rt_last_error(RtContext* ctx); /* ctx may be NULL: error of the calling thread's last failed rt_create */
void rt_default_params(RtParams* out);     /* spp 5, max_bounces 20, BVH, all rows, no flags     */
/* The header a host was compiled against and the library it loaded must agree: RT_ABI_VERSION changes whenever a
 * struct of this header changes size or layout or an entry point changes meaning; rt_abi_version() is the value the
 * library was built with, rt_stats_size() its sizeof(RtStats) (the struct that grows most often: rt_get_stats writes
 * that many bytes).  No counterpart in the reference. */
#define RT_ABI_VERSION 2
int  rt_abi_version(void);
size_t rt_stats_size(void);

/* ------------------------------------------------------------------ several GPUs of one node
 * The reference renders on one GPU; these entry points have no counterpart in it beyond the per-frame
 * call they multiply, drawFrame (main.cpp:1308-1392).  A pixel depends only on (x, y, time, frameCount)
 * and the read-only scene (raytracer.comp:959-961), and frames couple only through the additive
 * accumulation (990-992), so the work divides two ways with the scene replicated on every GPU:
 *   row strips - one frame's rows dealt to the GPUs in strips of strip_rows rows (RtParams.strip_*),
 *                completed everywhere by ONE all-gather of each GPU's packed rows; same bits as one GPU;
 *   samples    - whole frames (distinct frameCount, hence distinct random streams) dealt to the GPUs,
 *                each accumulating privately; one reduction (sum) of colours and counts per batch.   */

/* Bytes of one context's packed rows under a strip partition (the same for every strip_index: room for
 * the largest share, rounded up to 16): owned rows in ascending y, all float4 colours first, then all
 * int32 counts. */
size_t rt_strip_staging_bytes(int32_t width, int32_t height, int32_t strip_rows, int32_t strip_count);
/* Copy the context's own rows (accumulated colours + counts) into d_staging (device memory, 16-byte
 * aligned, rt_strip_staging_bytes).  Asynchronous on the context's stream. */
int rt_strip_pack(RtContext* ctx, int32_t strip_rows, int32_t strip_count, int32_t strip_index, void* d_staging);
/* d_gathered = the packed rows of strip_count contexts back to back (what an all-gather of the staging
 * buffers yields): write every context's rows except skip_index's (-1: none) into this context's
 * framebuffer and rebuild their BGRA8 bytes.  Asynchronous on the context's stream. */
int rt_strip_unpack(RtContext* ctx, int32_t strip_rows, int32_t strip_count, int32_t skip_index, const void* d_gathered);

/* One host thread driving n devices: a context per device plus an RCCL communicator over them
 * (single process, ncclCommInitAll; librccl.so is loaded here, on first use). */
typedef struct RtGroup RtGroup;
enum { RT_GROUP_STRIPS = 0, RT_GROUP_SAMPLES = 1 };
int rt_group_create(int n_devices, const int* device_ordinals, RtGroup** out); /* failure text: rt_last_error(NULL) */
int rt_group_destroy(RtGroup* grp);
int rt_group_size(const RtGroup* grp);
int rt_group_context(RtGroup* grp, int index, RtContext** out);  /* borrowed: e.g. for rt_read_*, rt_get_stats */
int rt_group_upload_scene_from(RtGroup* grp, const RtScene* s);  /* replicated on every device */
int rt_group_resize(RtGroup* grp, int32_t width, int32_t height);
/* RT_GROUP_STRIPS: one reference frame, rendered by all devices in strips of strip_rows rows and
 * completed on every device (all-gather over xGMI); afterwards each context holds the whole frame, bit
 * for bit what rt_render_frame gives on one device.  params->strip_* are overwritten.
 * RT_GROUP_SAMPLES: n frames at once - device i renders frameCount + i into its own accumulator
 * (strip_rows ignored); combine them with rt_group_reduce_samples when the batch is complete.
 * Asynchronous on the contexts' streams. */
int rt_group_render_frame(RtGroup* grp, const RtCamera* camera, const RtPushConstants* pc, const RtParams* params,
                          int mode, int32_t strip_rows);
/* `frames` frames of a still camera per call.  RT_GROUP_STRIPS: the frames numbered frameCount, frameCount + 1, ...; every
 * device renders its strips of all of them (one rt_render_frames call: one launch where that applies) and the
 * accumulated frame is completed everywhere by ONE all-gather - the result of `frames` rt_group_render_frame calls, bit for
 * bit, for one exchange instead of `frames`, and with a device's share of frame k + 1 filling the drain of its share of
 * frame k.  RT_GROUP_SAMPLES: device i renders the frames numbered frameCount + i, + i + n, ... into its own accumulator,
 * as `frames` calls with frameCount advanced by n each time.  pc->reset_frame_accumulation applies to the first frame
 * (of every device). */
int rt_group_render_frames(RtGroup* grp, const RtCamera* camera, const RtPushConstants* pc, const RtParams* params, int mode,
                           int32_t strip_rows, int32_t frames);
/* Sum of all devices' accumulated colours and counts on device 0 (one ncclReduce each) and its BGRA8 image
 * in context 0's output; the devices' own accumulators are left as they are, so a progressive
 * rendering can go on and be reduced again.  Float sums run in the reduction's order, not in frame
 * order: equal to sequential accumulation within rounding, not bit for bit (one device: bit for bit). */
int rt_group_reduce_samples(RtGroup* grp);
int rt_group_read_total(RtGroup* grp, float* accum_float4, int32_t* counts); /* the reduced buffers */
int rt_group_sync(RtGroup* grp);
const char* rt_group_last_error(RtGroup* grp);

/* ------------------------------------------------------------------ host scene surface
 * C access to the C++ Scene builder that mirrors src/scene.hpp:8-37 / src/scene.cpp. */

enum { RT_SCENE_EMPTY = 0,
       RT_SCENE_CORNELL_STAR = 1,   /* Scene::createCornellBox verbatim (scene.cpp:350-695)    */
       RT_SCENE_PRESET1 = 2,        /* Scene::createPreset1 (scene.cpp:37-348)                  */
       RT_SCENE_CORNELL_TEAPOT = 3, /* Cornell box with teapot.glb in place of star.glb         */
       RT_SCENE_TEAPOT_GRID = 4,    /* 13x13 teapots baked into one mesh (~1.07 M triangles)    */
       RT_SCENE_PRESET1_TEAPOT = 5 }; /* createPreset1 with the teapot it defines but leaves out
                                       (scene.cpp:333-343: blue matte, pitch 90, at (0,-1,10))   */

int rt_scene_create(int preset, const char* assets_dir, RtScene** out);
int rt_scene_destroy(RtScene* s);
/* Builder calls, same meaning as Scene::addMaterial/addSphere/addTriangle/addQuad/addLight/
 * addModel (scene.cpp:716-846).  rt_scene_add_material returns the material index (>= 0). */
int rt_scene_add_material(RtScene* s, const RtMaterial* m);
int rt_scene_add_sphere(RtScene* s, const float pos[3], float r, int32_t mat);
int rt_scene_add_triangle(RtScene* s, const float v0[3], const float v1[3], const float v2[3], int32_t mat);
int rt_scene_add_quad(RtScene* s, const float v0[3], const float v1[3], const float v2[3], const float v3[3], int32_t mat);
int rt_scene_add_light(RtScene* s, const float pos_angle_aux[4], const float color_str[4], int32_t type);
int rt_scene_add_model(RtScene* s, const char* file_name, const float pos[3],
                       float pitch, float yaw, float roll, float scale, int32_t material);

/* rt_scene_add_model with optional departures from the reference's loader and index handling:
 * RT_MODEL_FULL_GLTF reads the whole default scene of the file (node transforms, every mesh and
 * triangle primitive, byteStride, 8/16/32-bit or absent indices) instead of mesh 0 / primitive 0
 * with tightly packed positions (loader.cpp:37-43, 76-91); RT_MODEL_REBASE_INDICES adds the
 * model's first vertex position in the scene's vertex buffer to its indices, which addModel
 * does not (scene.cpp:844-845), so that a scene can hold more than one model. */
enum { RT_MODEL_FULL_GLTF = 1u, RT_MODEL_REBASE_INDICES = 2u };
int rt_scene_add_model_ex(RtScene* s, const char* file_name, const float pos[3],
                          float pitch, float yaw, float roll, float scale, int32_t material, uint32_t flags);

typedef struct RtSceneView {  /* borrowed pointers into the RtScene, valid until it changes */
    const RtSphere* spheres;     int32_t n_spheres;    /* vector sizes (>= 1: placeholder rule) */
    const RtMaterial* materials; int32_t n_materials;
    const RtLight* lights;       int32_t n_lights;
    const RtTriangle* triangles; int32_t n_triangles;
    const RtVertex* vertices;    int32_t n_vertices;
    const uint32_t* indices;     int32_t n_indices;
    const RtMeshInfo* meshes;    int32_t n_meshes;
    int32_t total_spheres, total_triangles, total_meshes, total_lights; /* scene.hpp:18-21 */
    float   lights_strength_sum;                                          /* scene.hpp:17    */
} RtSceneView;
int rt_scene_view(const RtScene* s, RtSceneView* out);
/* rt_upload_scene with the scene's vectors and total_* counts. */
int rt_upload_scene_from(RtContext* ctx, const RtScene* s);

/* Camera block as updateUniformBuffer fills it (main.cpp:1466-1497) from a pose. */
int rt_camera_look(RtCamera* out, const float pos[3], const float front[3], const float up[3],
                   float fov_degrees, int32_t width, int32_t height);
/* The reference's initial pose after its first updateCamera(): position (2.75,2.75,-6),
 * front +z, up +y, fov 50 (main.cpp:75-86, 554-565). */
int rt_camera_default(RtCamera* out, int32_t width, int32_t height);
/* Push constants as updatePushConstantsPre fills them (main.cpp:1424-1438). */
int rt_push_constants(RtPushConstants* out, const RtScene* s, float time, uint32_t frame_count,
                      int reset_accumulation);

/* ------------------------------------------------------------------ camera controller
 * The application's interactive camera and its per-frame seeding policy as a headless
 * state machine: RaytracingApp's camera/accumulation members (main.cpp:250-273), mainLoop's
 * clock step (283-285), processInput (402-496), the mouse callbacks (499-550), updateCamera
 * (554-565), the frame's camera block and push constants (1331-1340) and showFPS's frame
 * counter (2156-2174).  Window-system events become explicit calls.  It starts from the pose
 * after the application's first updateCamera(): position (2.75,2.75,-6), yaw 180, pitch 0. */
typedef struct RtController RtController;

enum { RT_KEY_W = 1u, RT_KEY_S = 2u, RT_KEY_A = 4u, RT_KEY_D = 8u, RT_KEY_SPACE = 16u,
       RT_KEY_LEFT_CONTROL = 32u, RT_KEY_LEFT_SHIFT = 64u, RT_KEY_Q = 128u, RT_KEY_E = 256u,
       RT_KEY_F = 512u, RT_KEY_R = 1024u, RT_KEY_X = 2048u,
       RT_KEY_RELEASE_MOUSE = 4096u, /* ESCAPE, LEFT_ALT or TAB (main.cpp:490-495)          */
       RT_KEY_K = 8192u };           /* asks the window to close (main.cpp:445-446)          */

typedef struct RtControllerState {
    float pos[3], front[3], up[3];
    float yaw, pitch, roll, fov;   /* degrees                                                */
    float last_frame;              /* the clock value that becomes pushConstants.time        */
    int32_t frame_count;           /* the counter that becomes pushConstants.frameCount      */
    int32_t mouse_captured, reset_accumulation, accumulation_on, should_close;
} RtControllerState;

int rt_controller_create(RtController** out);
int rt_controller_destroy(RtController* c);
int rt_controller_get(const RtController* c, RtControllerState* out);
/* planeMode (main.cpp:88-90): constant forward motion instead of the movement keys. */
int rt_controller_set_plane_mode(RtController* c, int on);
/* One main-loop iteration up to drawFrame: the clock step (deltaTime = now - lastFrame) and
 * processInput with the keys that are down (a mask of RT_KEY_*). */
int rt_controller_begin_frame(RtController* c, double now_seconds, uint32_t keys, float* delta_time_out);
int rt_controller_mouse_button(RtController* c, int left_pressed);      /* focus_callback  */
int rt_controller_mouse_move(RtController* c, double xpos, double ypos); /* mouse_callback  */
int rt_controller_scroll(RtController* c, double yoffset);               /* scroll callback */
/* The frame's inputs: push constants (time = the clock value of begin_frame, frameCount, reset
 * rule of main.cpp:1430-1434) and camera block; clears the pending reset like
 * updatePushConstantsPost. */
int rt_controller_frame_inputs(RtController* c, const RtScene* s, int32_t width, int32_t height,
                               RtCamera* cam, RtPushConstants* pc);
/* showFPS: frameCount++, and back to 0 once a second has passed since the window began. */
int rt_controller_end_frame(RtController* c, double now_seconds);

/* Builds the host BVH of a scene (no GPU needed), checks its structure (every primitive in
 * exactly one leaf, boxes nested, depth within the traversal limit; RT_ERR_UNSUPPORTED if
 * not) and reports node count, depth, primitive count and SAH cost. */
int rt_scene_bvh_info(const RtScene* s, int32_t* n_nodes, int32_t* depth, int32_t* n_prims, float* sah_cost);
/* Builds the four-wide form of that tree (what the pooled kernel walks on scenes beyond the caches, see
 * RT_OPT_BVH_WIDTH), checks that it holds exactly the binary tree's leaves (RT_ERR_UNSUPPORTED if not) and
 * reports its node count, depth and the most stack entries a traversal of it can hold.  No GPU needed.
 * No counterpart in the reference, which has no acceleration structure (raytracer.comp:469-508). */
int rt_scene_wide_bvh_info(const RtScene* s, int32_t* n_nodes, int32_t* depth, int32_t* stack_need);
/* Fingerprint (FNV-1a) of the node array and leaf order rt_upload_scene would build for this scene:
 * the build is deterministic, whatever the number of host threads it uses. */
int rt_scene_bvh_hash(const RtScene* s, uint64_t* hash);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
