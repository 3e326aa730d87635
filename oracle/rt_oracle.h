/* rt_oracle.h — C interface of the CPU oracle (oracle/rt_oracle.cpp).  TEST INFRASTRUCTURE:
 * only tests/, the smoke test and the CPU-baseline leg of the benchmark may use it.
 * The element layouts restate the shader's buffer blocks (shaders/raytracer.comp:47-106,
 * 27-35, 127-137) independently of the product's header. */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float pos[3]; float r; int32_t mat; int32_t pad[3]; } RtoSphere;              /* 32 B */
typedef struct { float v0[3], p0, v1[3], p1, v2[3], p2, normal[3]; int32_t mat; } RtoTriangle; /* 64 B */
typedef struct { float pos[3]; float pad; } RtoVertex;                                          /* 16 B */
typedef struct { float albedo[4], subsurface[4], specular_tint[4], emission_color[4];
                 float roughness, metallic, ior, trs_weight; } RtoMaterial;                     /* 80 B */
typedef struct { float pos_angle_aux[4], color_str[4]; int32_t type; float accumulated_str; int32_t pad[2]; } RtoLight; /* 48 B */
typedef struct { uint32_t index_start, index_end; int32_t material; } RtoMeshInfo;              /* 12 B */
typedef struct { float view[16], viewInv[16], proj[16], projInv[16], viewproj[16]; float position[3]; float tanHalfFOV; } RtoCamera; /* 336 B */
typedef struct { float time; uint32_t frameCount; int32_t total_lights; float lights_strength_sum; float world_up[3];
                 uint32_t reset_frame_accumulation; int32_t total_spheres, total_triangles, total_meshes; } RtoPushConstants; /* 44 B */

typedef struct {
    const RtoSphere* spheres;
    const RtoMaterial* materials;
    const RtoLight* lights;
    const RtoTriangle* triangles;
    const RtoVertex* vertices;
    const uint32_t* indices;
    const RtoMeshInfo* meshes;
} RtoScene;

typedef struct {
    int32_t width, height;
    int32_t spp, max_bounces;      /* rays_per_pixel / max_bounces of raytracer.comp:117-118 */
    int32_t math_mode;             /* 0: explicit binary32 polynomials, 1: C library */
    int32_t threads;               /* rows are dealt round-robin to this many threads */
    int32_t row_begin, row_end;    /* render rows [row_begin, row_end); row_end <= 0: height */
    int32_t strip_rows, strip_count, strip_index; /* strip_rows > 0: only rows with (y/strip_rows)%strip_count == strip_index */
    int32_t skybox;                /* 0 grey (the shader's choice, raytracer.comp:575-577), 1 day, 2 night, 3 white, 4 black */
} RtoParams;

typedef struct {
    float* accum;          /* w*h*4, accumulated_colors (in/out) */
    int32_t* counts;       /* w*h, sample_counts (in/out)        */
    uint8_t* bgra;         /* w*h*4 or NULL                      */
    int32_t* primary_ids;  /* w*h or NULL                        */
    uint32_t* path_hash;   /* w*h or NULL                        */
    uint64_t* rays;        /* 1 or NULL: hit_scene calls         */
} RtoOutputs;

int rto_render(const RtoScene* scene, const RtoCamera* camera, const RtoPushConstants* pc, const RtoParams* params,
               const RtoOutputs* out);
float rto_math(int fn, int math_mode, float x, float y);
uint32_t rto_hash(uint32_t x);
uint32_t rto_xorshift(uint32_t state);
float rto_random_from(uint32_t state);
uint32_t rto_pixel_seed(float time, uint32_t frame_count, int x, int y);
int rto_hit_scene(const RtoScene* scene, const RtoPushConstants* pc, const float orig[3], const float dir[3], float* t_out);

#ifdef __cplusplus
}
#endif
#endif
