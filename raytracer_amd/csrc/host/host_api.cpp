// host_api.cpp — C entry points for the host scene surface (rt_scene_*, rt_camera_*,
// rt_push_constants) declared in include/rt_abi.h.  No HIP calls in this file.
#include <cmath>
#include <cstring>
#include <new>
#include <stdexcept>

#include "../../../include/rt_abi.h"
#include "bvh_builder.h"
#include "guarded.h"
#include "scene_handle.h"

using rt::Scene;

using rt::guarded;

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }
size_t rt_stats_size(void) { return sizeof(RtStats); }

void rt_default_params(RtParams* out) {
    if (!out) return;
    out->spp = 5;           // raytracer.comp:117
    out->max_bounces = 20;  // raytracer.comp:118
    out->accel = RT_ACCEL_BVH;
    out->strip_rows = 0;
    out->strip_count = 1;
    out->strip_index = 0;
    out->flags = 0;
    out->skybox = RT_SKY_GREY;  // raytracer.comp:575-577
}

int rt_scene_create(int preset, const char* assets_dir, RtScene** out) {
    if (!out || preset < RT_SCENE_EMPTY || preset > RT_SCENE_PRESET1_TEAPOT) return RT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    return guarded([&]() -> int {
        RtScene* s = new RtScene(preset, assets_dir ? assets_dir : "assets/");
        *out = s;
        // A model that failed to load leaves an empty mesh behind, as in the reference; tell the caller.
        return s->scene.last_error.empty() ? RT_OK : RT_ERR_IO;
    });
}

int rt_scene_destroy(RtScene* s) {
    delete s;
    return RT_OK;
}

static bool mat_ok(const Scene& sc, int32_t mat) { return mat >= 0 && mat < (int32_t)sc.materialVec.size(); }

int rt_scene_add_material(RtScene* s, const RtMaterial* m) {
    if (!s || !m) return RT_ERR_INVALID_ARGUMENT;
    return guarded([&]() -> int { return s->scene.addMaterial(*m); });
}

int rt_scene_add_sphere(RtScene* s, const float pos[3], float r, int32_t mat) {
    if (!s || !pos || !mat_ok(s->scene, mat)) return RT_ERR_INVALID_ARGUMENT;
    RtSphere sp;
    memset(&sp, 0, sizeof sp);
    memcpy(sp.pos, pos, 12);
    sp.r = r;
    sp.mat = mat;
    return guarded([&]() -> int { s->scene.addSphere(sp); return RT_OK; });
}

int rt_scene_add_triangle(RtScene* s, const float v0[3], const float v1[3], const float v2[3], int32_t mat) {
    if (!s || !v0 || !v1 || !v2 || !mat_ok(s->scene, mat)) return RT_ERR_INVALID_ARGUMENT;
    RtTriangle t;
    memset(&t, 0, sizeof t);
    memcpy(t.v0, v0, 12);
    memcpy(t.v1, v1, 12);
    memcpy(t.v2, v2, 12);
    t.mat = mat;
    return guarded([&]() -> int { s->scene.addTriangle(t); return RT_OK; });
}

int rt_scene_add_quad(RtScene* s, const float v0[3], const float v1[3], const float v2[3], const float v3[3], int32_t mat) {
    if (!s || !v0 || !v1 || !v2 || !v3 || !mat_ok(s->scene, mat)) return RT_ERR_INVALID_ARGUMENT;
    rt::Quad q;
    q.v0 = rtm::v3(v0[0], v0[1], v0[2]);
    q.v1 = rtm::v3(v1[0], v1[1], v1[2]);
    q.v2 = rtm::v3(v2[0], v2[1], v2[2]);
    q.v3 = rtm::v3(v3[0], v3[1], v3[2]);
    q.mat = mat;
    return guarded([&]() -> int { s->scene.addQuad(q); return RT_OK; });
}

int rt_scene_add_light(RtScene* s, const float pos_angle_aux[4], const float color_str[4], int32_t type) {
    if (!s || !pos_angle_aux || !color_str) return RT_ERR_INVALID_ARGUMENT;
    RtLight l;
    memset(&l, 0, sizeof l);
    memcpy(l.pos_angle_aux, pos_angle_aux, 16);
    memcpy(l.color_str, color_str, 16);
    l.type = type;
    return guarded([&]() -> int { s->scene.addLight(l); return RT_OK; });
}

int rt_scene_add_model(RtScene* s, const char* file_name, const float pos[3], float pitch, float yaw, float roll,
                       float scale, int32_t material) {
    if (!s || !file_name || !pos || !mat_ok(s->scene, material)) return RT_ERR_INVALID_ARGUMENT;
    return guarded([&]() -> int {
        rt::Model m{file_name, rtm::v3(pos[0], pos[1], pos[2]), pitch, yaw, roll, scale, material};
        return s->scene.addModel(m) ? RT_OK : RT_ERR_IO;
    });
}

int rt_scene_add_model_ex(RtScene* s, const char* file_name, const float pos[3], float pitch, float yaw, float roll,
                          float scale, int32_t material, uint32_t flags) {
    if (!s || !file_name || !pos || !mat_ok(s->scene, material)) return RT_ERR_INVALID_ARGUMENT;
    if (flags & ~(uint32_t)(RT_MODEL_FULL_GLTF | RT_MODEL_REBASE_INDICES)) return RT_ERR_INVALID_ARGUMENT;
    return guarded([&]() -> int {
        rt::Model m{file_name, rtm::v3(pos[0], pos[1], pos[2]), pitch, yaw, roll, scale, material};
        return s->scene.addModelEx(m, flags) ? RT_OK : RT_ERR_IO;
    });
}

int rt_scene_view(const RtScene* s, RtSceneView* out) {
    if (!s || !out) return RT_ERR_INVALID_ARGUMENT;
    const Scene& sc = s->scene;
    out->spheres = sc.sphereVec.data();     out->n_spheres = (int32_t)sc.sphereVec.size();
    out->materials = sc.materialVec.data(); out->n_materials = (int32_t)sc.materialVec.size();
    out->lights = sc.lightsVec.data();      out->n_lights = (int32_t)sc.lightsVec.size();
    out->triangles = sc.triangleVec.data(); out->n_triangles = (int32_t)sc.triangleVec.size();
    out->vertices = sc.vertexVec.data();    out->n_vertices = (int32_t)sc.vertexVec.size();
    out->indices = sc.indexVec.data();      out->n_indices = (int32_t)sc.indexVec.size();
    out->meshes = sc.meshVec.data();        out->n_meshes = (int32_t)sc.meshVec.size();
    out->total_spheres = sc.total_spheres;
    out->total_triangles = sc.total_triangles;
    out->total_meshes = sc.total_meshes;
    out->total_lights = sc.total_lights;
    out->lights_strength_sum = sc.lights_strength_sum;
    return RT_OK;
}

// main.cpp:1466-1497 updateUniformBuffer.  Quirk kept: the aspect ratio passed to
// perspective() is the INTEGER quotient width/height (main.cpp:1478); proj/projInv/viewproj
// are never read by the path tracer.  tanHalfFOV is computed in double and narrowed
// (main.cpp:1494: tan(glm::radians(fov) / 2.0)).
int rt_camera_look(RtCamera* out, const float pos[3], const float front[3], const float up[3], float fov_degrees,
                   int32_t width, int32_t height) {
    if (!out || !pos || !front || !up || width <= 0 || height <= 0) return RT_ERR_INVALID_ARGUMENT;
    rtm::vec3 p = rtm::v3(pos[0], pos[1], pos[2]);
    rtm::vec3 f = rtm::v3(front[0], front[1], front[2]);
    rtm::vec3 u = rtm::v3(up[0], up[1], up[2]);
    rtm::mat4 view = rtm::lookAt(p, p + f, u);
    rtm::mat4 viewInv = rtm::inverse(view);
    float aspect = (float)((uint32_t)width / (uint32_t)height);
    if (aspect == 0.0f) aspect = 1.0f;  // integer quotient of a portrait image; proj is unused anyway
    rtm::mat4 proj = rtm::perspective(rtm::radians(fov_degrees), aspect, 0.1f, 100.0f);
    rtm::mat4 projInv = rtm::inverse(proj);
    rtm::mat4 viewproj = rtm::mul(view, proj);
    memcpy(out->view, &view, 64);
    memcpy(out->viewInv, &viewInv, 64);
    memcpy(out->proj, &proj, 64);
    memcpy(out->projInv, &projInv, 64);
    memcpy(out->viewproj, &viewproj, 64);
    out->position[0] = p.x; out->position[1] = p.y; out->position[2] = p.z;
    out->tanHalfFOV = (float)std::tan((double)rtm::radians(fov_degrees) / 2.0);
    return RT_OK;
}

int rt_camera_default(RtCamera* out, int32_t width, int32_t height) {
    const float pos[3] = {2.75f, 2.75f, -6.0f};  // main.cpp:80
    const float front[3] = {0.0f, 0.0f, 1.0f};   // yaw 180, pitch 0 through updateCamera (main.cpp:554-565)
    const float up[3] = {0.0f, 1.0f, 0.0f};      // main.cpp:65,82
    return rt_camera_look(out, pos, front, up, 50.0f /* main.cpp:75 */, width, height);
}

// main.cpp:1424-1438 updatePushConstantsPre
int rt_push_constants(RtPushConstants* out, const RtScene* s, float time, uint32_t frame_count, int reset_accumulation) {
    if (!out || !s) return RT_ERR_INVALID_ARGUMENT;
    const Scene& sc = s->scene;
    out->time = time;
    out->frameCount = frame_count;
    out->total_lights = sc.total_lights;
    out->lights_strength_sum = sc.lights_strength_sum;
    out->world_up[0] = 0.0f; out->world_up[1] = 1.0f; out->world_up[2] = 0.0f;
    out->reset_frame_accumulation = reset_accumulation ? 1u : 0u;
    out->total_spheres = sc.total_spheres;
    out->total_triangles = sc.total_triangles;
    out->total_meshes = sc.total_meshes;
    return RT_OK;
}

static rt::SceneArrays arrays_of(const Scene& sc) {
    rt::SceneArrays a;
    a.spheres = sc.sphereVec.data();     a.n_spheres = sc.total_spheres;
    a.triangles = sc.triangleVec.data(); a.n_triangles = sc.total_triangles;
    a.vertices = sc.vertexVec.data();    a.n_vertices = (int32_t)sc.vertexVec.size();
    a.indices = sc.indexVec.data();      a.n_indices = (int32_t)sc.indexVec.size();
    a.meshes = sc.meshVec.data();        a.n_meshes = sc.total_meshes;
    return a;
}

int rt_scene_bvh_info(const RtScene* s, int32_t* n_nodes, int32_t* depth, int32_t* n_prims, float* sah_cost) {
    if (!s) return RT_ERR_INVALID_ARGUMENT;
    return guarded([&]() -> int {
    const rt::SceneArrays a = arrays_of(s->scene);
    rt::HostBvh bvh;
    std::string err;
    if (!rt::build_bvh(a, bvh, &err)) return RT_ERR_INVALID_ARGUMENT;
    if (!rt::validate_bvh(bvh, &err)) return RT_ERR_UNSUPPORTED;
    if (n_nodes) *n_nodes = (int32_t)bvh.nodes.size();
    if (depth) *depth = bvh.depth;
    if (n_prims) *n_prims = (int32_t)bvh.prim_order.size();
    if (sah_cost) *sah_cost = bvh.sah_cost;
    return RT_OK;
    });
}

// The four-wide form of the scene's tree (what the pooled kernel walks on large scenes), built and checked.
int rt_scene_wide_bvh_info(const RtScene* s, int32_t* n_nodes, int32_t* depth, int32_t* stack_need) {
    if (!s) return RT_ERR_INVALID_ARGUMENT;
    return guarded([&]() -> int {
    const rt::SceneArrays a = arrays_of(s->scene);
    rt::HostBvh bvh;
    rt::HostWideBvh wide;
    std::string err;
    if (!rt::build_bvh(a, bvh, &err)) return RT_ERR_INVALID_ARGUMENT;
    rt::collapse_wide(bvh, wide);
    if (!rt::validate_wide(bvh, wide, &err)) return RT_ERR_UNSUPPORTED;
    if (n_nodes) *n_nodes = (int32_t)wide.nodes.size();
    if (depth) *depth = wide.depth;
    if (stack_need) *stack_need = wide.stack_need;
    return RT_OK;
    });
}

// FNV-1a over the BVH's node array and leaf order (what rt_upload_scene would copy to the device).
int rt_scene_bvh_hash(const RtScene* s, uint64_t* hash) {
    if (!s || !hash) return RT_ERR_INVALID_ARGUMENT;
    return guarded([&]() -> int {
    const rt::SceneArrays a = arrays_of(s->scene);
    rt::HostBvh bvh;
    std::string err;
    if (!rt::build_bvh(a, bvh, &err)) return RT_ERR_INVALID_ARGUMENT;
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](const void* p, size_t n) {
        const unsigned char* b = (const unsigned char*)p;
        for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
    };
    mix(bvh.nodes.data(), bvh.nodes.size() * sizeof(rt::BvhNode));
    mix(bvh.prim_order.data(), bvh.prim_order.size() * sizeof(int32_t));
    *hash = h;
    return RT_OK;
    });
}

}  // extern "C"
