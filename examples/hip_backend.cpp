// hip_backend.cpp — the binding a maintainer of the reference application would add: the
// application keeps its own scene vectors (element layouts as the shader declares them,
// raytracer.comp:47-106, with a 12-byte vec3) and its camera / push-constant blocks, and the
// Vulkan members of RaytracingApp are replaced by one small class on the C ABI.
//
//   make -C examples        (g++ -std=c++17 -I include ... -L raytracer_amd/lib -lraytracer_amd)
//   examples/hip_backend          layout checks only (no GPU needed)
//   examples/hip_backend run      renders the default scene from application-owned vectors and
//                                 from the library's own scene object; the images must be equal
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_abi.h"

namespace app {  // what the application side looks like (its own types, not the library's)

struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct alignas(16) Sphere { vec3 pos; float r; int mat; };
struct alignas(16) Triangle { alignas(16) vec3 v0; alignas(16) vec3 v1; alignas(16) vec3 v2; alignas(16) vec3 normal; int mat; };
struct alignas(16) Vertex { vec3 pos; };
struct alignas(16) Material { vec4 albedo, subsurface, specular_tint, emission_color; float roughness, metallic, ior, trs_weight; };
struct alignas(16) Light { vec4 pos_angle_aux, color_str; int type; float accumulated_str; };
struct MeshInfo { uint32_t index_start, index_end; int material; };

struct Scene {  // the seven vectors the application uploads, in binding order 1..7, and its counters
    std::vector<Sphere> sphereVec;
    std::vector<Material> materialVec;
    std::vector<Light> lightsVec;
    std::vector<Triangle> triangleVec;
    std::vector<Vertex> vertexVec;
    std::vector<uint32_t> indexVec;
    std::vector<MeshInfo> meshVec;
    int total_spheres = 0, total_triangles = 0, total_meshes = 0, total_lights = 0;
    float lights_strength_sum = 0.0f;
};

struct PushConstants {  // host-side block with the 1-byte bool the application uses
    float time; int frameCount; int total_lights; float lights_strength_sum; vec3 world_up;
    bool reset_frame_accumulation; int total_spheres, total_triangles, total_meshes;
};

}  // namespace app

// The element layouts must be the ABI's, or the casts below are wrong.
static_assert(sizeof(app::Sphere) == sizeof(RtSphere) && offsetof(app::Sphere, r) == offsetof(RtSphere, r) && offsetof(app::Sphere, mat) == offsetof(RtSphere, mat), "Sphere");
static_assert(sizeof(app::Triangle) == sizeof(RtTriangle) && offsetof(app::Triangle, v1) == offsetof(RtTriangle, v1) && offsetof(app::Triangle, v2) == offsetof(RtTriangle, v2) &&
              offsetof(app::Triangle, normal) == offsetof(RtTriangle, normal) && offsetof(app::Triangle, mat) == offsetof(RtTriangle, mat), "Triangle");
static_assert(sizeof(app::Vertex) == sizeof(RtVertex), "Vertex");
static_assert(sizeof(app::Material) == sizeof(RtMaterial) && offsetof(app::Material, roughness) == offsetof(RtMaterial, roughness) && offsetof(app::Material, trs_weight) == offsetof(RtMaterial, trs_weight), "Material");
static_assert(sizeof(app::Light) == sizeof(RtLight) && offsetof(app::Light, type) == offsetof(RtLight, type) && offsetof(app::Light, accumulated_str) == offsetof(RtLight, accumulated_str), "Light");
static_assert(sizeof(app::MeshInfo) == sizeof(RtMeshInfo), "MeshInfo");

struct HipBackend {
    RtContext* ctx = nullptr;

    void init(const app::Scene& scene, int w, int h) {  // initVulkan(), main.cpp:568-588
        check(rt_create(0, &ctx));
        check(rt_upload_scene(ctx,
            reinterpret_cast<const RtSphere*>(scene.sphereVec.data()),     scene.total_spheres,
            reinterpret_cast<const RtMaterial*>(scene.materialVec.data()), (int32_t)scene.materialVec.size(),
            reinterpret_cast<const RtLight*>(scene.lightsVec.data()),      scene.total_lights,
            reinterpret_cast<const RtTriangle*>(scene.triangleVec.data()), scene.total_triangles,
            reinterpret_cast<const RtVertex*>(scene.vertexVec.data()),     (int32_t)scene.vertexVec.size(),
            scene.indexVec.data(),                                         (int32_t)scene.indexVec.size(),
            reinterpret_cast<const RtMeshInfo*>(scene.meshVec.data()),     scene.total_meshes));
        check(rt_resize(ctx, w, h));
    }

    // drawFrame(), main.cpp:1308-1392: the camera block updateUniformBuffer fills and the push
    // constants updatePushConstantsPre fills
    void draw(const RtCamera& cam, const app::PushConstants& p, const RtParams& params, void* bgra_out) {
        RtPushConstants pc{};
        pc.time = p.time; pc.frameCount = (uint32_t)p.frameCount;
        pc.total_lights = p.total_lights; pc.lights_strength_sum = p.lights_strength_sum;
        pc.world_up[0] = p.world_up.x; pc.world_up[1] = p.world_up.y; pc.world_up[2] = p.world_up.z;
        pc.reset_frame_accumulation = p.reset_frame_accumulation ? 1u : 0u;  // the host bool is one byte: widen it
        pc.total_spheres = p.total_spheres; pc.total_triangles = p.total_triangles; pc.total_meshes = p.total_meshes;
        check(rt_render_frame(ctx, &cam, &pc, &params));
        if (bgra_out) check(rt_read_output_bgra8(ctx, bgra_out));  // then blit / present as before
    }

    void resize(int w, int h) { check(rt_resize(ctx, w, h)); }  // recreateSwapChain(), main.cpp:1396-1420
    void shutdown() { if (ctx) rt_destroy(ctx); ctx = nullptr; }
    void check(int rc) { if (rc != RT_OK) throw std::runtime_error(rt_last_error(ctx)); }  // the application throws on VkResult != VK_SUCCESS
};

namespace {

template <typename A, typename R>
void copy_vec(std::vector<A>& dst, const R* src, int n) {
    static_assert(sizeof(A) == sizeof(R), "layouts differ");
    dst.resize((size_t)n);
    if (n > 0) std::memcpy(dst.data(), src, (size_t)n * sizeof(A));
}

// An application scene holding the default Cornell box + star: the library's builder produces
// the arrays, the application owns the copies (as its Scene object would).
app::Scene application_scene(const RtScene* built) {
    RtSceneView v;
    if (rt_scene_view(built, &v) != RT_OK) throw std::runtime_error("rt_scene_view");
    app::Scene s;
    copy_vec(s.sphereVec, v.spheres, v.n_spheres);
    copy_vec(s.materialVec, v.materials, v.n_materials);
    copy_vec(s.lightsVec, v.lights, v.n_lights);
    copy_vec(s.triangleVec, v.triangles, v.n_triangles);
    copy_vec(s.vertexVec, v.vertices, v.n_vertices);
    copy_vec(s.indexVec, v.indices, v.n_indices);
    copy_vec(s.meshVec, v.meshes, v.n_meshes);
    s.total_spheres = v.total_spheres; s.total_triangles = v.total_triangles; s.total_meshes = v.total_meshes;
    s.total_lights = v.total_lights; s.lights_strength_sum = v.lights_strength_sum;
    return s;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2 || std::string(argv[1]) != "run") {
        std::printf("layouts agree with rt_abi.h (ABI version %d)\n", rt_abi_version());
        return 0;
    }
    const int w = 96, h = 54, frames = 2;
    try {
        const char* assets = std::getenv("RT_ASSETS_DIR");
        RtScene* built = nullptr;
        if (rt_scene_create(RT_SCENE_CORNELL_STAR, assets ? assets : "assets/", &built) != RT_OK) throw std::runtime_error("rt_scene_create");
        const app::Scene scene = application_scene(built);

        RtParams params;
        rt_default_params(&params);
        params.spp = 2; params.max_bounces = 4;
        RtCamera cam;
        rt_camera_default(&cam, w, h);

        // (a) the application's own vectors through HipBackend
        HipBackend backend;
        backend.init(scene, w, h);
        std::vector<uint8_t> img_a((size_t)w * h * 4), img_b(img_a.size());
        for (int f = 0; f < frames; ++f) {
            app::PushConstants p{};
            p.time = 0.0f; p.frameCount = f; p.total_lights = scene.total_lights; p.lights_strength_sum = scene.lights_strength_sum;
            p.world_up = {0.0f, 1.0f, 0.0f}; p.reset_frame_accumulation = (f == 0);
            p.total_spheres = scene.total_spheres; p.total_triangles = scene.total_triangles; p.total_meshes = scene.total_meshes;
            backend.draw(cam, p, params, f + 1 == frames ? img_a.data() : nullptr);
        }
        backend.shutdown();

        // (b) the library's scene object and push-constant helper
        RtContext* ctx = nullptr;
        if (rt_create(0, &ctx) != RT_OK || rt_upload_scene_from(ctx, built) != RT_OK || rt_resize(ctx, w, h) != RT_OK) throw std::runtime_error(rt_last_error(ctx));
        for (int f = 0; f < frames; ++f) {
            RtPushConstants pc;
            rt_push_constants(&pc, built, 0.0f, (uint32_t)f, f == 0);
            if (rt_render_frame(ctx, &cam, &pc, &params) != RT_OK) throw std::runtime_error(rt_last_error(ctx));
        }
        if (rt_read_output_bgra8(ctx, img_b.data()) != RT_OK) throw std::runtime_error(rt_last_error(ctx));
        rt_destroy(ctx);

        // (c) the same frames shared by the GPUs of the node in row strips: RT_DEVICES=n (default 1) devices
        // 0..n-1, one all-gather over xGMI per frame; every device ends up with the whole picture
        const char* nd = std::getenv("RT_DEVICES");
        const int n_dev = nd ? std::atoi(nd) : 1;
        std::vector<int> ordinals;
        for (int i = 0; i < (n_dev > 0 ? n_dev : 1); ++i) ordinals.push_back(i);
        RtGroup* grp = nullptr;
        if (rt_group_create((int)ordinals.size(), ordinals.data(), &grp) != RT_OK) throw std::runtime_error(rt_last_error(nullptr));
        if (rt_group_upload_scene_from(grp, built) != RT_OK || rt_group_resize(grp, w, h) != RT_OK) throw std::runtime_error(rt_group_last_error(grp));
        for (int f = 0; f < frames; ++f) {
            RtPushConstants pc;
            rt_push_constants(&pc, built, 0.0f, (uint32_t)f, f == 0);
            if (rt_group_render_frame(grp, &cam, &pc, &params, RT_GROUP_STRIPS, 9) != RT_OK) throw std::runtime_error(rt_group_last_error(grp));
        }
        std::vector<uint8_t> img_c(img_a.size());
        RtContext* last = nullptr;
        rt_group_context(grp, (int)ordinals.size() - 1, &last);  // any device holds the whole frame
        if (rt_read_output_bgra8(last, img_c.data()) != RT_OK) throw std::runtime_error(rt_last_error(last));
        rt_group_destroy(grp);
        rt_scene_destroy(built);
        if (img_c != img_b) {
            std::printf("group of %zu device(s) in row strips: DIFFERENT from one device\n", ordinals.size());
            return 1;
        }
        std::printf("group of %zu device(s) in row strips: identical to one device\n", ordinals.size());

        size_t lit = 0;
        for (size_t i = 0; i < img_a.size(); i += 4) lit += (img_a[i] | img_a[i + 1] | img_a[i + 2]) != 0;
        const bool same = img_a == img_b;
        std::printf("%dx%d, %d frames: %zu lit pixels, application vectors vs library scene: %s\n", w, h, frames, lit, same ? "identical" : "DIFFERENT");
        return (same && lit > (size_t)w * h / 2) ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
