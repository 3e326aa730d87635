// scene.h — host scene container and builders with the surface of the reference's
// Scene class (src/scene.hpp:8-37): the seven public vectors that become the device
// scene buffers, the total_* counters, lights_strength_sum, and the add* builders.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/rt_abi.h"
#include "rtm.h"

namespace rt {

// Host-only helper records (reference: src/definitions.hpp:32-38 Quad, 89-97 Model).
struct Quad {
    rtm::vec3 v0, v1, v2, v3;
    int mat;
};
struct Model {
    std::string file_name;
    rtm::vec3 pos;
    float pitch, yaw, roll;  // degrees
    float scale;
    int material;
};

class Scene {
public:
    // scene.hpp:10-21
    std::vector<RtSphere> sphereVec;
    std::vector<RtMaterial> materialVec;
    std::vector<RtLight> lightsVec;
    std::vector<RtTriangle> triangleVec;
    std::vector<RtVertex> vertexVec;
    std::vector<uint32_t> indexVec;
    std::vector<RtMeshInfo> meshVec;
    float lights_strength_sum = 0.0f;
    int total_lights = 0;
    int total_spheres = 0;
    int total_triangles = 0;
    int total_meshes = 0;

    std::string assets_dir = "assets/";  // scene.hpp:6 ASSETS_DIRECTORY
    std::string last_error;
    bool quiet = true;  // the reference prints scene info to stdout (scene.cpp:848-860)

    // The reference constructor always builds the Cornell box (scene.cpp:25-35); here the
    // preset is a parameter.  RT_SCENE_EMPTY leaves only the placeholder elements.
    explicit Scene(int preset = RT_SCENE_CORNELL_STAR, const std::string& assets = "assets/");

    void createPreset1(bool with_teapot);  // scene.cpp:37-348; the teapot of 333-343 is optional as there
    void createCornellBox();    // scene.cpp:350-695
    void createCornellTeapot(); // Cornell box geometry with teapot.glb instead of star.glb
    void createTeapotGrid();    // 13 x 13 teapots baked into one mesh, rebased indices

    // scene.cpp:716-846 (private in the reference; public here so the C ABI can reach them)
    void addSphere(RtSphere s);
    int addMaterial(RtMaterial m);
    void addLight(RtLight l);
    void addTriangle(RtTriangle t);
    void addQuad(const Quad& q);
    bool addModel(const Model& model);
    bool addModelEx(const Model& model, unsigned flags);  // flags: RT_MODEL_* (rt_abi.h)
    void printSceneInfo() const;

private:
    void cornellMaterialsAndWalls(int* gold_out);
};

}  // namespace rt
