// scene.cpp — scene builders.  Behaviour follows the reference's src/scene.cpp; the scene
// *data* (material parameters, wall/block coordinates, sphere placements) is the
// reference's default content, kept here as tables.
#include "scene.h"

#include <cstdio>
#include <cstring>

#include "glb_loader.h"

namespace rt {
namespace {

using rtm::vec3;

// One material row: albedo rgba | subsurface rgba | specular_tint rgba | emission rgba |
// roughness, metallic, ior, trs_weight   (field meaning: definitions.hpp:45-73)
struct MatRow {
    float v[20];
};

RtMaterial material_from(const MatRow& r) {
    RtMaterial m;
    memcpy(&m, r.v, sizeof m);
    return m;
}

inline float gmax(float a, float b) { return (a < b) ? b : a; }  // glm::max
inline float gmin(float a, float b) { return (b < a) ? b : a; }  // glm::min
inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }

RtTriangle make_tri(vec3 a, vec3 b, vec3 c, int mat) {
    RtTriangle t;
    memset(&t, 0, sizeof t);
    t.v0[0] = a.x; t.v0[1] = a.y; t.v0[2] = a.z;
    t.v1[0] = b.x; t.v1[1] = b.y; t.v1[2] = b.z;
    t.v2[0] = c.x; t.v2[1] = c.y; t.v2[2] = c.z;
    t.mat = mat;
    return t;
}

RtSphere make_sphere(float x, float y, float z, float r, int mat) {
    RtSphere s;
    memset(&s, 0, sizeof s);
    s.pos[0] = x; s.pos[1] = y; s.pos[2] = z;
    s.r = r;
    s.mat = mat;
    return s;
}

// Model matrix of addModel (scene.cpp:820-827): T(pos) * yawPitchRoll(yaw,pitch,roll) * S(scale)
rtm::mat4 model_matrix(const Model& m) {
    float pitchRad = rtm::radians(m.pitch);
    float yawRad = rtm::radians(m.yaw);
    float rollRad = rtm::radians(m.roll);
    rtm::mat4 rot = rtm::yawPitchRoll(yawRad, pitchRad, rollRad);
    rtm::mat4 T = rtm::translate(rtm::identity(), m.pos);
    rtm::mat4 S = rtm::scale(rtm::identity(), rtm::v3(m.scale, m.scale, m.scale));
    return rtm::mul(rtm::mul(T, rot), S);
}

void transform_vertices(std::vector<RtVertex>& vs, const rtm::mat4& M) {
    for (auto& v : vs) {
        rtm::vec4 p = rtm::mul(M, rtm::v4(v.pos[0], v.pos[1], v.pos[2], 1.0f));
        v.pos[0] = p.x; v.pos[1] = p.y; v.pos[2] = p.z;
        v._pad = 0.0f;
    }
}

}  // namespace

// scene.cpp:25-35 — every buffer needs at least one element, so each vector starts with a
// placeholder that the first real add* call removes.  materialVec has no placeholder.
Scene::Scene(int preset, const std::string& assets) : assets_dir(assets) {
    RtLight l;    memset(&l, 0, sizeof l);   lightsVec.push_back(l);
    RtSphere s;   memset(&s, 0, sizeof s);   sphereVec.push_back(s);
    RtTriangle t; memset(&t, 0, sizeof t);   triangleVec.push_back(t);
    RtMeshInfo mi; memset(&mi, 0, sizeof mi); meshVec.push_back(mi);
    RtVertex v;   memset(&v, 0, sizeof v);   vertexVec.push_back(v);
    indexVec.push_back(0);
    if (!assets_dir.empty() && assets_dir.back() != '/') assets_dir.push_back('/');

    switch (preset) {
        case RT_SCENE_CORNELL_STAR: createCornellBox(); break;
        case RT_SCENE_PRESET1: createPreset1(false); break;
        case RT_SCENE_PRESET1_TEAPOT: createPreset1(true); break;
        case RT_SCENE_CORNELL_TEAPOT: createCornellTeapot(); break;
        case RT_SCENE_TEAPOT_GRID: createTeapotGrid(); break;
        default: break;
    }
}

// scene.cpp:716-729
void Scene::addSphere(RtSphere s) {
    if (materialVec[s.mat].emission_color[3] > 0.0f) {
        RtLight l;
        memset(&l, 0, sizeof l);
        l.pos_angle_aux[0] = s.pos[0]; l.pos_angle_aux[1] = s.pos[1]; l.pos_angle_aux[2] = s.pos[2];
        l.pos_angle_aux[3] = s.r;
        memcpy(l.color_str, materialVec[s.mat].emission_color, sizeof l.color_str);
        l.type = RT_LIGHT_SPHERE;
        addLight(l);
    }
    if (total_spheres == 0) sphereVec.pop_back();
    s._pad[0] = s._pad[1] = s._pad[2] = 0;
    sphereVec.push_back(s);
    total_spheres = (int)sphereVec.size();
}

// scene.cpp:731-749
int Scene::addMaterial(RtMaterial m) {
    for (int i = 0; i < 4; ++i) m.albedo[i] = gclamp(m.albedo[i], 0.0f, 1.0f);
    for (int i = 0; i < 4; ++i) m.subsurface[i] = gclamp(m.subsurface[i], 0.0f, 1.0f);
    for (int i = 0; i < 4; ++i) m.specular_tint[i] = gclamp(m.specular_tint[i], 0.0f, 1.0f);
    for (int i = 0; i < 3; ++i) m.emission_color[i] = gclamp(m.emission_color[i], 0.0f, 1.0f);  // alpha = strength, kept
    m.roughness = gclamp(m.roughness, 0.005f, 1.0f);
    m.metallic = gclamp(m.metallic, 0.0f, 1.0f);
    m.ior = gmax(m.ior, 0.0f);
    if (m.ior == 1.0f) m.ior = 1.00001f;
    m.trs_weight = gclamp(m.trs_weight, 0.0f, 1.0f);
    materialVec.push_back(m);
    return (int)materialVec.size() - 1;
}

// scene.cpp:751-763 — running sum of strengths = the CDF the shader binary-searches
void Scene::addLight(RtLight l) {
    if (lightsVec.size() == 0) {
        l.accumulated_str = l.color_str[3];
    } else {
        l.accumulated_str = lightsVec[lightsVec.size() - 1].accumulated_str + l.color_str[3];
    }
    if (total_lights == 0) lightsVec.pop_back();
    l._pad[0] = l._pad[1] = 0;
    lightsVec.push_back(l);
    total_lights = (int)lightsVec.size();
    lights_strength_sum += l.color_str[3];
}

// scene.cpp:765-782
void Scene::addTriangle(RtTriangle t) {
    vec3 v0 = rtm::v3(t.v0[0], t.v0[1], t.v0[2]);
    vec3 v1 = rtm::v3(t.v1[0], t.v1[1], t.v1[2]);
    vec3 v2 = rtm::v3(t.v2[0], t.v2[1], t.v2[2]);
    vec3 n = rtm::normalize(rtm::cross(v1 - v0, v2 - v0));
    t.normal[0] = n.x; t.normal[1] = n.y; t.normal[2] = n.z;
    t._p0 = t._p1 = t._p2 = 0.0f;

    if (total_triangles == 0) triangleVec.pop_back();
    triangleVec.push_back(t);
    total_triangles = (int)triangleVec.size();

    if (materialVec[t.mat].emission_color[3] > 0.0f) {
        RtLight l;
        memset(&l, 0, sizeof l);
        l.pos_angle_aux[0] = (float)(total_triangles - 1);
        memcpy(l.color_str, materialVec[t.mat].emission_color, sizeof l.color_str);
        l.type = RT_LIGHT_TRIANGLE;
        addLight(l);
    }
}

// scene.cpp:784-798
void Scene::addQuad(const Quad& q) {
    addTriangle(make_tri(q.v0, q.v1, q.v2, q.mat));
    addTriangle(make_tri(q.v2, q.v3, q.v0, q.mat));
}

// scene.cpp:801-846.  A load failure is reported and the model is recorded as an empty
// mesh, as in the reference (812-814); the return value additionally tells the caller.
bool Scene::addModel(const Model& model) { return addModelEx(model, 0u); }

// addModel with two optional departures from the reference (flags = RT_MODEL_*): the general GLB
// reader instead of the mesh-0 / primitive-0 one, and indices rebased onto this model's vertices
// so that several models can share the vertex buffer.
bool Scene::addModelEx(const Model& model, unsigned flags) {
    if (total_meshes == 0) {
        meshVec.pop_back();
        vertexVec.pop_back();
        indexVec.pop_back();
    }
    std::vector<RtVertex> vs;
    std::vector<uint32_t> is;
    std::string err;
    bool ok = (flags & RT_MODEL_FULL_GLTF) ? LoadModelFull(assets_dir + model.file_name, vs, is, &err)
                                           : LoadModel(assets_dir + model.file_name, vs, is, &err);
    if (!ok) {
        last_error = "Error loading model " + assets_dir + model.file_name + ": " + err;
        fprintf(stderr, "%s\n", last_error.c_str());
        vs.clear();
        is.clear();
    } else if (!quiet) {
        printf("Model %s%s loaded successfully!\nVertex count: %zu\nIndex count: %zu\n", assets_dir.c_str(),
               model.file_name.c_str(), vs.size(), is.size());
    }
    transform_vertices(vs, model_matrix(model));

    RtMeshInfo mi;
    mi.index_start = (uint32_t)indexVec.size();
    mi.index_end = (uint32_t)(indexVec.size() + is.size());
    mi.material = model.material;
    meshVec.push_back(mi);
    total_meshes = (int)meshVec.size();

    // Indices are appended as loaded (mesh-local), exactly like scene.cpp:844-845: a second
    // model's indices are NOT rebased onto its vertices, unless RT_MODEL_REBASE_INDICES asks for it.
    if (flags & RT_MODEL_REBASE_INDICES) {
        const uint32_t base = (uint32_t)vertexVec.size();
        for (uint32_t& i : is) i += base;
    }
    vertexVec.insert(vertexVec.end(), vs.begin(), vs.end());
    indexVec.insert(indexVec.end(), is.begin(), is.end());
    return ok;
}

// scene.cpp:848-860
void Scene::printSceneInfo() const {
    if (quiet) return;
    printf("Scene loaded\nNumber of spheres: %zu\nNumber of materials: %zu\nNumber of lights: %zu\n", sphereVec.size(),
           materialVec.size(), lightsVec.size());
    for (const auto& l : lightsVec)
        printf("Light {\n  pos_angle_aux: (%g, %g, %g, %g)\n  color_str: (%g, %g, %g, %g)\n  type: %d\n  accumulated_str: %g\n}\n",
               l.pos_angle_aux[0], l.pos_angle_aux[1], l.pos_angle_aux[2], l.pos_angle_aux[3], l.color_str[0],
               l.color_str[1], l.color_str[2], l.color_str[3], l.type, l.accumulated_str);
    printf("Number of triangles: %zu\nNumber of models: %zu\nNumber of vertices: %zu\nNumber of indices: %zu\n",
           triangleVec.size(), meshVec.size(), vertexVec.size(), indexVec.size());
}

// ------------------------------------------------------------------------------------
// Cornell box content (scene.cpp:353-679): 13 materials, 18 quads, 4 spheres.
// Material order defines the indices: white 0, red 1, green 2, white_light 3, blue_light 4,
// glass 5, frosted_blue_glass 6, frosted_glass 7, gold 8, yellowMetal 9, mirror 10,
// blueMatte 11, weird 12.
static const MatRow kCornellMaterials[13] = {
    /* white      */ {{1, 1, 1, 1,              0, 0, 0, 0,       1, 1, 1, 0.5f,              0, 0, 0, 0,                1.0f, 0.0f, 1.5f, 0.0f}},
    /* red        */ {{0.75f, 0.15f, 0.15f, 1,  0, 0, 0, 0,       1, 1, 1, 0.5f,              0, 0, 0, 0,                1.0f, 0.0f, 1.5f, 0.0f}},
    /* green      */ {{0.15f, 0.75f, 0.15f, 1,  0, 0, 0, 0,       1, 1, 1, 0.5f,              0, 0, 0, 0,                1.0f, 0.0f, 1.5f, 0.0f}},
    /* white_light*/ {{1, 1, 1, 1,              1, 1, 1, 1,       1, 1, 1, 0.5f,              1, 1, 1, 20.0f,            0.0f, 1.0f, 1.33f, 0.0f}},
    /* blue_light */ {{1, 1, 1, 1,              1, 1, 1, 1,       1, 1, 1, 0.5f,              0.101f, 0.643f, 0.835f, 2.0f, 0.0f, 1.0f, 1.33f, 0.0f}},
    /* glass      */ {{1, 1, 1, 1,              1, 1, 1, 1,       1, 1, 1, 0.5f,              0, 0, 0, 0,                0.0f, 0.0f, 1.5f, 1.0f}},
    /* frosted_blue_glass */ {{1, 1, 1, 1,      0.3f, 0.3f, 1, 1, 1, 1, 1, 0.5f,              0, 0, 0, 0,                0.3f, 0.0f, 1.5f, 1.0f}},
    /* frosted_glass */ {{1, 1, 1, 1,           1, 1, 1, 1,       1, 1, 1, 0.0f,              0, 0, 0, 0,                0.0f, 0.0f, 1.5f, 1.0f}},
    /* gold       */ {{1.000f, 0.720f, 0.315f, 1, 0, 0, 0, 0,     1.000f, 0.973f, 0.597f, 0.5f, 0, 0, 0, 0,              0.0f, 0.0f, 0.47f, 0.0f}},
    /* yellowMetal*/ {{0.945f, 0.949f, 0.212f, 1, 0, 0, 0, 0,     1, 1, 1, 0.5f,              0, 0, 0, 0,                0.1f, 1.0f, 1.5f, 0.0f}},
    /* mirror     */ {{1, 1, 1, 1,              1, 1, 1, 1,       1, 1, 1, 0.5f,              0, 0, 0, 0,                0.0f, 1.0f, 1.33f, 0.0f}},
    /* blueMatte  */ {{0.208f, 0.612f, 0.8f, 1, 0, 0, 0, 0,       1, 1, 1, 0.8f,              0, 0, 0, 0,                0.2f, 0.0f, 1.5f, 0.0f}},
    /* weird      */ {{0.267f, 0.588f, 0.475f, 0.8f, 1, 0.3f, 0.3f, 1, 0, 1, 1, 1,           0, 0, 0, 0,                0.2f, 0.5f, 0.8f, 0.8f}},
};
enum { kWhite = 0, kRed, kGreen, kWhiteLight, kBlueLight, kGlass, kFrostedBlueGlass, kFrostedGlass, kGold,
       kYellowMetal, kMirror, kBlueMatte, kWeird };

struct QuadRow {
    float p[12];
    int mat;
};
// wall mirror, translucent wall, floor, ceiling, back, right (green), left (red), light
static const QuadRow kCornellWalls[8] = {
    {{0.000f, 0.0f, 4.592f,   2.000f, 0.0f, 5.592f,   2.000f, 5.49f, 5.592f,  0.000f, 5.49f, 4.592f}, kMirror},
    {{0.000f, 4.0f, 4.0f,     0.9f, 4.0f, 4.0f,       0.9f, 5.0f, 4.0f,       0.000f, 5.0f, 4.0f}, kFrostedGlass},
    {{5.528f, -0.001f, 0.000f, 0.000f, -0.001f, 0.000f, 0.000f, -0.001f, 5.592f, 5.496f, -0.001f, 5.592f}, kWhite},
    {{5.560f, 5.488f, 0.000f, 5.560f, 5.488f, 5.592f, 0.000f, 5.488f, 5.592f, 0.000f, 5.488f, 0.000f}, kWhite},
    {{5.496f, 0.000f, 5.592f, 0.000f, 0.000f, 5.592f, 0.000f, 5.488f, 5.592f, 5.560f, 5.488f, 5.592f}, kWhite},
    {{0.000f, 0.000f, 5.592f, 0.000f, 0.000f, 0.000f, 0.000f, 5.488f, 0.000f, 0.000f, 5.488f, 5.592f}, kGreen},
    {{5.528f, 0.000f, 0.000f, 5.496f, 0.000f, 5.592f, 5.560f, 5.488f, 5.592f, 5.560f, 5.488f, 0.000f}, kRed},
    {{3.430f, 5.47f, 2.270f,  3.430f, 5.47f, 3.320f,  2.130f, 5.47f, 3.320f,  2.130f, 5.47f, 2.270f}, kWhiteLight},
};
// short block (blueMatte) then tall block (glass), five faces each
static const QuadRow kCornellBlocks[10] = {
    {{1.300f, 1.650f, 0.650f, 0.820f, 1.650f, 2.250f, 2.400f, 1.650f, 2.720f, 2.900f, 1.650f, 1.140f}, kBlueMatte},
    {{2.900f, 0.000f, 1.140f, 2.900f, 1.650f, 1.140f, 2.400f, 1.650f, 2.720f, 2.400f, 0.000f, 2.720f}, kBlueMatte},
    {{1.300f, 0.000f, 0.650f, 1.300f, 1.650f, 0.650f, 2.900f, 1.650f, 1.140f, 2.900f, 0.000f, 1.140f}, kBlueMatte},
    {{0.820f, 0.000f, 2.250f, 0.820f, 1.650f, 2.250f, 1.300f, 1.650f, 0.650f, 1.300f, 0.000f, 0.650f}, kBlueMatte},
    {{2.400f, 0.000f, 2.720f, 2.400f, 1.650f, 2.720f, 0.820f, 1.650f, 2.250f, 0.820f, 0.000f, 2.250f}, kBlueMatte},
    {{4.230f, 3.300f, 2.470f, 2.650f, 3.300f, 2.960f, 3.140f, 3.300f, 4.560f, 4.720f, 3.300f, 4.060f}, kGlass},
    {{4.230f, 0.000f, 2.470f, 4.230f, 3.300f, 2.470f, 4.720f, 3.300f, 4.060f, 4.720f, 0.000f, 4.060f}, kGlass},
    {{4.720f, 0.000f, 4.060f, 4.720f, 3.300f, 4.060f, 3.140f, 3.300f, 4.560f, 3.140f, 0.000f, 4.560f}, kGlass},
    {{3.140f, 0.000f, 4.560f, 3.140f, 3.300f, 4.560f, 2.650f, 3.300f, 2.960f, 2.650f, 0.000f, 2.960f}, kGlass},
    {{2.650f, 0.000f, 2.960f, 2.650f, 3.300f, 2.960f, 4.230f, 3.300f, 2.470f, 4.230f, 0.000f, 2.470f}, kGlass},
};

static Quad quad_from(const QuadRow& r) {
    Quad q;
    q.v0 = rtm::v3(r.p[0], r.p[1], r.p[2]);
    q.v1 = rtm::v3(r.p[3], r.p[4], r.p[5]);
    q.v2 = rtm::v3(r.p[6], r.p[7], r.p[8]);
    q.v3 = rtm::v3(r.p[9], r.p[10], r.p[11]);
    q.mat = r.mat;
    return q;
}

// Order of insertion matters (it fixes primitive and light indices): materials, the eight
// wall quads, the four spheres, then the ten block quads (scene.cpp:353-679).
void Scene::cornellMaterialsAndWalls(int* gold_out) {
    for (const auto& row : kCornellMaterials) addMaterial(material_from(row));
    for (const auto& w : kCornellWalls) addQuad(quad_from(w));
    addSphere(make_sphere(5.0f, 0.6f, 1.0f, 0.6f, kMirror));             // ground sphere 1
    addSphere(make_sphere(3.0f, 0.6f, 2.5f, 0.6f, kWeird));              // ground sphere 2
    addSphere(make_sphere(0.8f, 0.15f, 1.3f, 0.15f, kBlueLight));        // ground light
    addSphere(make_sphere(1.85f, 2.150f, 1.5f, 0.5f, kFrostedBlueGlass)); // on top of the short block
    for (const auto& b : kCornellBlocks) addQuad(quad_from(b));
    *gold_out = kGold;
}

// scene.cpp:350-695
void Scene::createCornellBox() {
    int gold;
    cornellMaterialsAndWalls(&gold);
    Model star{"star.glb", rtm::v3(3.5f, 4.0f, 3.1f), /*pitch*/ 0.0f, /*yaw*/ 18.0f, /*roll*/ 180.0f, 0.1f, gold};
    addModel(star);
    printSceneInfo();
}

// The benchmark scene: the same box with teapot.glb (6320 triangles) hovering above the
// blocks.  pitch 90 maps the file's -z axis to +y (the loader ignores the glTF node
// rotation, as the reference's does).
void Scene::createCornellTeapot() {
    int gold;
    cornellMaterialsAndWalls(&gold);
    Model teapot{"teapot.glb", rtm::v3(2.75f, 3.5f, 2.8f), /*pitch*/ 90.0f, /*yaw*/ 30.0f, /*roll*/ 0.0f, 0.45f, gold};
    addModel(teapot);
    printSceneInfo();
}

// ~1.07 M triangles: 13 x 13 teapots over a large floor, baked into ONE mesh whose indices
// are rebased per copy (the reference's addModel would not rebase them: scene.cpp:835-845).
void Scene::createTeapotGrid() {
    int gold;
    for (const auto& row : kCornellMaterials) addMaterial(material_from(row));
    gold = kGold;
    Quad floor;
    floor.v0 = rtm::v3(40.0f, -0.001f, -10.0f);
    floor.v1 = rtm::v3(-40.0f, -0.001f, -10.0f);
    floor.v2 = rtm::v3(-40.0f, -0.001f, 60.0f);
    floor.v3 = rtm::v3(40.0f, -0.001f, 60.0f);
    floor.mat = kWhite;
    addQuad(floor);
    Quad lamp;  // a large emissive panel high above the grid
    lamp.v0 = rtm::v3(10.0f, 14.0f, 6.0f);
    lamp.v1 = rtm::v3(10.0f, 14.0f, 16.0f);
    lamp.v2 = rtm::v3(-4.0f, 14.0f, 16.0f);
    lamp.v3 = rtm::v3(-4.0f, 14.0f, 6.0f);
    lamp.mat = kWhiteLight;
    addQuad(lamp);

    std::vector<RtVertex> base_v;
    std::vector<uint32_t> base_i;
    std::string err;
    if (!LoadModel(assets_dir + "teapot.glb", base_v, base_i, &err)) {
        last_error = "Error loading model " + assets_dir + "teapot.glb: " + err;
        fprintf(stderr, "%s\n", last_error.c_str());
        return;
    }
    if (total_meshes == 0) {
        meshVec.pop_back();
        vertexVec.pop_back();
        indexVec.pop_back();
    }
    RtMeshInfo mi;
    mi.index_start = (uint32_t)indexVec.size();
    mi.material = gold;
    const int N = 13;
    for (int gz = 0; gz < N; ++gz) {
        for (int gx = 0; gx < N; ++gx) {
            Model m{"teapot.glb", rtm::v3(-6.85f + 1.6f * (float)gx, 0.0f, 1.0f + 1.6f * (float)gz), 90.0f,
                    (float)((gx * 37 + gz * 101) % 360), 0.0f, 0.2f, gold};
            std::vector<RtVertex> vs = base_v;
            transform_vertices(vs, model_matrix(m));
            uint32_t rebase = (uint32_t)vertexVec.size();
            vertexVec.insert(vertexVec.end(), vs.begin(), vs.end());
            for (uint32_t i : base_i) indexVec.push_back(i + rebase);
        }
    }
    mi.index_end = (uint32_t)indexVec.size();
    meshVec.push_back(mi);
    total_meshes = (int)meshVec.size();
    printSceneInfo();
}

// ------------------------------------------------------------------------------------
// Sphere scene (scene.cpp:37-348): 15 materials, 13 spheres, 2 triangles.  The teapot
// model is declared in the reference (333-341) but its addModel is commented out (343).
static const MatRow kPreset1Materials[15] = {
    /* ground       */ {{0.129f, 0.388f, 0.082f, 1, 0, 0, 0, 0,  0, 0, 0, 0,        0, 0, 0, 0,                 1.0f, 0.0f, 1.5f, 0.0f}},
    /* redMatte     */ {{1, 0, 0, 1,                0, 0, 0, 0,  1, 1, 1, 0.5f,     0, 0, 0, 0,                 0.5f, 0.0f, 1.5f, 0.0f}},
    /* gold         */ {{1.000f, 0.720f, 0.315f, 1, 0, 0, 0, 0,  1.000f, 0.973f, 0.597f, 0.5f, 0, 0, 0, 0,     0.0f, 0.0f, 0.47f, 0.0f}},
    /* blueLight    */ {{1, 0, 0, 1,                0, 0, 0, 0,  1, 1, 1, 0.5f,     0, 1, 0.984f, 4.0f,         0.0f, 0.0f, 1.5f, 0.0f}},
    /* cloudyGlass  */ {{1, 1, 1, 1,                1, 1, 1, 1,  1, 1, 1, 0.5f,     0, 0, 0, 0,                 0.0f, 0.0f, 1.52f, 1.0f}},
    /* blueMatte    */ {{0.208f, 0.612f, 0.8f, 1,   0, 0, 0, 0,  1, 1, 1, 0.5f,     0, 0, 0, 0,                 0.2f, 0.0f, 1.5f, 0.0f}},
    /* redMetal     */ {{0.82f, 0.118f, 0.118f, 1,  0, 0, 0, 0,  1, 1, 1, 0.5f,     0, 0, 0, 0,                 0.8f, 1.0f, 1.5f, 0.0f}},
    /* yellowMetal  */ {{0.945f, 0.949f, 0.212f, 1, 0, 0, 0, 0,  1, 1, 1, 0.5f,     0, 0, 0, 0,                 0.1f, 1.0f, 1.5f, 0.0f}},
    /* dielectric075*/ {{1, 1, 1, 1,                1, 1, 1, 1,  1, 1, 1, 0.5f,     0, 0, 0, 0,                 0.0f, 0.0f, 0.75f, 1.0f}},
    /* dielectric133*/ {{1, 1, 1, 1,                1, 1, 1, 1,  1, 1, 1, 0.5f,     0, 0, 0, 0,                 0.0f, 0.0f, 1.33f, 1.0f}},
    /* glass        */ {{1, 1, 1, 1,                1, 1, 1, 1,  1, 1, 1, 0.5f,     0, 0, 0, 0,                 0.3f, 0.0f, 1.5f, 1.0f}},
    /* mirror       */ {{1, 1, 1, 1,                1, 1, 1, 1,  1, 1, 1, 0.5f,     0, 0, 0, 0,                 0.0f, 1.0f, 1.33f, 0.0f}},
    /* white_light  */ {{1, 1, 1, 1,                1, 1, 1, 1,  1, 1, 1, 0.5f,     1, 1, 1, 4.0f,              0.0f, 1.0f, 1.33f, 0.0f}},
    /* blue_light   */ {{1, 1, 1, 1,                1, 1, 1, 1,  1, 1, 1, 0.5f,     0.101f, 0.643f, 0.835f, 4.0f, 0.0f, 1.0f, 1.33f, 0.0f}},
    /* warm_light   */ {{1, 1, 1, 1,                1, 1, 1, 1,  1, 1, 1, 0.5f,     0.984f, 0.882f, 0.337f, 4.0f, 0.0f, 1.0f, 1.33f, 0.0f}},
};
enum { pGround = 0, pRedMatte, pGold, pBlueLight, pCloudyGlass, pBlueMatte, pRedMetal, pYellowMetal, pDielectric075,
       pDielectric133, pGlass, pMirror, pWhiteLight, pBlueLight2, pWarmLight };

void Scene::createPreset1(bool with_teapot) {
    for (const auto& row : kPreset1Materials) addMaterial(material_from(row));
    static const struct { float x, y, z, r; int mat; } spheres[13] = {
        {0.0f, -1000.0f, -10.0f, 999.0f, pGround},
        {0.0f, 0.0f, -10.0f, 1.0f, pBlueMatte},
        {0.9f, -0.8f, -10.0f, 0.2f, pMirror},
        {0.7f, -0.8f, -10.2f, 0.2f, pMirror},
        {3.0f, 0.0f, -10.0f, 1.0f, pRedMetal},
        {-3.0f, 0.0f, -10.0f, 1.0f, pYellowMetal},
        {0.0f, 0.01f, -7.8f, 1.0f, pGlass},
        {0.0f, 0.01f, -7.8f, 0.8f, pDielectric075},
        {2.005f, 0.01f, -7.8f, 1.0f, pDielectric133},
        {-2.005f, 0.01f, -7.8f, 1.0f, pDielectric075},
        {-2.5f, -0.7f, -6.0f, 0.3f, pWarmLight},
        {-1.0f, -0.7f, -10.0f, 0.3f, pWhiteLight},
        {1.0f, -0.7f, -10.0f, 0.3f, pBlueLight2},
    };
    for (const auto& s : spheres) addSphere(make_sphere(s.x, s.y, s.z, s.r, s.mat));
    addTriangle(make_tri(rtm::v3(0.0f, -1.0f, -5.0f), rtm::v3(2.5f, 2.0f, -5.0f), rtm::v3(-2.5f, 2.0f, -5.0f), pBlueMatte));
    addTriangle(make_tri(rtm::v3(0.0f, -1.0f, -15.0f), rtm::v3(2.5f, 2.0f, -15.0f), rtm::v3(-2.5f, 2.0f, -15.0f), pBlueMatte));
    if (with_teapot) {  // the model scene.cpp:333-341 defines; its addModel call (343) is commented out there
        Model teapot{"teapot.glb", rtm::v3(0.0f, -1.0f, 10.0f), /*pitch*/ 90.0f, /*yaw*/ 0.0f, /*roll*/ 0.0f, 1.0f, pBlueMatte};
        addModel(teapot);
    }
    printSceneInfo();
}

}  // namespace rt
