// rt_layout_check.cpp — compile-time proof that the C ABI structs have the byte layout of
// the shader's buffer blocks: std430 element strides and member offsets of
// shaders/raytracer.comp:47-106, the std140 Camera block (27-35) and the push-constant
// block (127-137).
#include <cstddef>

#include "../../include/rt_abi.h"

static_assert(sizeof(RtSphere) == 32 && offsetof(RtSphere, r) == 12 && offsetof(RtSphere, mat) == 16, "Sphere");
static_assert(sizeof(RtTriangle) == 64 && offsetof(RtTriangle, v1) == 16 && offsetof(RtTriangle, v2) == 32 &&
                  offsetof(RtTriangle, normal) == 48 && offsetof(RtTriangle, mat) == 60, "Triangle");
static_assert(sizeof(RtVertex) == 16, "Vertex");
static_assert(sizeof(RtMaterial) == 80 && offsetof(RtMaterial, subsurface) == 16 && offsetof(RtMaterial, specular_tint) == 32 &&
                  offsetof(RtMaterial, emission_color) == 48 && offsetof(RtMaterial, roughness) == 64 &&
                  offsetof(RtMaterial, metallic) == 68 && offsetof(RtMaterial, ior) == 72 && offsetof(RtMaterial, trs_weight) == 76,
              "Material");
static_assert(sizeof(RtLight) == 48 && offsetof(RtLight, color_str) == 16 && offsetof(RtLight, type) == 32 &&
                  offsetof(RtLight, accumulated_str) == 36, "Light");
static_assert(sizeof(RtMeshInfo) == 12 && offsetof(RtMeshInfo, index_end) == 4 && offsetof(RtMeshInfo, material) == 8, "MeshInfo");
static_assert(sizeof(RtCamera) == 336 && offsetof(RtCamera, viewInv) == 64 && offsetof(RtCamera, proj) == 128 &&
                  offsetof(RtCamera, projInv) == 192 && offsetof(RtCamera, viewproj) == 256 && offsetof(RtCamera, position) == 320 &&
                  offsetof(RtCamera, tanHalfFOV) == 332, "Camera");
static_assert(sizeof(RtPushConstants) == 44 && offsetof(RtPushConstants, frameCount) == 4 && offsetof(RtPushConstants, total_lights) == 8 &&
                  offsetof(RtPushConstants, lights_strength_sum) == 12 && offsetof(RtPushConstants, world_up) == 16 &&
                  offsetof(RtPushConstants, reset_frame_accumulation) == 28 && offsetof(RtPushConstants, total_spheres) == 32 &&
                  offsetof(RtPushConstants, total_triangles) == 36 && offsetof(RtPushConstants, total_meshes) == 40, "PushConstants");
