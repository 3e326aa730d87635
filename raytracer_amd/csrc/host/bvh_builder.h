// bvh_builder.h — host-side construction of the device acceleration structure.
//
// The reference has no acceleration structure: hit_scene tests every ray against every
// sphere, triangle and mesh triangle (shaders/raytracer.comp:469-508).  This builder
// produces a binary BVH over ALL primitives whose traversal returns the same winner as
// those flat loops, including their exact-t tie rules:
//   * across spheres / triangles / meshes the earliest-tested primitive wins
//     (strict `closest_so_far > t`, raytracer.comp:478,489,500);
//   * inside one mesh the LAST triangle wins (`t_r > closest_so_far` rejects only larger
//     t, raytracer.comp:447).
// Both rules are folded into one integer `prio` per primitive: lower prio wins a tie.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/rt_abi.h"

namespace rt {

struct SceneArrays {
    const RtSphere* spheres = nullptr;     int32_t n_spheres = 0;
    const RtTriangle* triangles = nullptr; int32_t n_triangles = 0;
    const RtVertex* vertices = nullptr;    int32_t n_vertices = 0;
    const uint32_t* indices = nullptr;     int32_t n_indices = 0;
    const RtMeshInfo* meshes = nullptr;    int32_t n_meshes = 0;
};

// One primitive as the kernels read it: 48 bytes, three 16-byte loads.
//   triangle: a = v0, b = v1 - v0, c = v2 - v0 (the edges the shader forms at
//             raytracer.comp:368-369 / 417-418, same binary32 subtraction)
//   sphere:   a = centre, b.x = radius
struct PrimRecord {
    float a[3];
    int32_t id;    // global primitive id: spheres [0,S), triangles [S,S+T), mesh triangles S+T+index/3
    float b[3];
    int32_t prio;  // tie-break key, lower wins
    float c[3];
    int32_t mat;   // material index; bit 31 set marks a sphere record
};
static_assert(sizeof(PrimRecord) == 48, "PrimRecord must be 48 bytes");
static const int32_t kSphereFlag = (int32_t)0x80000000u;

// Binary node holding BOTH children's boxes (64 bytes, four 16-byte loads).
struct BvhNode {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t child0, child1;  // count == 0: index of an internal node; count > 0: first PrimRecord of a leaf
    int32_t count0, count1;  // primitives in the leaf (0 for an internal child)
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

struct HostBvh {
    std::vector<BvhNode> nodes;         // nodes[0] is the root
    std::vector<PrimRecord> prims;      // leaf order
    std::vector<int32_t> prim_order;    // prims[i] came from flat-order primitive prim_order[i]
    int32_t depth = 0;                  // deepest leaf (root children are at depth 1)
    float sah_cost = 0.0f;
    int32_t max_leaf = 0;
};

// Flat-order primitive records (the order the reference's loops test them in): spheres,
// triangles, then each mesh's triangles.  Used by the flat kernel and as builder input.
bool gather_primitives(const SceneArrays& s, std::vector<PrimRecord>& out, std::string* err);

static const int kMaxBvhDepth = 30;  // traversal stacks hold kMaxBvhDepth + 2 entries

// max_leaf: most primitives per leaf, 1..8 (anything else: 4).
bool build_bvh(const SceneArrays& s, HostBvh& out, std::string* err, int max_leaf = 4);

// Re-lays the nodes out in treelets of `levels` levels (breadth first inside a treelet, treelets depth
// first); levels <= 1 keeps the builder's depth-first order.  Same tree, same traversal, other addresses.
void layout_treelets(HostBvh& bvh, int levels);

// Four-wide form of the same tree for the pooled kernel: a node holds up to four children's boxes as centre and
// half extent (the slab test then needs no min / max per axis: entry = (c - o) / d - h / |d|, exit = ... + h / |d|),
// one component of all four per 16-byte load, and the children as the traversal's cursor words (an internal
// node's index; bit 31 | (count - 1) << 28 | first record for a leaf; kWideUnused for an empty place, whose
// half extent is negative: its entry lies behind its exit for every ray).  The half extents are rounded up, so a
// box contains the binary tree's box it was made from.  128 bytes: one cache line per visit, about half the
// visits of the binary tree.
struct WideNode {
    float cx[4], cy[4], cz[4];
    float hx[4], hy[4], hz[4];
    int32_t child[4];
    int32_t pad[4];
};
static_assert(sizeof(WideNode) == 128, "WideNode must be 128 bytes");
static const int32_t kWideUnused = 0x7fffffff;

// The same node in 80 bytes, for scenes whose trees do not fit the caches (five 16-byte loads per visit instead of
// seven, 5/8 of the bytes): a point of the node (binary32), the children's box centres RELATIVE to it and their half
// extents in binary16, the cursor words.  The half extents are rounded up until origin + centre -+ half extent (as real
// numbers) contain the binary tree's box, so the traversal still tests a superset of the binary traversal's leaves; a
// binary16 carries 11 significant bits of a distance that is at most the node's own extent, so the boxes grow by less
// than a thousandth of the node's size.  Subnormal binary16 values are not used (a centre that would be one is stored as
// zero, a half extent as the smallest normal number), so the result does not depend on how the device treats them.
// The device widens the halves inside its fused multiply-adds (v_fma_mix_f32): no conversion instructions.
struct WideNode16 {
    float origin[3];
    float pad;
    uint16_t c[3][4];   // centre - origin: x of the four children, y, z
    uint16_t h[3][4];   // half extent (negative for an empty place)
    int32_t child[4];
};
static_assert(sizeof(WideNode16) == 80, "WideNode16 must be 80 bytes");

struct HostWideBvh {
    std::vector<WideNode> nodes;      // nodes[0] is the root
    std::vector<WideNode16> nodes16;  // the same nodes, same indices, in the 80-byte form (empty if half_ok is false)
    bool half_ok = false;             // every relative centre and half extent fits binary16's range
    int32_t stack_need = 0;           // most entries a traversal can hold at once (every node on a path entered through all its children)
    int32_t depth = 0;
};

// Collapses the binary tree: a node's children are replaced, largest box first, by their own children
// until four are held.  Every child box is one the binary tree stores, so a traversal tests a superset
// of the leaves the binary traversal tests, in another order: the same nearest hit.
void collapse_wide(const HostBvh& bvh, HostWideBvh& out);

// Every primitive range of the binary tree's leaves appears in exactly one leaf word of the wide tree, every
// node is reachable once, and stack_need / depth are what the stored tree implies.
bool validate_wide(const HostBvh& bvh, const HostWideBvh& wide, std::string* err);

// Structural self-check: every node reachable exactly once, every primitive in exactly one
// leaf, every leaf box contains its primitives, every stored child box contains the boxes
// stored below it, depth and leaf size within the traversal kernel's limits.
bool validate_bvh(const HostBvh& bvh, std::string* err);

}  // namespace rt
