// bvh_builder.cpp — binned-SAH binary BVH over spheres, triangles and mesh triangles.
// Large scenes are built in parallel: the top of the tree serially down to subtrees of a fixed size
// (a function of the primitive count only), the subtrees by a pool of threads, and the nodes are
// then laid out exactly as the serial depth-first build would lay them out, so the result does not
// depend on the number of threads.
#include "bvh_builder.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace rt {
namespace {

const float kInf = std::numeric_limits<float>::infinity();

struct Box {
    float lo[3] = {kInf, kInf, kInf};
    float hi[3] = {-kInf, -kInf, -kInf};
    void grow(const float p[3]) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::fmin(lo[k], p[k]);
            hi[k] = std::fmax(hi[k], p[k]);
        }
    }
    void grow(const Box& b) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::fmin(lo[k], b.lo[k]);
            hi[k] = std::fmax(hi[k], b.hi[k]);
        }
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct BuildPrim {
    Box box;
    float centroid[3];
    int32_t flat_index;
};

// Boxes are padded so that (a) a hit the intersection routine accepts a few ulps outside the exact primitive
// (rounding in Moeller-Trumbore's u, v, t) still lies inside its leaf box and (b) the traversal's slab test, which
// runs in fused arithmetic on a reciprocal good to 1 ulp, never culls a box the exact test would enter.  Both
// errors grow with the magnitude of the ray's origin and the distance walked, not with the primitive's own
// coordinates: in position units at most about 2.5e-7 x (|origin| + |box| + distance) <= 1e-6 x S for rays that
// start inside a scene whose coordinates reach S.  The pad is the same for every primitive, kPadOfScale x max(1, S):
// five times that bound (a camera further than about 4 S from the origin eats into the margin).  An earlier
// pad was 1e-4 x max(1, the primitive's own largest coordinate): up to 2e-3 on the 0.03-wide triangles of
// the 1.07 M-triangle scene, 3 % of whose node visits and 6 % of whose primitive tests were the pad's.  Measured
// with the pad at 5e-8 of the primitive's coordinate: still no pixel differs from the flat search (tools/edge_ties.py).
const float kPadOfScale = 5e-6f;
float pad_of_scale(float scale) {
    static const float k = getenv("RT_BVH_PAD") ? (float)atof(getenv("RT_BVH_PAD")) : kPadOfScale;  // the variable: tuning aid
    return k * std::fmax(1.0f, scale);
}
Box padded(const Box& b, float pad) {
    Box r = b;
    for (int k = 0; k < 3; ++k) {
        r.lo[k] = b.lo[k] - pad;
        r.hi[k] = b.hi[k] + pad;
    }
    return r;
}

Box exact_box(const PrimRecord& p) {
    Box b;
    if (p.mat & kSphereFlag) {
        float r = std::fabs(p.b[0]);
        float lo[3] = {p.a[0] - r, p.a[1] - r, p.a[2] - r};
        float hi[3] = {p.a[0] + r, p.a[1] + r, p.a[2] + r};
        b.grow(lo);
        b.grow(hi);
    } else {
        float v1[3] = {p.a[0] + p.b[0], p.a[1] + p.b[1], p.a[2] + p.b[2]};
        float v2[3] = {p.a[0] + p.c[0], p.a[1] + p.c[1], p.a[2] + p.c[2]};
        b.grow(p.a);
        b.grow(v1);
        b.grow(v2);
    }
    return b;
}
// Largest coordinate magnitude of the scene's primitives.
float scene_scale(const std::vector<PrimRecord>& prims) {
    float m = 0.0f;
    for (const PrimRecord& p : prims) {
        const Box b = exact_box(p);
        for (int k = 0; k < 3; ++k) m = std::fmax(m, std::fmax(std::fabs(b.lo[k]), std::fabs(b.hi[k])));
    }
    return m;
}

struct ChildRef {
    int32_t index;  // node or first primitive; -1: empty slot; <= -2: subtree task -2 - index (parallel build only)
    int32_t count;
    Box box;
};

struct SubtreeTask {
    int begin, end, depth;
};

class Builder {
public:
    Builder(std::vector<BuildPrim>& prims, HostBvh& out) : prims_(prims), out_(out) {}

    // Subtrees of at most this many primitives (below the root) are not built but recorded in `tasks`.
    int defer_at_most = 0;
    std::vector<SubtreeTask> tasks;

    int leaf_cap = 4;  // most primitives a leaf may hold (1..8)
    static const int kBins = 16;

    ChildRef build(int begin, int end, int depth) {
        Box box, cbox;
        for (int i = begin; i < end; ++i) {
            box.grow(prims_[i].box);
            cbox.grow(prims_[i].centroid);
        }
        int n = end - begin;
        if (depth > 0 && n <= defer_at_most && n > 1) {
            ChildRef ref;
            ref.index = -2 - (int32_t)tasks.size();
            ref.count = 0;
            ref.box = box;
            tasks.push_back({begin, end, depth});
            return ref;
        }
        out_.depth = std::max(out_.depth, depth);
        if (n <= 1) return leaf(begin, end, box);

        // Levels still available below this node; a subtree of n primitives needs at least
        // ceil(log2(n / leaf_cap)) of them.  When SAH's freedom would risk exceeding the cap,
        // fall back to splitting the object list in half.
        int remaining = kMaxBvhDepth - depth;
        int needed = 0;
        while (((long long)leaf_cap << needed) < n) ++needed;
        bool force_median = needed + 2 >= remaining;

        int axis = -1;
        int mid = -1;
        if (!force_median) {
            float best_cost = kInf;
            int best_bin = -1;
            float parent_area = box.half_area();
            for (int ax = 0; ax < 3; ++ax) {
                float lo = cbox.lo[ax], ext = cbox.hi[ax] - cbox.lo[ax];
                if (!(ext > 0.0f)) continue;
                Box bin_box[kBins];
                int bin_cnt[kBins] = {0};
                float scale = (float)kBins / ext;
                for (int i = begin; i < end; ++i) {
                    int b = std::min(kBins - 1, std::max(0, (int)((prims_[i].centroid[ax] - lo) * scale)));
                    bin_box[b].grow(prims_[i].box);
                    bin_cnt[b]++;
                }
                float right_area[kBins];
                int right_cnt[kBins];
                Box acc;
                int cnt = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    acc.grow(bin_box[b]);
                    cnt += bin_cnt[b];
                    right_area[b] = acc.half_area();
                    right_cnt[b] = cnt;
                }
                Box accl;
                int cntl = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    accl.grow(bin_box[b]);
                    cntl += bin_cnt[b];
                    if (cntl == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = accl.half_area() * (float)cntl + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_bin = b;
                        axis = ax;
                    }
                }
            }
            if (axis >= 0) {
                // SAH with traversal cost 1 and intersection cost 1 per primitive
                float split_cost = 1.0f + best_cost / std::fmax(parent_area, 1e-30f);
                if (n <= leaf_cap && (float)n <= split_cost) return leaf(begin, end, box);
                float lo = cbox.lo[axis], ext = cbox.hi[axis] - cbox.lo[axis];
                float scale = (float)kBins / ext;
                auto it = std::partition(prims_.begin() + begin, prims_.begin() + end, [&](const BuildPrim& p) {
                    int b = std::min(kBins - 1, std::max(0, (int)((p.centroid[axis] - lo) * scale)));
                    return b <= best_bin;
                });
                mid = (int)(it - prims_.begin());
            } else if (n <= leaf_cap) {
                return leaf(begin, end, box);
            }
        } else if (n <= leaf_cap) {
            return leaf(begin, end, box);
        }
        if (mid <= begin || mid >= end) {
            // median split along the widest centroid axis (or by index when all centroids coincide)
            int ax = 0;
            float best = -1.0f;
            for (int k = 0; k < 3; ++k) {
                float e = cbox.hi[k] - cbox.lo[k];
                if (e > best) { best = e; ax = k; }
            }
            mid = begin + n / 2;
            std::nth_element(prims_.begin() + begin, prims_.begin() + mid, prims_.begin() + end,
                             [ax](const BuildPrim& a, const BuildPrim& b) {
                                 if (a.centroid[ax] != b.centroid[ax]) return a.centroid[ax] < b.centroid[ax];
                                 return a.flat_index < b.flat_index;
                             });
        }
        int32_t me = (int32_t)out_.nodes.size();
        out_.nodes.emplace_back();
        ChildRef l = build(begin, mid, depth + 1);
        ChildRef r = build(mid, end, depth + 1);
        set_children(me, l, r);
        sah_ += box.half_area();
        ChildRef ref;
        ref.index = me;
        ref.count = 0;
        ref.box = box;
        return ref;
    }

    void set_children(int32_t me, const ChildRef& l, const ChildRef& r) {
        BvhNode& nd = out_.nodes[me];
        for (int k = 0; k < 3; ++k) {
            nd.lo0[k] = l.box.lo[k]; nd.hi0[k] = l.box.hi[k];
            nd.lo1[k] = r.box.lo[k]; nd.hi1[k] = r.box.hi[k];
        }
        nd.child0 = l.index; nd.count0 = l.count;
        nd.child1 = r.index; nd.count1 = r.count;
    }

    float sah_ = 0.0f;
    float leaf_sah_ = 0.0f;

private:
    ChildRef leaf(int begin, int end, const Box& box) {
        ChildRef ref;
        ref.index = begin;
        ref.count = end - begin;
        ref.box = box;
        out_.max_leaf = std::max(out_.max_leaf, end - begin);
        leaf_sah_ += box.half_area() * (float)(end - begin);
        return ref;
    }
    std::vector<BuildPrim>& prims_;
    HostBvh& out_;
};

bool fail(std::string* err, const std::string& m) {
    if (err) *err = m;
    return false;
}

struct SubtreeResult {
    HostBvh bvh;      // nodes with indices local to the subtree (its root, if internal, is node 0)
    ChildRef root;
    float sah = 0.0f, leaf_sah = 0.0f;
};

// Copies the serially built top of the tree and the subtrees into `out` in depth-first order.
int32_t emit_node(const std::vector<BvhNode>& top, int32_t t, const std::vector<SubtreeResult>& sub, std::vector<BvhNode>& out) {
    const int32_t me = (int32_t)out.size();
    out.push_back(top[t]);
    for (int c = 0; c < 2; ++c) {
        const int32_t child = c == 0 ? top[t].child0 : top[t].child1;
        const int32_t count = c == 0 ? top[t].count0 : top[t].count1;
        int32_t new_child = child, new_count = count;
        if (count == 0 && child >= 0) {
            new_child = emit_node(top, child, sub, out);
        } else if (count == 0 && child <= -2) {
            const SubtreeResult& r = sub[(size_t)(-2 - child)];
            if (r.root.count > 0) {
                new_child = r.root.index;
                new_count = r.root.count;
            } else {
                const int32_t base = (int32_t)out.size();
                for (BvhNode nd : r.bvh.nodes) {
                    if (nd.count0 == 0 && nd.child0 >= 0) nd.child0 += base;
                    if (nd.count1 == 0 && nd.child1 >= 0) nd.child1 += base;
                    out.push_back(nd);
                }
                new_child = base + r.root.index;
            }
        }
        if (c == 0) { out[me].child0 = new_child; out[me].count0 = new_count; }
        else        { out[me].child1 = new_child; out[me].count1 = new_count; }
    }
    return me;
}

}  // namespace

bool gather_primitives(const SceneArrays& s, std::vector<PrimRecord>& out, std::string* err) {
    out.clear();
    if (s.n_spheres < 0 || s.n_triangles < 0 || s.n_meshes < 0 || s.n_vertices < 0 || s.n_indices < 0)
        return fail(err, "negative element count");
    const int32_t S = s.n_spheres, T = s.n_triangles;
    for (int32_t i = 0; i < S; ++i) {
        PrimRecord p;
        memset(&p, 0, sizeof p);
        memcpy(p.a, s.spheres[i].pos, 12);
        p.b[0] = s.spheres[i].r;
        p.id = i;
        p.prio = i;
        p.mat = s.spheres[i].mat | kSphereFlag;
        out.push_back(p);
    }
    for (int32_t i = 0; i < T; ++i) {
        const RtTriangle& t = s.triangles[i];
        PrimRecord p;
        for (int k = 0; k < 3; ++k) {
            p.a[k] = t.v0[k];
            p.b[k] = t.v1[k] - t.v0[k];
            p.c[k] = t.v2[k] - t.v0[k];
        }
        p.id = S + i;
        p.prio = S + i;
        p.mat = t.mat;
        out.push_back(p);
    }
    for (int32_t m = 0; m < s.n_meshes; ++m) {
        const RtMeshInfo& mi = s.meshes[m];
        if (mi.index_end < mi.index_start || (long long)mi.index_end > (long long)s.n_indices)
            return fail(err, "mesh index range outside the index buffer");
        // hit_mesh steps i += 3 while i < index_end (raytracer.comp:412); a trailing partial
        // triple would read past index_end, so only whole triangles are taken.
        uint32_t ntri = (mi.index_end - mi.index_start) / 3;
        int32_t first_id = S + T + (int32_t)(mi.index_start / 3);
        for (uint32_t k = 0; k < ntri; ++k) {
            uint32_t i = mi.index_start + 3 * k;
            uint32_t i0 = s.indices[i], i1 = s.indices[i + 1], i2 = s.indices[i + 2];
            if (i0 >= (uint32_t)s.n_vertices || i1 >= (uint32_t)s.n_vertices || i2 >= (uint32_t)s.n_vertices)
                return fail(err, "mesh index beyond the vertex buffer");
            const float* v0 = s.vertices[i0].pos;
            const float* v1 = s.vertices[i1].pos;
            const float* v2 = s.vertices[i2].pos;
            PrimRecord p;
            for (int c = 0; c < 3; ++c) {
                p.a[c] = v0[c];
                p.b[c] = v1[c] - v0[c];
                p.c[c] = v2[c] - v0[c];
            }
            p.id = S + T + (int32_t)(i / 3);
            // last triangle of the mesh wins ties inside the mesh; earlier meshes win over later ones
            p.prio = first_id + (int32_t)(ntri - 1 - k);
            p.mat = mi.material;
            out.push_back(p);
        }
    }
    return true;
}

bool build_bvh(const SceneArrays& s, HostBvh& out, std::string* err, int max_leaf) {
    out = HostBvh();
    if (max_leaf < 1 || max_leaf > 8) max_leaf = 4;
    std::vector<PrimRecord> flat;
    if (!gather_primitives(s, flat, err)) return false;
    std::vector<BuildPrim> bp(flat.size());
    const float pad = pad_of_scale(scene_scale(flat));
    for (size_t i = 0; i < flat.size(); ++i) {
        bp[i].box = padded(exact_box(flat[i]), pad);
        for (int k = 0; k < 3; ++k) bp[i].centroid[k] = 0.5f * (bp[i].box.lo[k] + bp[i].box.hi[k]);
        bp[i].flat_index = (int32_t)i;
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(bp[i].box.lo[k]) || !std::isfinite(bp[i].box.hi[k]))
                return fail(err, "primitive with non-finite coordinates");
    }
    Builder b(bp, out);
    b.leaf_cap = max_leaf;
    ChildRef empty;
    empty.index = -1;
    empty.count = 0;  // box stays inverted: never entered
    if (flat.empty()) {
        out.nodes.emplace_back();
        b.set_children(0, empty, empty);
    } else {
        // subtrees of up to 1/64 of the scene (at least 8192 primitives) are built by other threads
        const int n_all = (int)bp.size();
        if (n_all >= 65536 && !getenv("RT_BVH_SERIAL")) b.defer_at_most = std::max(8192, n_all / 64);  // the variable: testing aid
        ChildRef root = b.build(0, n_all, 0);
        if (!b.tasks.empty()) {
            std::vector<SubtreeResult> sub(b.tasks.size());
            std::atomic<size_t> next{0};
            auto work = [&]() {
                for (size_t k = next.fetch_add(1); k < b.tasks.size(); k = next.fetch_add(1)) {
                    Builder sb(bp, sub[k].bvh);
                    sb.leaf_cap = max_leaf;
                    sub[k].root = sb.build(b.tasks[k].begin, b.tasks[k].end, b.tasks[k].depth);
                    sub[k].sah = sb.sah_;
                    sub[k].leaf_sah = sb.leaf_sah_;
                }
            };
            unsigned n_threads = std::thread::hardware_concurrency();
            n_threads = std::max(1u, std::min({n_threads, 16u, (unsigned)b.tasks.size()}));
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(work);
            work();
            for (std::thread& t : pool) t.join();
            std::vector<BvhNode> top;
            top.swap(out.nodes);
            emit_node(top, 0, sub, out.nodes);
            for (const SubtreeResult& r : sub) {
                out.depth = std::max(out.depth, r.bvh.depth);
                out.max_leaf = std::max(out.max_leaf, r.bvh.max_leaf);
                b.sah_ += r.sah;
                b.leaf_sah_ += r.leaf_sah;
            }
        }
        if (root.count > 0 || out.nodes.empty()) {
            // the whole scene fits one leaf: wrap it so that node 0 is always an internal node
            out.nodes.clear();
            out.nodes.emplace_back();
            b.set_children(0, root, empty);
            out.depth = 1;
        }
        Box rb = root.box;
        float ra = std::fmax(rb.half_area(), 1e-30f);
        out.sah_cost = (b.sah_ + b.leaf_sah_) / ra;
    }
    out.prims.resize(bp.size());
    out.prim_order.resize(bp.size());
    for (size_t i = 0; i < bp.size(); ++i) {
        out.prims[i] = flat[bp[i].flat_index];
        out.prim_order[i] = bp[i].flat_index;
    }
    return true;
}

// Re-lays the nodes out in treelets: a subtree's top `levels` levels (up to 2^levels - 1 nodes) back to
// back, breadth first, then the same for each subtree hanging below, in order.  levels <= 1 keeps the
// builder's depth-first order.  The root stays node 0; only positions and child indices change.
void layout_treelets(HostBvh& bvh, int levels) {
    if (levels <= 1 || bvh.nodes.size() < 3) return;
    const std::vector<BvhNode>& old = bvh.nodes;
    std::vector<int32_t> new_index(old.size(), -1);
    std::vector<int32_t> order;  // order[new] = old
    order.reserve(old.size());
    std::vector<int32_t> roots = {0};  // treelet roots still to emit (a stack: subtrees stay contiguous)
    std::vector<int32_t> level, next;
    while (!roots.empty()) {
        const int32_t root = roots.back();
        roots.pop_back();
        level.assign(1, root);
        std::vector<int32_t> frontier;
        for (int d = 0; d < levels && !level.empty(); ++d) {
            next.clear();
            for (int32_t n : level) {
                new_index[(size_t)n] = (int32_t)order.size();
                order.push_back(n);
                const BvhNode& nd = old[(size_t)n];
                if (nd.count0 == 0 && nd.child0 >= 0) next.push_back(nd.child0);
                if (nd.count1 == 0 && nd.child1 >= 0) next.push_back(nd.child1);
            }
            level.swap(next);
        }
        // what hangs below the treelet, first child's subtree emitted first
        for (size_t k = level.size(); k-- > 0;) roots.push_back(level[k]);
    }
    std::vector<BvhNode> out(old.size());
    for (size_t k = 0; k < order.size(); ++k) {
        BvhNode nd = old[(size_t)order[k]];
        if (nd.count0 == 0 && nd.child0 >= 0) nd.child0 = new_index[(size_t)nd.child0];
        if (nd.count1 == 0 && nd.child1 >= 0) nd.child1 = new_index[(size_t)nd.child1];
        out[k] = nd;
    }
    bvh.nodes.swap(out);
}

namespace {
struct WideRef {
    int32_t child, count;  // as in BvhNode: count > 0 a leaf, count == 0 an internal node
    float lo[3], hi[3];
    float area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};
void child_refs(const BvhNode& n, std::vector<WideRef>& out) {
    for (int side = 0; side < 2; ++side) {
        WideRef r;
        r.child = side ? n.child1 : n.child0;
        r.count = side ? n.count1 : n.count0;
        if (r.child < 0) continue;  // an unused place of a one-child root
        for (int k = 0; k < 3; ++k) {
            r.lo[k] = side ? n.lo1[k] : n.lo0[k];
            r.hi[k] = side ? n.hi1[k] : n.hi0[k];
        }
        out.push_back(r);
    }
}
}  // namespace

namespace {
// binary32 -> binary16 bits, round to nearest even (up = false) or towards +infinity (up = true), for finite values
// below binary16's largest (the caller checks); results that would be subnormal are flushed: to zero when rounding to
// nearest, to the smallest normal number when rounding up (positive values) or to zero (negative ones).
uint16_t to_half(float f, bool up) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const bool neg = sign != 0;
    if ((x & 0x7fffffffu) == 0) return (uint16_t)sign;
    const int32_t e = (int32_t)((x >> 23) & 0xffu) - 127 + 15;
    if (e <= 0) return (uint16_t)((up && !neg) ? 0x0400u : sign);
    const uint32_t m = x & 0x7fffffu;
    uint32_t h = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rest = m & 0x1fffu;
    if (up) {
        if (rest != 0 && !neg) h += 1;  // towards +inf: positive values grow, negative ones are truncated
    } else if (rest > 0x1000u || (rest == 0x1000u && (h & 1u))) {
        h += 1;
    }
    return (uint16_t)(sign | h);
}
double from_half(uint16_t h) {
    const int s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 0x3ff;
    const double v = e == 0 ? std::ldexp((double)m, -24) : std::ldexp((double)(m | 0x400), e - 25);
    return s ? -v : v;
}
const float kHalfMax = 65000.0f;

// The 80-byte form of a wide node (see WideNode16).  False if a value does not fit binary16's range.
bool make_node16(const WideNode& w, WideNode16& o) {
    std::memset(&o, 0, sizeof o);
    const float* cc[3] = {w.cx, w.cy, w.cz};
    const float* hh[3] = {w.hx, w.hy, w.hz};
    for (int a = 0; a < 3; ++a) {
        float lo = kInf, hi = -kInf;
        for (int k = 0; k < 4; ++k) {
            if (w.child[k] == kWideUnused) continue;
            lo = std::fmin(lo, cc[a][k] - hh[a][k]);
            hi = std::fmax(hi, cc[a][k] + hh[a][k]);
        }
        o.origin[a] = lo <= hi ? 0.5f * lo + 0.5f * hi : 0.0f;
        if (!std::isfinite(o.origin[a])) return false;
    }
    for (int k = 0; k < 4; ++k) {
        o.child[k] = w.child[k];
        for (int a = 0; a < 3; ++a) {
            if (w.child[k] == kWideUnused) {
                o.c[a][k] = 0;
                o.h[a][k] = 0xbc00u;  // -1: entry behind exit for every ray
                continue;
            }
            const float crel = cc[a][k] - o.origin[a];
            if (!(std::fabs(crel) < kHalfMax)) return false;
            o.c[a][k] = to_half(crel, false);
            // the box the binary32 node stands for, as real numbers, inside origin + c16 -+ h16
            const double c = (double)o.origin[a] + from_half(o.c[a][k]);
            const double blo = (double)cc[a][k] - (double)hh[a][k], bhi = (double)cc[a][k] + (double)hh[a][k];
            const double need = std::fmax(bhi - c, c - blo);
            if (!(need < (double)kHalfMax)) return false;
            uint16_t h16 = to_half((float)need, true);
            if ((h16 & 0x7c00u) == 0) h16 = 0x0400u;                   // never zero or subnormal
            while (from_half(h16) < need) h16 = (uint16_t)(h16 + 1);  // (float)need may have rounded down
            o.h[a][k] = h16;
        }
    }
    return true;
}
}  // namespace

void collapse_wide(const HostBvh& bvh, HostWideBvh& out) {
    out.nodes.clear();
    out.nodes16.clear();
    out.half_ok = false;
    out.stack_need = 0;
    out.depth = 0;
    if (bvh.nodes.empty()) return;
    // depth-first, parents before children, without recursion: (binary node, wide node that stands for it)
    struct Todo { int32_t binary, wide, depth; };
    std::vector<Todo> todo;
    std::vector<int32_t> need_above;  // per wide node: stack entries held by its ancestors when it is entered
    out.nodes.emplace_back();
    need_above.push_back(0);
    todo.push_back({0, 0, 1});
    std::vector<WideRef> refs;
    while (!todo.empty()) {
        const Todo t = todo.back();
        todo.pop_back();
        refs.clear();
        child_refs(bvh.nodes[(size_t)t.binary], refs);
        while (refs.size() < 4) {
            int best = -1;
            for (size_t k = 0; k < refs.size(); ++k)
                if (refs[k].count == 0 && (best < 0 || refs[k].area() > refs[(size_t)best].area())) best = (int)k;
            if (best < 0) break;
            const WideRef open = refs[(size_t)best];
            refs.erase(refs.begin() + best);
            child_refs(bvh.nodes[(size_t)open.child], refs);
        }
        WideNode w;
        std::memset(&w, 0, sizeof w);
        const int n = (int)refs.size();
        const int32_t held = need_above[(size_t)t.wide] + (n > 0 ? n - 1 : 0);
        if (held > out.stack_need) out.stack_need = held;
        if (t.depth > out.depth) out.depth = t.depth;
        for (int k = 0; k < 4; ++k) {
            if (k >= n) {
                w.child[k] = kWideUnused;
                w.hx[k] = w.hy[k] = w.hz[k] = -1.0f;
                continue;
            }
            const WideRef& r = refs[(size_t)k];
            float c[3], h[3];
            for (int a = 0; a < 3; ++a) {
                c[a] = 0.5f * r.lo[a] + 0.5f * r.hi[a];
                const float need = std::fmax(r.hi[a] - c[a], c[a] - r.lo[a]);
                h[a] = std::nextafter(std::nextafter(std::fmax(need, 0.0f), kInf), kInf);  // c - h <= lo and hi <= c + h as rounded sums too
                while (c[a] - h[a] > r.lo[a] || c[a] + h[a] < r.hi[a]) h[a] = std::nextafter(h[a], kInf);
            }
            w.cx[k] = c[0]; w.cy[k] = c[1]; w.cz[k] = c[2];
            w.hx[k] = h[0]; w.hy[k] = h[1]; w.hz[k] = h[2];
            if (r.count > 0) {
                w.child[k] = (int32_t)(0x80000000u | ((uint32_t)(r.count - 1) << 28) | (uint32_t)r.child);
            } else {
                w.child[k] = (int32_t)out.nodes.size();
                out.nodes.emplace_back();
                need_above.push_back(held);
                todo.push_back({r.child, w.child[k], t.depth + 1});
            }
        }
        out.nodes[(size_t)t.wide] = w;
    }
    out.nodes16.resize(out.nodes.size());
    out.half_ok = true;
    for (size_t i = 0; i < out.nodes.size() && out.half_ok; ++i) out.half_ok = make_node16(out.nodes[i], out.nodes16[i]);
    if (!out.half_ok) out.nodes16.clear();
}

bool validate_wide(const HostBvh& bvh, const HostWideBvh& wide, std::string* err) {
    if (wide.nodes.empty()) return fail(err, "no wide root");
    // the binary tree's leaves, by first record
    std::vector<int32_t> leaf_count(bvh.prims.size() + 1, 0);
    size_t n_leaves = 0;
    for (const BvhNode& n : bvh.nodes) {
        if (n.child0 >= 0 && n.count0 > 0) { leaf_count[(size_t)n.child0] = n.count0; n_leaves++; }
        if (n.child1 >= 0 && n.count1 > 0) { leaf_count[(size_t)n.child1] = n.count1; n_leaves++; }
    }
    std::vector<int> seen(wide.nodes.size(), 0);
    struct Item { int32_t node, above, depth; };
    std::vector<Item> stack = {{0, 0, 1}};
    seen[0] = 1;
    size_t leaves = 0;
    int32_t need = 0, depth = 0;
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        const WideNode& w = wide.nodes[(size_t)it.node];
        int n = 0;
        for (int k = 0; k < 4; ++k) n += w.child[k] != kWideUnused;
        const int32_t held = it.above + (n > 0 ? n - 1 : 0);
        need = held > need ? held : need;
        depth = it.depth > depth ? it.depth : depth;
        for (int k = 0; k < 4; ++k) {
            const int32_t c = w.child[k];
            if (c == kWideUnused) {
                if (!(w.hx[k] < 0.0f && w.hy[k] < 0.0f && w.hz[k] < 0.0f)) return fail(err, "empty wide child can be entered");
                continue;
            }
            if (!(w.hx[k] >= 0.0f && w.hy[k] >= 0.0f && w.hz[k] >= 0.0f)) return fail(err, "wide child box inverted");
            if (c < 0) {
                const size_t first = (size_t)(c & 0x0fffffff);
                const int32_t count = ((c >> 28) & 7) + 1;
                if (first >= bvh.prims.size() || leaf_count[first] != count) return fail(err, "wide leaf is not a leaf of the binary tree");
                leaf_count[first] = 0;  // once only
                leaves++;
            } else {
                if ((size_t)c >= wide.nodes.size() || seen[(size_t)c]) return fail(err, "wide node index out of range or reached twice");
                seen[(size_t)c] = 1;
                stack.push_back({c, held, it.depth + 1});
            }
        }
    }
    if (wide.half_ok) {
        // the 80-byte form: same children, and every child's box contains the box of the 128-byte form
        if (wide.nodes16.size() != wide.nodes.size()) return fail(err, "the two forms of the wide tree differ in size");
        for (size_t i = 0; i < wide.nodes.size(); ++i) {
            const WideNode& w = wide.nodes[i];
            const WideNode16& q = wide.nodes16[i];
            const float* cc[3] = {w.cx, w.cy, w.cz};
            const float* hh[3] = {w.hx, w.hy, w.hz};
            for (int k = 0; k < 4; ++k) {
                if (q.child[k] != w.child[k]) return fail(err, "the two forms of a wide node name different children");
                for (int a = 0; a < 3; ++a) {
                    const double c = (double)q.origin[a] + from_half(q.c[a][k]), h = from_half(q.h[a][k]);
                    if (w.child[k] == kWideUnused) {
                        if (!(h < 0.0)) return fail(err, "empty child of an 80-byte node can be entered");
                        continue;
                    }
                    if ((q.h[a][k] & 0x7c00u) == 0 || ((q.c[a][k] & 0x7fffu) != 0 && (q.c[a][k] & 0x7c00u) == 0))
                        return fail(err, "subnormal binary16 in a node");
                    if (c - h > (double)cc[a][k] - (double)hh[a][k] || c + h < (double)cc[a][k] + (double)hh[a][k])
                        return fail(err, "80-byte node's box does not contain the 128-byte node's");
                }
            }
        }
    }
    if (leaves != n_leaves) return fail(err, "wide tree lost leaves");
    for (int v : seen) if (!v) return fail(err, "unreachable wide node");
    if (need != wide.stack_need || depth != wide.depth) return fail(err, "wide stack_need / depth do not match the tree");
    return true;
}

bool validate_bvh(const HostBvh& bvh, std::string* err) {
    if (bvh.nodes.empty()) return fail(err, "no root node");
    std::vector<int> node_seen(bvh.nodes.size(), 0), prim_seen(bvh.prims.size(), 0);
    struct Item { int32_t child, count; Box box; int depth; };
    std::vector<Item> stack;
    auto child_items = [&](const BvhNode& n, int depth, Item out[2]) {
        for (int k = 0; k < 3; ++k) {
            out[0].box.lo[k] = n.lo0[k]; out[0].box.hi[k] = n.hi0[k];
            out[1].box.lo[k] = n.lo1[k]; out[1].box.hi[k] = n.hi1[k];
        }
        out[0].child = n.child0; out[0].count = n.count0; out[0].depth = depth;
        out[1].child = n.child1; out[1].count = n.count1; out[1].depth = depth;
    };
    auto inside = [](const Box& inner, const Box& outer) {
        for (int k = 0; k < 3; ++k)
            if (inner.lo[k] < outer.lo[k] || inner.hi[k] > outer.hi[k]) return false;
        return true;
    };
    const float pad = pad_of_scale(scene_scale(bvh.prims));
    Item it[2];
    node_seen[0] = 1;
    child_items(bvh.nodes[0], 1, it);
    stack.push_back(it[0]);
    stack.push_back(it[1]);
    int max_depth = 0;
    while (!stack.empty()) {
        Item cur = stack.back();
        stack.pop_back();
        if (cur.count == 0 && cur.child < 0) continue;  // empty slot
        max_depth = std::max(max_depth, cur.depth);
        if (cur.depth > kMaxBvhDepth) return fail(err, "BVH deeper than the traversal stack allows");
        if (cur.count > 0) {
            if (cur.count > 8 || cur.child < 0 || (size_t)cur.child + (size_t)cur.count > bvh.prims.size())
                return fail(err, "leaf range out of bounds");
            for (int k = 0; k < cur.count; ++k) {
                if (prim_seen[cur.child + k]++) return fail(err, "primitive referenced by two leaves");
                Box pb = padded(exact_box(bvh.prims[cur.child + k]), pad);
                if (!inside(pb, cur.box)) return fail(err, "leaf box does not contain its primitive");
            }
        } else {
            if ((size_t)cur.child >= bvh.nodes.size()) return fail(err, "child index out of bounds");
            if (node_seen[cur.child]++) return fail(err, "node reachable twice");
            child_items(bvh.nodes[cur.child], cur.depth + 1, it);
            for (int c = 0; c < 2; ++c) {
                if (it[c].count == 0 && it[c].child < 0) continue;
                if (!inside(it[c].box, cur.box)) return fail(err, "child box not inside its parent's box");
                stack.push_back(it[c]);
            }
        }
    }
    for (size_t i = 0; i < prim_seen.size(); ++i)
        if (prim_seen[i] != 1) return fail(err, "primitive missing from the leaves");
    for (size_t i = 0; i < node_seen.size(); ++i)
        if (node_seen[i] != 1) return fail(err, "unreachable node");
    std::vector<int> order_seen(bvh.prims.size(), 0);
    for (int32_t o : bvh.prim_order) {
        if (o < 0 || (size_t)o >= order_seen.size() || order_seen[o]++) return fail(err, "prim_order is not a permutation");
    }
    (void)max_depth;
    return true;
}

}  // namespace rt
