// glb_loader.cpp — binary glTF (.glb) reader with the semantics of the reference's
// LoadModel (src/tinygltf/loader.cpp:12-94), written against the glTF 2.0 container
// format directly instead of through tinygltf:
//   * mesh[0].primitive[0] only (loader.cpp:37-43);
//   * indices: u16 widened to u32, or u32 copied; any other component type is an error
//     (loader.cpp:57-68);
//   * POSITION read as tightly packed float3 from bufferView.byteOffset +
//     accessor.byteOffset, byteStride ignored (loader.cpp:76-91);
//   * node transforms, normals, UVs and materials ignored.
// LoadModelFull is the general reader next to it: the default scene's node hierarchy with its
// transforms, every mesh and triangle primitive, byteStride, 8/16/32-bit and absent indices,
// indices rebased onto the concatenated vertices.
#include "glb_loader.h"
#include "rtm.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace rt {
namespace {

// ---- a minimal JSON DOM, enough for the glTF header chunk -------------------------
struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0.0;
    bool b = false;
    std::string str;
    std::vector<JVal> arr;
    std::map<std::string, JVal> obj;

    const JVal* get(const char* key) const {
        if (kind != Obj) return nullptr;
        auto it = obj.find(key);
        return it == obj.end() ? nullptr : &it->second;
    }
    const JVal* at(size_t i) const { return (kind == Arr && i < arr.size()) ? &arr[i] : nullptr; }
    // Numbers come from an untrusted file: anything that is not a number of modest size reads as `dflt`
    // (a conversion of an out-of-range double would be undefined behaviour).
    long long as_int(long long dflt) const { return (kind == Num && num >= -9.0e15 && num <= 9.0e15) ? (long long)num : dflt; }
    // A byte offset / length / count: a number in [0, limit], else false.
    bool as_size(size_t limit, size_t& out) const {
        if (kind != Num || !(num >= 0.0) || !(num <= (double)limit) || num > 9.0e15) return false;
        out = (size_t)num;
        return true;
    }
};

class JParser {
public:
    JParser(const char* p, size_t n) : p_(p), end_(p + n) {}
    bool parse(JVal& out) { return value(out, 0); }

private:
    const char* p_;
    const char* end_;
    void ws() { while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\r' || *p_ == '\t')) ++p_; }
    bool lit(const char* s) {
        size_t n = strlen(s);
        if ((size_t)(end_ - p_) < n || memcmp(p_, s, n) != 0) return false;
        p_ += n;
        return true;
    }
    bool string(std::string& out) {
        if (p_ >= end_ || *p_ != '"') return false;
        ++p_;
        while (p_ < end_ && *p_ != '"') {
            if (*p_ == '\\') {
                if (++p_ >= end_) return false;
                char c = *p_++;
                switch (c) {
                    case 'n': out.push_back('\n'); break;
                    case 't': out.push_back('\t'); break;
                    case 'r': out.push_back('\r'); break;
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'u':  // keep the escape verbatim: names are never compared by it here
                        if (end_ - p_ < 4) return false;
                        out.append("\\u").append(p_, 4);
                        p_ += 4;
                        break;
                    default: out.push_back(c);
                }
            } else {
                out.push_back(*p_++);
            }
        }
        if (p_ >= end_) return false;
        ++p_;
        return true;
    }
    bool value(JVal& v, int depth) {
        if (depth > 64) return false;
        ws();
        if (p_ >= end_) return false;
        char c = *p_;
        if (c == '{') {
            v.kind = JVal::Obj;
            ++p_; ws();
            if (p_ < end_ && *p_ == '}') { ++p_; return true; }
            for (;;) {
                ws();
                std::string key;
                if (!string(key)) return false;
                ws();
                if (p_ >= end_ || *p_ != ':') return false;
                ++p_;
                if (!value(v.obj[key], depth + 1)) return false;
                ws();
                if (p_ < end_ && *p_ == ',') { ++p_; continue; }
                if (p_ < end_ && *p_ == '}') { ++p_; return true; }
                return false;
            }
        }
        if (c == '[') {
            v.kind = JVal::Arr;
            ++p_; ws();
            if (p_ < end_ && *p_ == ']') { ++p_; return true; }
            for (;;) {
                v.arr.emplace_back();
                if (!value(v.arr.back(), depth + 1)) return false;
                ws();
                if (p_ < end_ && *p_ == ',') { ++p_; continue; }
                if (p_ < end_ && *p_ == ']') { ++p_; return true; }
                return false;
            }
        }
        if (c == '"') { v.kind = JVal::Str; return string(v.str); }
        if (lit("true")) { v.kind = JVal::Bool; v.b = true; return true; }
        if (lit("false")) { v.kind = JVal::Bool; v.b = false; return true; }
        if (lit("null")) { v.kind = JVal::Null; return true; }
        // number
        const char* s = p_;
        while (p_ < end_ && (strchr("+-0123456789.eE", *p_) != nullptr)) ++p_;
        if (p_ == s) return false;
        v.kind = JVal::Num;
        v.num = strtod(std::string(s, p_).c_str(), nullptr);
        return true;
    }
};

uint32_t rd32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

bool fail(std::string* err, const std::string& msg) {
    if (err) *err = msg;
    return false;
}

// Resolve accessor -> (pointer into BIN chunk, count, componentType).
bool accessor_data(const JVal& root, const std::vector<unsigned char>& bin, long long acc_idx,
                   const unsigned char** data, size_t* count, long long* ctype, size_t* avail, std::string* err) {
    const JVal* accs = root.get("accessors");
    const JVal* acc = accs ? accs->at((size_t)acc_idx) : nullptr;
    if (!acc) return fail(err, "accessor index out of range");
    const JVal* bvi = acc->get("bufferView");
    const JVal* bvs = root.get("bufferViews");
    const JVal* bv = (bvi && bvs) ? bvs->at((size_t)bvi->as_int(-1)) : nullptr;
    if (!bv) return fail(err, "accessor has no bufferView");
    const JVal* bufi = bv->get("buffer");
    if (!bufi || bufi->as_int(-1) != 0) return fail(err, "only the GLB-embedded buffer 0 is supported");
    size_t off = 0, view_off = 0, acc_off = 0;
    if (const JVal* o = bv->get("byteOffset"))
        if (!o->as_size(bin.size(), view_off)) return fail(err, "bufferView offset beyond the BIN chunk");
    if (const JVal* o = acc->get("byteOffset"))
        if (!o->as_size(bin.size(), acc_off)) return fail(err, "accessor offset beyond the BIN chunk");
    off = view_off + acc_off;  // each at most bin.size(): no wrap
    const JVal* cnt = acc->get("count");
    const JVal* ct = acc->get("componentType");
    if (!cnt || !ct) return fail(err, "accessor lacks count/componentType");
    if (off > bin.size()) return fail(err, "accessor offset beyond the BIN chunk");
    *data = bin.data() + off;
    *avail = bin.size() - off;
    if (!cnt->as_size(bin.size(), *count)) return fail(err, "accessor count is not a number the BIN chunk could hold");
    *ctype = ct->as_int(0);
    return true;
}


// Reads a .glb file into its parsed JSON chunk and its BIN chunk.
bool read_glb(const std::string& filename, JVal& root, std::vector<unsigned char>& bin, std::string* err) {
    FILE* f = fopen(filename.c_str(), "rb");
    if (!f) return fail(err, "cannot open " + filename);
    std::vector<unsigned char> file;
    {
        unsigned char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
        fclose(f);
    }
    if (file.size() < 20 || rd32(&file[0]) != 0x46546C67u /* "glTF" */) return fail(err, "not a GLB file: " + filename);
    if (rd32(&file[4]) != 2) return fail(err, "unsupported GLB version");
    size_t total = rd32(&file[8]);
    if (total > file.size()) total = file.size();

    std::string json;
    size_t pos = 12;
    while (pos + 8 <= total) {
        size_t len = rd32(&file[pos]);
        uint32_t type = rd32(&file[pos + 4]);
        pos += 8;
        if (pos + len > total) return fail(err, "truncated GLB chunk");
        if (type == 0x4E4F534Au /* JSON */ && json.empty()) json.assign((const char*)&file[pos], len);
        else if (type == 0x004E4942u /* BIN */ && bin.empty()) bin.assign(file.begin() + pos, file.begin() + pos + len);
        pos += (len + 3) & ~(size_t)3;
    }
    if (json.empty()) return fail(err, "GLB has no JSON chunk");
    JParser parser(json.data(), json.size());
    if (!parser.parse(root)) return fail(err, "malformed glTF JSON");
    return true;
}

// ---- the general reader (LoadModelFull) ------------------------------------------------

// An accessor resolved to its bytes: element i starts at data + i * stride.
struct AccessorView {
    const unsigned char* data = nullptr;
    size_t count = 0, stride = 0;
    long long ctype = 0;
    std::string type;
};

size_t component_size(long long ctype) {
    switch (ctype) {
        case 5120: case 5121: return 1;
        case 5122: case 5123: return 2;
        case 5125: case 5126: return 4;
        default: return 0;
    }
}

bool accessor_view(const JVal& root, const std::vector<unsigned char>& bin, long long acc_idx, AccessorView& out, std::string* err) {
    const JVal* accs = root.get("accessors");
    const JVal* acc = (accs && acc_idx >= 0) ? accs->at((size_t)acc_idx) : nullptr;
    if (!acc) return fail(err, "accessor index out of range");
    if (acc->get("sparse")) return fail(err, "sparse accessors are not supported");
    const JVal* bvi = acc->get("bufferView");
    const JVal* bvs = root.get("bufferViews");
    const JVal* bv = (bvi && bvs && bvi->as_int(-1) >= 0) ? bvs->at((size_t)bvi->as_int(-1)) : nullptr;
    if (!bv) return fail(err, "accessor has no bufferView");
    const JVal* bufi = bv->get("buffer");
    if (!bufi || bufi->as_int(-1) != 0) return fail(err, "only the GLB-embedded buffer 0 is supported");
    const JVal* cnt = acc->get("count");
    const JVal* ct = acc->get("componentType");
    const JVal* ty = acc->get("type");
    if (!cnt || !ct || !ty || ty->kind != JVal::Str) return fail(err, "accessor lacks count/componentType/type");
    if (!cnt->as_size(bin.size(), out.count)) return fail(err, "accessor count is not a number the BIN chunk could hold");
    out.ctype = ct->as_int(0);
    out.type = ty->str;
    const size_t comps = out.type == "SCALAR" ? 1 : out.type == "VEC2" ? 2 : out.type == "VEC3" ? 3 : out.type == "VEC4" ? 4 : 0;
    const size_t csize = component_size(out.ctype);
    if (comps == 0 || csize == 0) return fail(err, "unsupported accessor type");
    const size_t elem = comps * csize;
    size_t off = 0, view_len = bin.size();
    if (const JVal* o = bv->get("byteOffset"))
        if (!o->as_size(bin.size(), off)) return fail(err, "bufferView beyond the BIN chunk");
    if (const JVal* o = bv->get("byteLength"))
        if (!o->as_size(bin.size(), view_len)) return fail(err, "bufferView beyond the BIN chunk");
    if (off > bin.size() || view_len > bin.size() - off) return fail(err, "bufferView beyond the BIN chunk");
    size_t acc_off = 0;
    if (const JVal* o = acc->get("byteOffset"))
        if (!o->as_size(bin.size(), acc_off)) return fail(err, "accessor data beyond its bufferView");
    out.stride = elem;
    if (const JVal* o = bv->get("byteStride")) {
        size_t st = 0;
        if (!o->as_size(bin.size(), st)) return fail(err, "byteStride is not a number the BIN chunk could hold");
        if (st != 0) {
            if (st < elem) return fail(err, "byteStride smaller than the element");
            out.stride = st;
        }
    }
    // (count - 1) * stride + elem <= room, compared by division so that nothing can wrap
    if (out.count != 0) {
        if (acc_off > view_len || elem > view_len - acc_off) return fail(err, "accessor data beyond its bufferView");
        if (out.count - 1 > (view_len - acc_off - elem) / out.stride) return fail(err, "accessor data beyond its bufferView");
    }
    out.data = bin.data() + off + acc_off;
    return true;
}

float rdf(const unsigned char* p) {
    const uint32_t u = rd32(p);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// Local transform of a glTF node: its `matrix` (column-major), or T * R * S from `translation`,
// `rotation` (unit quaternion x, y, z, w) and `scale`.
bool node_local_matrix(const JVal& node, rtm::mat4& m, std::string* err) {
    m = rtm::identity();
    if (const JVal* mx = node.get("matrix")) {
        if (mx->kind != JVal::Arr || mx->arr.size() != 16) return fail(err, "node matrix must have 16 numbers");
        float* a = &m.c[0].x;
        for (int i = 0; i < 16; ++i) a[i] = (float)mx->arr[(size_t)i].num;
        return true;
    }
    float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, sc[3] = {1, 1, 1};
    auto read = [&](const char* key, float* dst, size_t n) -> bool {
        const JVal* v = node.get(key);
        if (!v) return true;
        if (v->kind != JVal::Arr || v->arr.size() != n) return false;
        for (size_t i = 0; i < n; ++i) dst[i] = (float)v->arr[i].num;
        return true;
    };
    if (!read("translation", t, 3) || !read("rotation", q, 4) || !read("scale", sc, 3))
        return fail(err, "malformed node translation/rotation/scale");
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    rtm::mat4 r = rtm::identity();
    r.c[0] = {1.0f - 2.0f * (y * y + z * z), 2.0f * (x * y + z * w), 2.0f * (x * z - y * w), 0.0f};
    r.c[1] = {2.0f * (x * y - z * w), 1.0f - 2.0f * (x * x + z * z), 2.0f * (y * z + x * w), 0.0f};
    r.c[2] = {2.0f * (x * z + y * w), 2.0f * (y * z - x * w), 1.0f - 2.0f * (x * x + y * y), 0.0f};
    m = rtm::mul(rtm::mul(rtm::translate(rtm::identity(), rtm::v3(t[0], t[1], t[2])), r),
                 rtm::scale(rtm::identity(), rtm::v3(sc[0], sc[1], sc[2])));
    return true;
}

// Appends every triangle primitive of mesh `mesh_idx`, positions transformed by `world`.
bool append_mesh(const JVal& root, const std::vector<unsigned char>& bin, long long mesh_idx, const rtm::mat4& world,
                 std::vector<RtVertex>& vertices, std::vector<uint32_t>& indices, std::string* err) {
    const JVal* meshes = root.get("meshes");
    const JVal* mesh = (meshes && mesh_idx >= 0) ? meshes->at((size_t)mesh_idx) : nullptr;
    if (!mesh) return fail(err, "mesh index out of range");
    const JVal* prims = mesh->get("primitives");
    if (!prims || prims->kind != JVal::Arr) return fail(err, "mesh without primitives");
    for (const JVal& prim : prims->arr) {
        const JVal* mode = prim.get("mode");
        if (mode && mode->as_int(4) != 4) continue;  // points, lines, strips and fans carry no triangle list
        const JVal* attrs = prim.get("attributes");
        const JVal* posacc = attrs ? attrs->get("POSITION") : nullptr;
        if (!posacc) return fail(err, "Primitive has no POSITION attribute");
        AccessorView pv;
        if (!accessor_view(root, bin, posacc->as_int(-1), pv, err)) return false;
        if (pv.type != "VEC3" || pv.ctype != 5126) return fail(err, "POSITION must be float VEC3");
        if (vertices.size() + pv.count >= (size_t)0xffffffffu) return fail(err, "too many vertices");
        const uint32_t base = (uint32_t)vertices.size();
        for (size_t i = 0; i < pv.count; ++i) {
            const unsigned char* e = pv.data + i * pv.stride;
            const rtm::vec4 p = rtm::mul(world, rtm::v4(rdf(e), rdf(e + 4), rdf(e + 8), 1.0f));
            RtVertex v;
            v.pos[0] = p.x; v.pos[1] = p.y; v.pos[2] = p.z;
            v._pad = 0.0f;
            vertices.push_back(v);
        }
        const JVal* idx = prim.get("indices");
        if (!idx || idx->as_int(-1) < 0) {  // non-indexed: consecutive triples
            for (size_t i = 0; i + 3 <= pv.count; i += 3)
                for (uint32_t k = 0; k < 3; ++k) indices.push_back(base + (uint32_t)i + k);
            continue;
        }
        AccessorView iv;
        if (!accessor_view(root, bin, idx->as_int(-1), iv, err)) return false;
        if (iv.type != "SCALAR" || (iv.ctype != 5121 && iv.ctype != 5123 && iv.ctype != 5125))
            return fail(err, "Unsupported index component type");
        for (size_t i = 0; i < iv.count; ++i) {
            const unsigned char* e = iv.data + i * iv.stride;
            const uint32_t v = iv.ctype == 5121 ? (uint32_t)e[0] : iv.ctype == 5123 ? ((uint32_t)e[0] | ((uint32_t)e[1] << 8)) : rd32(e);
            if (v >= pv.count) return fail(err, "index beyond the primitive's vertices");
            indices.push_back(base + v);
        }
    }
    return true;
}

bool append_node(const JVal& root, const std::vector<unsigned char>& bin, long long node_idx, const rtm::mat4& parent, int depth,
                 std::vector<RtVertex>& vertices, std::vector<uint32_t>& indices, std::string* err) {
    if (depth > 64) return fail(err, "node hierarchy deeper than 64 (cycle?)");
    const JVal* nodes = root.get("nodes");
    const JVal* node = (nodes && node_idx >= 0) ? nodes->at((size_t)node_idx) : nullptr;
    if (!node) return fail(err, "node index out of range");
    rtm::mat4 local;
    if (!node_local_matrix(*node, local, err)) return false;
    const rtm::mat4 world = rtm::mul(parent, local);
    if (const JVal* m = node->get("mesh"))
        if (!append_mesh(root, bin, m->as_int(-1), world, vertices, indices, err)) return false;
    if (const JVal* ch = node->get("children")) {
        if (ch->kind != JVal::Arr) return fail(err, "malformed node children");
        for (const JVal& c : ch->arr)
            if (!append_node(root, bin, c.as_int(-1), world, depth + 1, vertices, indices, err)) return false;
    }
    return true;
}

}  // namespace

bool LoadModelFull(const std::string& filename, std::vector<RtVertex>& vertices, std::vector<uint32_t>& indices,
                   std::string* err) {
    JVal root;
    std::vector<unsigned char> bin;
    if (!read_glb(filename, root, bin, err)) return false;
    vertices.clear();
    indices.clear();
    const JVal* scenes = root.get("scenes");
    if (scenes && scenes->kind == JVal::Arr && !scenes->arr.empty()) {
        long long which = 0;
        if (const JVal* s = root.get("scene")) which = s->as_int(0);
        const JVal* scene = scenes->at((size_t)which);
        const JVal* roots = scene ? scene->get("nodes") : nullptr;
        if (!roots || roots->kind != JVal::Arr) return fail(err, "scene without nodes");
        for (const JVal& n : roots->arr)
            if (!append_node(root, bin, n.as_int(-1), rtm::identity(), 0, vertices, indices, err)) return false;
    } else {  // no scene graph: every mesh where it stands
        const JVal* meshes = root.get("meshes");
        if (!meshes || meshes->kind != JVal::Arr) return fail(err, "No meshes found in GLTF file");
        for (size_t m = 0; m < meshes->arr.size(); ++m)
            if (!append_mesh(root, bin, (long long)m, rtm::identity(), vertices, indices, err)) return false;
    }
    if (indices.empty()) return fail(err, "no triangles found in " + filename);
    return true;
}

bool LoadModel(const std::string& filename, std::vector<RtVertex>& vertices, std::vector<uint32_t>& indices,
               std::string* err) {
    JVal root;
    std::vector<unsigned char> bin;
    if (!read_glb(filename, root, bin, err)) return false;

    const JVal* meshes = root.get("meshes");
    if (!meshes || meshes->kind != JVal::Arr || meshes->arr.empty()) return fail(err, "No meshes found in GLTF file");
    const JVal* prims = meshes->arr[0].get("primitives");
    if (!prims || prims->kind != JVal::Arr || prims->arr.empty()) return fail(err, "No primitives found in mesh");
    const JVal& prim = prims->arr[0];
    const JVal* idx = prim.get("indices");
    if (!idx || idx->as_int(-1) < 0) return fail(err, "Primitive has no indices");

    const unsigned char* data;
    size_t count, avail;
    long long ctype;
    if (!accessor_data(root, bin, idx->as_int(-1), &data, &count, &ctype, &avail, err)) return false;
    if (ctype == 5123) {  // UNSIGNED_SHORT
        if (count > avail / 2) return fail(err, "index data beyond the BIN chunk");
        indices.resize(count);
        for (size_t i = 0; i < count; ++i) indices[i] = (uint32_t)data[2 * i] | ((uint32_t)data[2 * i + 1] << 8);
    } else if (ctype == 5125) {  // UNSIGNED_INT
        if (count > avail / 4) return fail(err, "index data beyond the BIN chunk");
        indices.resize(count);
        for (size_t i = 0; i < count; ++i) indices[i] = rd32(data + 4 * i);
    } else {
        return fail(err, "Unsupported index component type");
    }

    const JVal* attrs = prim.get("attributes");
    const JVal* posacc = attrs ? attrs->get("POSITION") : nullptr;
    if (!posacc) return fail(err, "Primitive has no POSITION attribute");
    if (!accessor_data(root, bin, posacc->as_int(-1), &data, &count, &ctype, &avail, err)) return false;
    if (count > avail / 12) return fail(err, "position data beyond the BIN chunk");
    vertices.resize(count);
    for (size_t i = 0; i < count; ++i) {
        RtVertex v;
        memcpy(v.pos, data + 12 * i, 12);
        v._pad = 0.0f;
        vertices[i] = v;
    }
    return true;
}

}  // namespace rt
