// glb_loader.cpp — binary glTF (.glb) reader with the semantics of the reference's
// LoadModel (src/tinygltf/loader.cpp:12-94), written against the glTF 2.0 container
// format directly instead of through tinygltf:
//   * mesh[0].primitive[0] only (loader.cpp:37-43);
//   * indices: u16 widened to u32, or u32 copied; any other component type is an error
//     (loader.cpp:57-68);
//   * POSITION read as tightly packed float3 from bufferView.byteOffset +
//     accessor.byteOffset, byteStride ignored (loader.cpp:76-91);
//   * node transforms, normals, UVs and materials ignored.
#include "glb_loader.h"

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
    long long as_int(long long dflt) const { return kind == Num ? (long long)num : dflt; }
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
    size_t off = 0;
    if (const JVal* o = bv->get("byteOffset")) off += (size_t)o->as_int(0);
    if (const JVal* o = acc->get("byteOffset")) off += (size_t)o->as_int(0);
    const JVal* cnt = acc->get("count");
    const JVal* ct = acc->get("componentType");
    if (!cnt || !ct) return fail(err, "accessor lacks count/componentType");
    if (off > bin.size()) return fail(err, "accessor offset beyond the BIN chunk");
    *data = bin.data() + off;
    *avail = bin.size() - off;
    *count = (size_t)cnt->as_int(0);
    *ctype = ct->as_int(0);
    return true;
}

}  // namespace

bool LoadModel(const std::string& filename, std::vector<RtVertex>& vertices, std::vector<uint32_t>& indices,
               std::string* err) {
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
    std::vector<unsigned char> bin;
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

    JVal root;
    JParser parser(json.data(), json.size());
    if (!parser.parse(root)) return fail(err, "malformed glTF JSON");

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
        if (count * 2 > avail) return fail(err, "index data beyond the BIN chunk");
        indices.resize(count);
        for (size_t i = 0; i < count; ++i) indices[i] = (uint32_t)data[2 * i] | ((uint32_t)data[2 * i + 1] << 8);
    } else if (ctype == 5125) {  // UNSIGNED_INT
        if (count * 4 > avail) return fail(err, "index data beyond the BIN chunk");
        indices.resize(count);
        for (size_t i = 0; i < count; ++i) indices[i] = rd32(data + 4 * i);
    } else {
        return fail(err, "Unsupported index component type");
    }

    const JVal* attrs = prim.get("attributes");
    const JVal* posacc = attrs ? attrs->get("POSITION") : nullptr;
    if (!posacc) return fail(err, "Primitive has no POSITION attribute");
    if (!accessor_data(root, bin, posacc->as_int(-1), &data, &count, &ctype, &avail, err)) return false;
    if (count * 12 > avail) return fail(err, "position data beyond the BIN chunk");
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
