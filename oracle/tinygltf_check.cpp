// tinygltf_check.cpp — TEST INFRASTRUCTURE ONLY.  Pins the product's own .glb reader
// (raytracer_amd/csrc/host/glb_loader.cpp) against the third-party parser the reference uses:
// this program is compiled against the tiny_gltf.h (+ json.hpp, stb_image*.h) that sits in the
// reference tree (src/tinygltf/), where it lies, and is never linked into the product.
//
// It reads a .glb with tinygltf::TinyGLTF::LoadBinaryFromFile and takes out what the reference's
// LoadModel takes (src/tinygltf/loader.cpp:12-94): mesh 0 / primitive 0 (37-43), its indices
// through accessor -> bufferView -> buffer offsets (50-54), 16-bit indices widened or 32-bit
// copied (57-68), and its POSITION data read as tightly packed float triples whatever the
// view's byteStride says (76-91).  Output (little endian, to the file named as 2nd argument):
//   u32 vertex_count, u32 index_count, float[3 * vertex_count], u32[index_count].
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define TINYGLTF_IMPLEMENTATION
#define STB_IMAGE_IMPLEMENTATION
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include "tiny_gltf.h"

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s model.glb out.bin\n", argv[0]);
        return 2;
    }
    tinygltf::Model model;
    tinygltf::TinyGLTF loader;
    std::string err, warn;
    if (!loader.LoadBinaryFromFile(&model, &err, &warn, argv[1])) {
        fprintf(stderr, "tinygltf: %s\n", err.c_str());
        return 1;
    }
    if (model.meshes.empty() || model.meshes[0].primitives.empty()) return 1;
    const tinygltf::Primitive& prim = model.meshes[0].primitives[0];
    if (prim.indices < 0) return 1;

    const tinygltf::Accessor& ia = model.accessors[prim.indices];
    const tinygltf::BufferView& iv = model.bufferViews[ia.bufferView];
    const unsigned char* idata = &model.buffers[iv.buffer].data[iv.byteOffset + ia.byteOffset];
    std::vector<uint32_t> indices(ia.count);
    if (ia.componentType == TINYGLTF_COMPONENT_TYPE_UNSIGNED_SHORT) {
        for (size_t i = 0; i < ia.count; ++i) {
            uint16_t v;
            memcpy(&v, idata + 2 * i, 2);
            indices[i] = v;
        }
    } else if (ia.componentType == TINYGLTF_COMPONENT_TYPE_UNSIGNED_INT) {
        memcpy(indices.data(), idata, ia.count * 4);
    } else {
        return 1;
    }

    auto pos = prim.attributes.find("POSITION");
    if (pos == prim.attributes.end()) return 1;
    const tinygltf::Accessor& pa = model.accessors[pos->second];
    const tinygltf::BufferView& pv = model.bufferViews[pa.bufferView];
    const unsigned char* pdata = &model.buffers[pv.buffer].data[pv.byteOffset + pa.byteOffset];
    std::vector<float> positions(3 * pa.count);
    memcpy(positions.data(), pdata, positions.size() * 4);

    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    const uint32_t head[2] = {(uint32_t)pa.count, (uint32_t)ia.count};
    fwrite(head, 4, 2, f);
    fwrite(positions.data(), 4, positions.size(), f);
    fwrite(indices.data(), 4, indices.size(), f);
    fclose(f);
    return 0;
}
