// glb_loader.h — LoadModel, same contract as the reference's src/tinygltf/loader.hpp:6-8:
// fills positions and 32-bit indices of mesh 0 / primitive 0 of a binary glTF file.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/rt_abi.h"

namespace rt {

bool LoadModel(const std::string& filename, std::vector<RtVertex>& vertices, std::vector<uint32_t>& indices,
               std::string* err = nullptr);

// The general reader: walks the default scene's nodes (matrix or translation/rotation/scale,
// parents applied), appends every triangle primitive of every referenced mesh with its positions
// in world space, honours bufferView.byteStride, accepts 8-, 16- and 32-bit or absent indices,
// and rebases each primitive's indices onto the concatenated vertex list.  What the reference's
// loader leaves out (loader.cpp:37-43, 76-91).
bool LoadModelFull(const std::string& filename, std::vector<RtVertex>& vertices, std::vector<uint32_t>& indices,
                   std::string* err = nullptr);

}  // namespace rt
