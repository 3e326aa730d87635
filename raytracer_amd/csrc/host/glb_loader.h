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

}  // namespace rt
