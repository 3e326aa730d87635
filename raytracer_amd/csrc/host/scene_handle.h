// scene_handle.h — the opaque RtScene of include/rt_abi.h: a host Scene behind a C handle.
#pragma once
#include <string>

#include "scene.h"

struct RtScene {
    rt::Scene scene;
    RtScene(int preset, const std::string& assets) : scene(preset, assets) {}
};
