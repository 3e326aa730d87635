"""The configurations of BASELINE.json (C1..C5), shared by bench.py, the tests and tools/."""
from __future__ import annotations

from . import capi
from .renderer import default_camera, look_camera

# name: (scene preset, width, height, spp, max_bounces, description)
WORKLOADS = {
    "c1": (capi.RT_SCENE_CORNELL_TEAPOT, 640, 360, 1, 1,
           "Cornell box + teapot.glb (6360 primitives), 640x360, 1 spp, primary + 1 bounce (the CPU-runnable case)"),
    "c2": (capi.RT_SCENE_CORNELL_STAR, 1920, 1080, 4, 20,
           "reference default scene (Cornell box + star.glb, 100 primitives), 1920x1080, 4 spp, 20 bounces"),
    "c3": (capi.RT_SCENE_CORNELL_TEAPOT, 1920, 1080, 16, 4,
           "Cornell box + teapot.glb (6360 primitives), 1920x1080, 16 spp, 4 bounces"),
    "c4": (capi.RT_SCENE_CORNELL_TEAPOT, 3840, 2160, 4, 20,
           "Cornell box + teapot.glb, 3840x2160, 64 spp as 16 frames x 4 spp, progressive accumulation"),
    "c5": (capi.RT_SCENE_TEAPOT_GRID, 1920, 1080, 8, 20,
           "13x13 teapot grid (1 068 084 primitives), 1920x1080, 8 spp, 20 bounces"),
    "preset1": (capi.RT_SCENE_PRESET1, 1920, 1080, 5, 20, "sphere scene of Scene::createPreset1, 1920x1080, 5 spp"),
}


def workload_camera(name: str, width: int, height: int):
    """Canonical pose of the reference (main.cpp:75-86) except where the scene is elsewhere."""
    if name == "c5":  # low over the teapot field, so that the deep part of the BVH fills the frame
        return look_camera((2.75, 2.6, -1.0), (0.0, -0.42, 1.0), (0.0, 1.0, 0.0), 50.0, width, height)
    if name == "preset1":  # that scene sits around z = -10; the reference camera would look away from it
        return look_camera((0.0, 0.5, 0.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), 50.0, width, height)
    return default_camera(width, height)
