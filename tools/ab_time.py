#!/usr/bin/env python3
"""Kernel time of one workload, N frames, for the library named by RT_LIB_PATH (or the default).
Run it alternately for two builds inside ONE gpurun call so both are timed on the same device."""
import os, sys, statistics, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, default_camera, look_camera, make_params
W = {"c3": (capi.RT_SCENE_CORNELL_TEAPOT, 16, 4), "c2": (capi.RT_SCENE_CORNELL_STAR, 4, 20), "c5": (capi.RT_SCENE_TEAPOT_GRID, 8, 20)}
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
preset, spp, mb = W[name]
s = Scene(preset); ctx = Context(0); ctx.upload_scene(s); ctx.resize(1920, 1080)
cam = look_camera((3.0, 7.0, -9.0), (0.0, -0.38, 1.0), (0.0, 1.0, 0.0), 50.0, 1920, 1080) if name == "c5" else default_camera(1920, 1080)
ms = []
for f in range(n + 2):
    ctx.sync()
    t0 = time.perf_counter()
    ctx.render_frame(cam, s.push_constants(0.0, f, f == 0), make_params(spp, mb, capi.RT_ACCEL_BVH))
    ctx.sync()  # wall time per frame: independent of the RtStats layout of the build under test
    if f >= 2: ms.append((time.perf_counter() - t0) * 1e3)
st = ctx.stats()
print(f"{os.path.basename(capi.LIB_PATH):40s} {name} median {statistics.median(ms):.3f} ms  min {min(ms):.3f}  rays {st.rays}")
