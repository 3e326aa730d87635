#!/usr/bin/env python3
"""Kernel time of one workload, N frames, for the library named by RT_LIB_PATH (or the default).
Run it alternately for two builds inside ONE gpurun call so both are timed on the same device."""
import os, sys, statistics, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, make_params
from raytracer_amd.workloads import WORKLOADS, workload_camera
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mode = sys.argv[3] if len(sys.argv) > 3 else "auto"
flags = {"pooled": capi.RT_FLAG_POOLED, "single": capi.RT_FLAG_SINGLE_KERNEL, "auto": 0}[mode]
strip = tuple(int(v) for v in os.environ.get("RT_STRIP", "0,1,0").split(","))  # rows per strip, ranks, this rank
preset, FW, FH, spp, mb, _ = WORKLOADS[name]
spp = int(os.environ.get("RT_SPP", spp)); mb = int(os.environ.get("RT_BOUNCES", mb))  # overrides, for scaling studies
FW = int(os.environ.get("RT_W", FW)); FH = int(os.environ.get("RT_H", FH))
s = Scene(preset); ctx = Context(0); ctx.upload_scene(s); ctx.resize(FW, FH)
cam = workload_camera(name, FW, FH)
ms = []
for f in range(n + 2):
    ctx.sync()
    t0 = time.perf_counter()
    ctx.render_frame(cam, s.push_constants(0.0, f, f == 0), make_params(spp, mb, capi.RT_ACCEL_BVH, strip[0], strip[1], strip[2], flags))
    ctx.sync()  # wall time per frame: independent of the RtStats layout of the build under test
    if f >= 2: ms.append((time.perf_counter() - t0) * 1e3)
st = ctx.stats()
print(f"{os.path.basename(capi.LIB_PATH):40s} {name} {mode} median {statistics.median(ms):.3f} ms  min {min(ms):.3f}  rays {st.rays}")
