#!/usr/bin/env python3
"""Frames of one camera rendered one launch each against several per launch (rt_render_frames): bit-equality of the
results and time per frame.   python tools/batch_time.py c3 [frames per launch] [launches]"""
import os, sys, statistics, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, make_params
from raytracer_amd.workloads import WORKLOADS, workload_camera
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = int(sys.argv[3]) if len(sys.argv) > 3 else 6
preset, FW, FH, spp, mb, _ = WORKLOADS[name]
FW = int(os.environ.get("RT_W", FW)); FH = int(os.environ.get("RT_H", FH))
s = Scene(preset); cam = workload_camera(name, FW, FH)
strip = tuple(int(v) for v in os.environ.get("RT_STRIP", "0,1,0").split(","))
params = make_params(spp, mb, capi.RT_ACCEL_BVH, strip[0], strip[1], strip[2], capi.RT_FLAG_POOLED)
out = {}
for mode in ("single-frame launches", f"{F} frames per launch"):
    ctx = Context(0); ctx.upload_scene(s); ctx.resize(FW, FH)
    ms = []
    for l in range(L):
        pcs = [s.push_constants(0.0, l * F + k, l == 0 and k == 0) for k in range(F)]
        ctx.sync(); t0 = time.perf_counter()
        if mode.startswith("single"):
            for pc in pcs: ctx.render_frame(cam, pc, params)
        else:
            ctx.render_frames(cam, pcs, params)
        ctx.sync(); ms.append((time.perf_counter() - t0) * 1e3 / F)
    acc, cnt = ctx.read_accum()
    out[mode] = (acc.copy(), cnt.copy(), ctx.read_bgra8().copy())
    print(f"{name} {FW}x{FH} {mode}: median {statistics.median(ms[2:]):.3f} ms per frame (min {min(ms[2:]):.3f}), rays of the last launch {ctx.stats().rays}", flush=True)
    ctx.close()
a, b = out.values()
same = np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
print("bit-identical:", same, "counts", int(a[1].min()), int(a[1].max()))
sys.exit(0 if same else 1)
