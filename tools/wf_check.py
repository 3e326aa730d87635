#!/usr/bin/env python3
"""Staged pipeline against the single-kernel pipeline on the device: bit comparison on small
frames of several scenes, then kernel time on a benchmark workload.
python tools/wf_check.py [c3|c2|c5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, make_params
from raytracer_amd.workloads import WORKLOADS, workload_camera
from helpers import gpu_frames, bit_equal_pixels, DEBUG_FLAGS

ctx = Context(0)
ok = True
for preset, W, H, spp, mb, accel in [(capi.RT_SCENE_CORNELL_STAR, 96, 54, 3, 6, capi.RT_ACCEL_BVH),
                                      (capi.RT_SCENE_CORNELL_STAR, 33, 17, 2, 4, capi.RT_ACCEL_FLAT),
                                      (capi.RT_SCENE_CORNELL_TEAPOT, 160, 90, 2, 4, capi.RT_ACCEL_BVH),
                                      (capi.RT_SCENE_PRESET1, 128, 72, 2, 8, capi.RT_ACCEL_BVH)]:
    s = Scene(preset)
    a = gpu_frames(ctx, s, W, H, spp, mb, frames=2, accel=accel, flags=DEBUG_FLAGS | capi.RT_FLAG_SINGLE_KERNEL)
    for label, fl in (("staged", capi.RT_FLAG_WAVEFRONT), ("pooled", capi.RT_FLAG_POOLED)):
        b = gpu_frames(ctx, s, W, H, spp, mb, frames=2, accel=accel, flags=DEBUG_FLAGS | fl)
        same = bit_equal_pixels(a["accum"], b["accum"])
        good = (same.all() and np.array_equal(a["path_hash"], b["path_hash"]) and np.array_equal(a["primary_ids"], b["primary_ids"])
                and np.array_equal(a["bgra"], b["bgra"]) and np.array_equal(a["counts"], b["counts"]) and a["total_rays"] == b["total_rays"])
        ok &= bool(good)
        print(f"{label}: preset {preset} {W}x{H} spp {spp} mb {mb} accel {accel}: identical={good} pixels_same={same.mean():.6f} "
              f"rays {a['total_rays']} vs {b['total_rays']}", flush=True)
if not ok:
    sys.exit(1)

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
preset, FW, FH, spp, mb, _ = WORKLOADS[name]
s = Scene(preset); ctx.upload_scene(s); ctx.resize(FW, FH)
cam = workload_camera(name, FW, FH)
for label, flags in (("single", capi.RT_FLAG_SINGLE_KERNEL), ("staged", capi.RT_FLAG_WAVEFRONT), ("pooled", capi.RT_FLAG_POOLED)):
    ms = []
    for f in range(5):
        ctx.sync(); t0 = time.perf_counter()
        ctx.render_frame(cam, s.push_constants(0.0, f, f == 0), make_params(spp, mb, capi.RT_ACCEL_BVH, flags=flags))
        ctx.sync(); ms.append((time.perf_counter() - t0) * 1e3)
    st = ctx.stats()
    print(f"{label}: wall ms {['%.2f' % m for m in ms]} event ms {st.kernel_ms:.3f} rays {st.rays}", flush=True)
