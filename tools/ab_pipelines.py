#!/usr/bin/env python3
"""Alternate the single and the pooled pipeline on one device, same frames, and print the median
kernel time of each.  python tools/ab_pipelines.py [c3|c2|c5] [rounds] [width height]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, make_params
from raytracer_amd.workloads import WORKLOADS, workload_camera
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
preset, FW, FH, spp, mb, _ = WORKLOADS[name]
if len(sys.argv) > 4:
    FW, FH = int(sys.argv[3]), int(sys.argv[4])
s = Scene(preset); ctx = Context(0); ctx.upload_scene(s); ctx.resize(FW, FH)
cam = workload_camera(name, FW, FH)
ms = {"single": [], "pooled": []}
for r in range(rounds + 2):
    for label, fl in (("single", capi.RT_FLAG_SINGLE_KERNEL), ("pooled", capi.RT_FLAG_POOLED)):
        ctx.render_frame(cam, s.push_constants(0.0, r, r == 0), make_params(spp, mb, capi.RT_ACCEL_BVH, flags=fl))
        ctx.render_frame(cam, s.push_constants(0.0, r, False), make_params(spp, mb, capi.RT_ACCEL_BVH, flags=fl))
        ctx.sync()
        if r >= 2:
            ms[label].append(ctx.stats().kernel_ms)
for k, v in ms.items():
    print(f"{name} {FW}x{FH} {k}: median {statistics.median(v):.3f} ms  min {min(v):.3f}  max {max(v):.3f}")
