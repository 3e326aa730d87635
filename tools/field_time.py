#!/usr/bin/env python3
"""Kernel time of an n x n field of teapots (each its own mesh; 6 320 triangles apiece) at 1080p, 8 spp x 8 bounces, pooled
kernel, for the four-wide tree's two node formats: where the 80-byte nodes start to pay.
python tools/field_time.py n [frames]"""
import os, sys, statistics, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from raytracer_amd import capi
from raytracer_amd.renderer import Context, look_camera, make_params
from test_gpu_large_scene import teapot_field
n = int(sys.argv[1]); frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
s = teapot_field(n)
info = s.bvh_info()
W, H = 1920, 1080
cam = look_camera((2.75, 3.2, -3.0), (0.0, -0.45, 1.0), (0.0, 1.0, 0.0), 55.0, W, H)
ctx = Context(0); ctx.upload_scene(s); ctx.resize(W, H)
for fmt in (32, 16, 32, 16):
    ctx.set_option(capi.RT_OPT_BVH_NODE_FORMAT, fmt)
    ms = []
    for f in range(frames + 2):
        ctx.sync(); t0 = time.perf_counter()
        ctx.render_frame(cam, s.push_constants(0.0, f, f == 0), make_params(8, 8, capi.RT_ACCEL_BVH, 0, 1, 0, capi.RT_FLAG_POOLED))
        ctx.sync()
        if f >= 2: ms.append((time.perf_counter() - t0) * 1e3)
    st = ctx.stats()
    print(f"{n}x{n} teapots, {info['prims']} prims, scene {(info['prims'] * 48 + info['nodes'] * 64) / 2**20:.1f} MB, nodes of {st.bvh_node_bytes} B: median {statistics.median(ms):.3f} ms  rays {st.rays}", flush=True)
