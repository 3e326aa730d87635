#!/usr/bin/env python3
"""The pooled kernel under combinations of its scheduling options (slots in flight, shading waves, job size, burst
length, refill threshold): every combination must give the bits of the default schedule, frame by frame and in a
fused launch.    python tools/option_matrix.py [width height]"""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import bit_equal_pixels
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, default_camera, make_params

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (320, 200)
ctx = Context(0)
scene = Scene(capi.RT_SCENE_CORNELL_TEAPOT)
ctx.upload_scene(scene)
cam = default_camera(W, H)
p = make_params(3, 4, capi.RT_ACCEL_BVH, 0, 1, 0, capi.RT_FLAG_POOLED, 0)
OPTS = (capi.RT_OPT_POOL_SLOTS, capi.RT_OPT_POOL_SHADER_WAVES, capi.RT_OPT_POOL_JOB_MIN, capi.RT_OPT_POOL_BURST, capi.RT_OPT_REFILL_MIN)


def render(values, fused):
    for o, v in zip(OPTS, values):
        ctx.set_option(o, v)
    ctx.resize(W, H)
    pcs = [scene.push_constants(0.0, f, f == 0) for f in range(3)]
    if fused:
        ctx.render_frames(cam, pcs, p)
    else:
        for pc in pcs:
            ctx.render_frame(cam, pc, p)
    acc, cnt = ctx.read_accum()
    return acc.copy(), cnt.copy(), ctx.read_bgra8().copy()


default = (0, 0, 0, 0, 1)
ref = render(default, False)
bad = n = 0
for values in itertools.product((16, 64, 272, 896, 0), (1, 4, 8, 9), (1, 24, 64), (1, 5, 64), (1, 64)):
    for fused in (False, True):
        try:
            got = render(values, fused)
            ok = bit_equal_pixels(got[0], ref[0]).all() and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
        except Exception as e:  # an abandoned frame is reported by the next read
            ok = False
            print("  error:", e)
        n += 1
        if not ok:
            bad += 1
            print(f"MISMATCH slots {values[0]} shading waves {values[1]} job {values[2]} burst {values[3]} refill {values[4]} fused {fused}", flush=True)
    if n % 60 == 0:
        print(f"{n} combinations, {bad} bad", flush=True)
render(default, False)
print(f"combinations: {n}, failed: {bad}")
sys.exit(1 if bad else 0)
