#!/usr/bin/env python3
"""A moving camera's frames (every frame resets the accumulation, so no frame depends on the one before) rendered by
ONE context, frame after frame, against TWO contexts on the same GPU taking the frames in turn: the second context's
kernel moves into the CUs the first one's drain leaves free.  Checks that both ways give the same images and prints the
time per frame.   python tools/two_contexts.py c5 [frames] [noread: only the last images are read back]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, make_params, look_camera
from raytracer_amd.workloads import WORKLOADS
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 24
READ = not (len(sys.argv) > 3 and sys.argv[3] == "noread")
preset, FW, FH, spp, mb, _ = WORKLOADS[name]
s = Scene(preset)
params = make_params(spp, mb, capi.RT_ACCEL_BVH, 0, 1, 0, capi.RT_FLAG_POOLED)


def camera(f):  # a slow turn about the canonical pose of the workload
    a = 0.004 * f
    if name == "c5":
        return look_camera((2.75 + 0.02 * f, 2.6, -1.0), (math.sin(a), -0.42, math.cos(a)), (0.0, 1.0, 0.0), 50.0, FW, FH)
    return look_camera((2.75 + 0.01 * f, 2.75, -6.0), (math.sin(a), 0.0, math.cos(a)), (0.0, 1.0, 0.0), 50.0, FW, FH)


def run(n_ctx):
    ctxs = [Context(0) for _ in range(n_ctx)]
    for c in ctxs:
        c.upload_scene(s); c.resize(FW, FH)
    images, t0 = [], 0.0
    for f in range(-4, N):  # four frames to settle (the library's choice of kernel, the tile order)
        if f == 0:
            for c in ctxs: c.sync()
            t0 = time.perf_counter()
        c = ctxs[f % n_ctx]
        if f >= n_ctx and READ:  # the image this context rendered n_ctx frames ago is read before its buffer is reused
            images.append(c.read_bgra8().copy())
        c.render_frame(camera(max(f, 0)), s.push_constants(0.0, max(f, 0), True), params)
    for c in ctxs: c.sync()
    ms = (time.perf_counter() - t0) * 1e3 / N
    for k in range(N - n_ctx, N):
        images.append(ctxs[k % n_ctx].read_bgra8().copy())
    for c in ctxs: c.close()
    return ms, images[-(N if READ else 1):]


one_ms, one = run(1)
two_ms, two = run(2)
same = len(one) == len(two) and all(np.array_equal(a, b) for a, b in zip(one, two))
print(f"{name} {FW}x{FH}, {N} frames of a moving camera, {'every image read back' if READ else 'no read-back'}: one context {one_ms:.3f} ms per frame, "
      f"two contexts in turn {two_ms:.3f} ms per frame; images identical: {same}")
sys.exit(0 if same else 1)
