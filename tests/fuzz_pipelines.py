#!/usr/bin/env python3
"""Random small configurations (size, samples, bounces, strips, skybox, scene, frames) rendered by the
single kernel on the binary tree (the reference run), and by the single kernel, the pooled kernel (128-byte and
80-byte nodes) and the library's own choice on the four-wide tree; all outputs must be bit-identical, and the flat-search run must
match the CPU oracle.  Scenes: the presets and random heaps of spheres and triangles with a light or two.
    python tests/fuzz_pipelines.py [cases] [seed]      (a checker, not collected by pytest: it runs as long as it is told to)"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))  # helpers.py lives beside this file
import numpy as np
from helpers import DEBUG_FLAGS, bit_equal_pixels, gpu_frames, oracle_frames
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = Context(0)
scenes = {"star": Scene(capi.RT_SCENE_CORNELL_STAR), "teapot": Scene(capi.RT_SCENE_CORNELL_TEAPOT), "preset1": Scene(capi.RT_SCENE_PRESET1),
          "empty": Scene(capi.RT_SCENE_EMPTY)}
def random_scene(r):
    s = Scene(capi.RT_SCENE_EMPTY)
    mats = [s.add_material((r.random(), r.random(), r.random(), r.choice([1, 1, 0.6])), (0, 0, 0, 0), (1, 1, 1, 0.5), (0, 0, 0, 0),
                           r.random(), r.choice([0.0, 0.0, 0.8]), 1.5, r.choice([0.0, 0.0, 1.0])) for _ in range(3)]
    for _ in range(r.choice([1, 2, 5, 17, 60, 300])):
        c = (r.uniform(-4, 4), r.uniform(-2, 3), r.uniform(-9, -2))
        if r.random() < 0.4:
            s.add_sphere(c, r.uniform(0.1, 0.9), r.choice(mats))
        else:
            d = lambda: tuple(ci + r.uniform(-1, 1) for ci in c)
            s.add_triangle(d(), d(), d(), r.choice(mats))
    s.add_quad((-12, -2.5, -14), (-12, -2.5, 4), (12, -2.5, 4), (12, -2.5, -14), mats[0])
    s.add_light((0.3, -0.8, -0.5, 0.0), (1, 1, 1, 3.0), capi.RT_LIGHT_DIRECTIONAL)
    if r.random() < 0.5:
        s.add_light((0, 0, 0, 0), (0.3, 0.4, 0.5, 1.0), capi.RT_LIGHT_AMBIENT)
    return s


bad = 0
for k in range(cases):
    scenes["random"] = random_scene(rng)
    name = rng.choice(list(scenes))
    s = scenes[name]
    W, H = rng.choice([1, 2, 7, 8, 9, 31, 64, 65, 130, 200, 333, 640, 801]), rng.choice([1, 3, 8, 17, 40, 72, 129, 360, 451])
    spp, mb, frames = rng.choice([1, 1, 2, 3, 5]), rng.choice([0, 1, 2, 4, 9, 20]), rng.choice([1, 2, 3])
    sky = rng.choice([0, 0, 1, 2, 3, 4])
    sc = rng.choice([1, 1, 2, 3])
    strip = (0, 1, 0) if sc == 1 else (rng.choice([1, 2, 5, 9]), sc, rng.randrange(sc))
    kw = dict(spp=spp, max_bounces=mb, frames=frames, strip=strip, skybox=sky)
    ctx.set_option(capi.RT_OPT_BVH_WIDTH, 2)
    ref = gpu_frames(ctx, s, W, H, accel=capi.RT_ACCEL_BVH, flags=DEBUG_FLAGS | capi.RT_FLAG_SINGLE_KERNEL, **kw)
    ctx.set_option(capi.RT_OPT_BVH_WIDTH, 0)
    ok = ref["stats"].bvh_width == 2 or ref["total_rays"] == 0  # (a rank that owns no rows launches nothing)
    for label, fl in (("single, wide", capi.RT_FLAG_SINGLE_KERNEL), ("pooled", capi.RT_FLAG_POOLED), ("auto", 0),
                      ("pooled, image order", capi.RT_FLAG_POOLED | capi.RT_FLAG_FIXED_TILE_ORDER), ("pooled, 80-byte nodes", capi.RT_FLAG_POOLED)):
        ctx.set_option(capi.RT_OPT_BVH_NODE_FORMAT, 16 if "80-byte" in label else 0)
        g = gpu_frames(ctx, s, W, H, accel=capi.RT_ACCEL_BVH, flags=DEBUG_FLAGS | fl, **kw)
        ctx.set_option(capi.RT_OPT_BVH_NODE_FORMAT, 0)
        if "80-byte" in label and g["total_rays"] != 0 and g["stats"].bvh_node_bytes != 80:
            ok = False
            print("  the 80-byte nodes were not used")
        same = (bit_equal_pixels(g["accum"], ref["accum"]).all() and np.array_equal(g["counts"], ref["counts"]) and
                np.array_equal(g["bgra"], ref["bgra"]) and np.array_equal(g["path_hash"], ref["path_hash"]) and
                g["total_rays"] == ref["total_rays"])
        if not same or (g["stats"].bvh_width != 4 and g["total_rays"] != 0):
            ok = False
            print(f"  MISMATCH {label} (width {g['stats'].bvh_width})")
    if frames > 1:
        # the same frames in one rt_render_frames call (one launch where the image is large enough), without the debug outputs
        from raytracer_amd.renderer import default_camera, make_params
        cam = default_camera(W, H)
        p = make_params(spp, mb, capi.RT_ACCEL_BVH, strip[0], strip[1], strip[2], capi.RT_FLAG_POOLED, sky)
        ctx.upload_scene(s); ctx.resize(W, H)
        ctx.render_frames(cam, [s.push_constants(0.0, f, f == 0) for f in range(frames)], p)
        acc, cnt = ctx.read_accum()
        if not (bit_equal_pixels(acc, ref["accum"]).all() and np.array_equal(cnt, ref["counts"]) and np.array_equal(ctx.read_bgra8(), ref["bgra"])):
            ok = False
            print(f"  MISMATCH rt_render_frames ({ctx.stats().frames} frames in the launch)")
    if W * H <= 8000 and name != "teapot":
        o = oracle_frames(s, W, H, spp, mb, frames=frames, strip=strip, skybox=sky)
        f = gpu_frames(ctx, s, W, H, accel=capi.RT_ACCEL_FLAT, **kw)
        if not (bit_equal_pixels(f["accum"], o["accum"]).all() and np.array_equal(f["path_hash"], o["path_hash"]) and
                f["total_rays"] == o["total_rays"] and np.array_equal(f["bgra"], o["bgra"])):
            ok = False
            print("  MISMATCH flat search vs oracle")
    bad += 0 if ok else 1
    print(f"case {k}: {name} {W}x{H} spp {spp} bounces {mb} frames {frames} strip {strip} sky {sky}: {'ok' if ok else 'FAILED'}", flush=True)
print("failed cases:", bad)
sys.exit(1 if bad else 0)
