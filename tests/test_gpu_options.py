"""The pooled kernel's scheduling options (paths in flight, shading waves, job size, burst length, refill threshold)
decide when a path's next stage runs, never what it computes: extreme combinations must give the bits of the default
schedule, frame by frame and in a fused launch.  (tools/option_matrix.py runs the full product.)"""
import itertools

import numpy as np
import pytest

from helpers import bit_equal_pixels
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, default_camera, make_params

pytestmark = pytest.mark.gpu

OPTS = (capi.RT_OPT_POOL_SLOTS, capi.RT_OPT_POOL_SHADER_WAVES, capi.RT_OPT_POOL_JOB_MIN, capi.RT_OPT_POOL_BURST, capi.RT_OPT_REFILL_MIN)


def test_extreme_schedules_give_the_default_schedules_bits():
    W, H = 256, 144
    ctx = Context(0)
    scene = Scene(capi.RT_SCENE_CORNELL_TEAPOT)
    ctx.upload_scene(scene)
    cam = default_camera(W, H)
    p = make_params(3, 4, capi.RT_ACCEL_BVH, 0, 1, 0, capi.RT_FLAG_POOLED, 0)

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

    ref = render((0, 0, 0, 0, 1), False)
    for values in itertools.product((16, 272, 0), (1, 8, 9), (1, 64), (1, 64), (1, 64)):
        for fused in (False, True):
            got = render(values, fused)
            assert bit_equal_pixels(got[0], ref[0]).all(), (values, fused)
            assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]), (values, fused)
