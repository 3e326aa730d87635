"""rt_render_frame_noalloc (the entry point for callers that cannot allocate: its scratch memory is a
fixed device array inside the library) against rt_render_frame: the same frames must come out bit for
bit - accumulated radiance, sample counts, BGRA8, ray count - on two calls in a row, the second on a
smaller image (so that whatever the first call left in the shared scratch array is stale), for the
pooled and the single kernel, and the call must leave the device's free memory untouched."""
import ctypes as C

import numpy as np
import pytest

from helpers import bit_equal_pixels
from raytracer_amd import capi
from raytracer_amd.renderer import Context, RtError, Scene, default_camera, make_params

pytestmark = pytest.mark.gpu


def _free_bytes():
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def _reference_frames(ctx, scene, W, H, spp, mb, flags, frames, strip=(0, 1, 0)):
    ctx.resize(W, H)
    rays = 0
    for f in range(frames):
        ctx.render_frame(default_camera(W, H), scene.push_constants(0.0, f, f == 0), make_params(spp, mb, capi.RT_ACCEL_BVH, *strip, flags))
        rays += ctx.stats().rays
    acc, cnt = ctx.read_accum()
    return acc, cnt, ctx.read_bgra8(), rays


def _noalloc_frames(ctx, fb, scene, W, H, spp, mb, flags, frames, strip=(0, 1, 0)):
    fb.resize(W, H)  # a second context only lends its (zeroed) framebuffer memory
    fb.sync()
    a, c, b = fb.device_buffers()
    rays = 0
    free0 = None
    for f in range(frames):
        ctx.render_frame_noalloc(default_camera(W, H), scene.push_constants(0.0, f, f == 0),
                                 make_params(spp, mb, capi.RT_ACCEL_BVH, *strip, flags), a, c, b, W, H)
        if free0 is None:
            ctx.sync()
            free0 = _free_bytes()  # after the first call: code objects and queues are set up by now
        rays += ctx.stats().rays
    ctx.sync()
    assert _free_bytes() == free0, "rt_render_frame_noalloc changed the device's free memory"
    acc, cnt = fb.read_accum()
    return acc, cnt, fb.read_bgra8(), rays


@pytest.mark.parametrize("flags", [0, capi.RT_FLAG_SINGLE_KERNEL, capi.RT_FLAG_POOLED | capi.RT_FLAG_FIXED_TILE_ORDER])
def test_noalloc_equals_render_frame_on_two_calls_the_second_smaller(flags):
    scene = Scene(capi.RT_SCENE_CORNELL_TEAPOT)
    ctx, fb = Context(0), Context(0)
    try:
        ctx.upload_scene(scene)
        for (W, H, strip) in ((320, 200, (0, 1, 0)), (200, 120, (0, 1, 0)), (130, 37, (3, 2, 1))):
            want = _reference_frames(ctx, scene, W, H, 2, 4, flags, frames=3, strip=strip)
            got = _noalloc_frames(ctx, fb, scene, W, H, 2, 4, flags, frames=3, strip=strip)
            assert bit_equal_pixels(got[0], want[0]).all(), (W, H)
            assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]) and got[3] == want[3]
            owned = want[1] > 0
            assert owned.any() and (want[1][owned] == 3).all()
    finally:
        ctx.close()
        fb.close()


def test_noalloc_on_a_scene_whose_wide_tree_outgrows_the_scratch_stack_area():
    """The 1.07 M-triangle scene's four-wide tree can hold 44 traversal stack entries, the scratch array has room for 31
    per lane: the no-allocation call walks the binary tree instead, rt_render_frame the wide one, and both must give
    the same frame."""
    from raytracer_amd.workloads import WORKLOADS, workload_camera
    scene = Scene(WORKLOADS["c5"][0])
    assert scene.wide_bvh_info()["stack_need"] > 31
    ctx, fb = Context(0), Context(0)
    try:
        ctx.upload_scene(scene)
        W, H = 320, 180
        cam = workload_camera("c5", W, H)
        flags = capi.RT_FLAG_POOLED
        ctx.resize(W, H)
        ctx.render_frame(cam, scene.push_constants(0.0, 0, True), make_params(2, 6, capi.RT_ACCEL_BVH, 0, 1, 0, flags))
        want_acc, want_cnt = ctx.read_accum()
        want_bgra, want_width = ctx.read_bgra8(), ctx.stats().bvh_width
        fb.resize(W, H)
        a, c, b = fb.device_buffers()
        ctx.render_frame_noalloc(cam, scene.push_constants(0.0, 0, True), make_params(2, 6, capi.RT_ACCEL_BVH, 0, 1, 0, flags), a, c, b, W, H)
        ctx.sync()
        assert want_width == 4 and ctx.stats().bvh_width == 2
        acc, cnt = fb.read_accum()
        assert bit_equal_pixels(acc, want_acc).all() and np.array_equal(cnt, want_cnt) and np.array_equal(fb.read_bgra8(), want_bgra)
    finally:
        ctx.close()
        fb.close()


def test_noalloc_limits_are_refused_with_a_clear_error():
    scene = Scene(capi.RT_SCENE_CORNELL_STAR)
    ctx, fb = Context(0), Context(0)
    try:
        ctx.upload_scene(scene)
        fb.resize(64, 64)
        a, c, b = fb.device_buffers()
        cam, pc = default_camera(64, 64), scene.push_constants(0.0, 0, True)
        with pytest.raises(RtError) as e:  # 4096 x 2200 = 9.0 M pixels = 140800 tiles: beyond the scratch array (never launched)
            ctx.render_frame_noalloc(default_camera(4096, 2200), pc, make_params(1, 1), a, c, b, 4096, 2200)
        assert e.value.code == capi.RT_ERR_TOO_LARGE and "131072" in str(e.value)
        with pytest.raises(RtError) as e:
            ctx.render_frame_noalloc(cam, pc, make_params(1, 1, flags=capi.RT_FLAG_PATH_HASH), a, c, b, 64, 64)
        assert e.value.code == capi.RT_ERR_UNSUPPORTED
        with pytest.raises(RtError) as e:
            ctx.render_frame_noalloc(cam, pc, make_params(1, 1), a + 4, c, b, 64, 64)
        assert e.value.code == capi.RT_ERR_INVALID_ARGUMENT
    finally:
        ctx.close()
        fb.close()
