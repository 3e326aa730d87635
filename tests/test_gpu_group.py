"""Several GPUs behind the C ABI, on the one GPU a test box has: the row-strip exchange kernels
(rt_strip_pack / rt_strip_unpack) reassemble a frame rendered share by share into the bits of the
whole frame, their byte layout is the one raytracer_amd.parallel restates in torch, and a group of one
device (rt_group_*, RCCL communicator included) equals rt_render_frame bit for bit in both modes."""
import numpy as np
import pytest

from helpers import bit_equal_pixels
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Group, Scene, default_camera, make_params

pytestmark = pytest.mark.gpu


def _frames(ctx, scene, W, H, frames, strip=(0, 1, 0)):
    for f in range(frames):
        ctx.render_frame(default_camera(W, H), scene.push_constants(0.0, f, f == 0), make_params(2, 4, capi.RT_ACCEL_BVH, *strip))


def test_strip_pack_and_unpack_rebuild_the_whole_frame():
    import torch
    from raytracer_amd import parallel
    scene = Scene(capi.RT_SCENE_CORNELL_TEAPOT)
    ctx = Context(0)
    try:
        W, H, rows, world = 130, 37, 4, 3
        ctx.upload_scene(scene)
        ctx.resize(W, H)
        _frames(ctx, scene, W, H, 2)
        whole_acc, whole_cnt = ctx.read_accum()
        whole_bgra = ctx.read_bgra8()
        per = capi.load().rt_strip_staging_bytes(W, H, rows, world)
        assert per == parallel.staging_bytes(H, W, rows, world) and per % 16 == 0
        gathered = torch.zeros(per * world, dtype=torch.uint8, device="cuda")
        for r in range(world):
            ctx.resize(W, H)  # zeroed: only this share's rows get written
            _frames(ctx, scene, W, H, 2, strip=(rows, world, r))
            ctx.strip_pack(rows, world, r, gathered.data_ptr() + r * per)
            ctx.sync()
            acc, cnt = ctx.read_accum()
            mine = torch.zeros(per, dtype=torch.uint8)
            parallel.pack_rows_torch(torch.from_numpy(acc), torch.from_numpy(cnt), rows, world, r, mine)
            assert torch.equal(gathered[r * per:(r + 1) * per].cpu(), mine), f"packed bytes of share {r}"
        ctx.resize(W, H)
        ctx.strip_unpack(rows, world, -1, gathered.data_ptr())
        acc, cnt = ctx.read_accum()
        assert bit_equal_pixels(acc, whole_acc).all() and np.array_equal(cnt, whole_cnt)
        assert np.array_equal(ctx.read_bgra8(), whole_bgra)
        # skipping a share leaves its rows alone
        ctx.resize(W, H)
        ctx.strip_unpack(rows, world, 1, gathered.data_ptr())
        _, cnt = ctx.read_accum()
        skipped = [y for y in range(H) if (y // rows) % world == 1]
        assert (cnt[skipped] == 0).all() and (np.delete(cnt, skipped, axis=0) == 2).all()
    finally:
        ctx.close()


def test_group_of_one_device_equals_render_frame():
    scene = Scene(capi.RT_SCENE_CORNELL_STAR)
    W, H = 200, 117
    ctx = Context(0)
    try:
        ctx.upload_scene(scene)
        ctx.resize(W, H)
        _frames(ctx, scene, W, H, 3)
        want_acc, want_cnt = ctx.read_accum()
        want_bgra = ctx.read_bgra8()
    finally:
        ctx.close()
    g = Group([0])
    try:
        assert len(g) == 1
        g.upload_scene(scene)
        g.resize(W, H)
        c0 = g.context(0)
        for mode in (capi.RT_GROUP_STRIPS, capi.RT_GROUP_SAMPLES):
            for f in range(3):
                g.render_frame(default_camera(W, H), scene.push_constants(0.0, f, f == 0), make_params(2, 4, capi.RT_ACCEL_BVH), mode, 9)
            if mode == capi.RT_GROUP_SAMPLES:
                g.reduce_samples()
                acc, cnt = g.read_total()
                assert bit_equal_pixels(acc, want_acc).all() and np.array_equal(cnt, want_cnt)
            g.sync()
            acc, cnt = c0.read_accum()
            assert bit_equal_pixels(acc, want_acc).all() and np.array_equal(cnt, want_cnt), mode
            assert np.array_equal(c0.read_bgra8(), want_bgra), mode
    finally:
        g.close()


@pytest.mark.parametrize("mode", [capi.RT_GROUP_SAMPLES, capi.RT_GROUP_STRIPS])
def test_group_batch_of_frames_equals_frame_by_frame(mode):
    """rt_group_render_frames: the frames of a batch in one call per device (fused into one launch at this size, one
    exchange for the batch) against the same frames one rt_group_render_frame call each."""
    scene = Scene(capi.RT_SCENE_CORNELL_TEAPOT)
    W, H, F = 512, 288, 5
    cam, params = default_camera(W, H), make_params(2, 4, capi.RT_ACCEL_BVH, 0, 1, 0, capi.RT_FLAG_POOLED)
    g = Group([0])
    try:
        g.upload_scene(scene)

        def result():
            if mode == capi.RT_GROUP_SAMPLES:
                g.reduce_samples()
                return g.read_total()
            g.sync()
            return g.context(0).read_accum()

        g.resize(W, H)
        for f in range(F):
            g.render_frame(cam, scene.push_constants(0.0, 10 + f * len(g), f == 0), params, mode, 9)
        want_acc, want_cnt = result()
        want_acc, want_cnt = want_acc.copy(), want_cnt.copy()
        g.resize(W, H)
        g.render_frames(cam, scene.push_constants(0.0, 10, True), params, mode, 9, F)
        assert g.context(0).stats().frames == F
        acc, cnt = result()
        assert bit_equal_pixels(acc, want_acc).all() and np.array_equal(cnt, want_cnt) and (cnt == F).all()
    finally:
        g.close()


def _one_context(scene, W, H, frame_counts, params):
    ctx = Context(0)
    try:
        ctx.upload_scene(scene)
        ctx.resize(W, H)
        for k, f in enumerate(frame_counts):
            ctx.render_frame(default_camera(W, H), scene.push_constants(0.0, f, k == 0), params)
        acc, cnt = ctx.read_accum()
        return acc.copy(), cnt.copy(), ctx.read_bgra8().copy()
    finally:
        ctx.close()


@pytest.mark.parametrize("n", [2, 3, 4])
def test_group_of_several_contexts_on_one_device(n):
    """The n-device code of RtGroup - per-context strips, staging offsets, the skipped own share, the exchange stream
    beside the next call's kernels, the deferred unpack, the samples' reduction - with n contexts on the one GPU a
    test box has (contexts that share a device exchange through device copies instead of RCCL, which wants a GPU
    per rank; everything else is the same code).  Strips: every context ends up with the whole frame, bit for bit
    what one context renders; frame by frame and in a batch.  Samples: the sum over the contexts equals sequential
    accumulation within 4e-7 x frames (the order of the float sums differs), counts exactly."""
    scene = Scene(capi.RT_SCENE_CORNELL_TEAPOT)
    W, H, rows = 200, 117, 9
    params = make_params(2, 4, capi.RT_ACCEL_BVH)
    cam = default_camera(W, H)
    want = _one_context(scene, W, H, range(5), params)
    g = Group([0] * n)
    try:
        assert len(g) == n
        g.upload_scene(scene)
        g.resize(W, H)
        # strips, one call per frame (each call's exchange runs beside the next call's kernels), then a batch on top
        for f in range(3):
            g.render_frame(cam, scene.push_constants(0.0, f, f == 0), params, capi.RT_GROUP_STRIPS, rows)
        g.render_frames(cam, scene.push_constants(0.0, 3, False), params, capi.RT_GROUP_STRIPS, rows, 2)
        g.sync()
        for i in range(n):
            c = g.context(i)
            acc, cnt = c.read_accum()
            assert bit_equal_pixels(acc, want[0]).all() and np.array_equal(cnt, want[1]) and (cnt == 5).all(), f"context {i} of {n}"
            assert np.array_equal(c.read_bgra8(), want[2]), f"context {i} of {n}"
        # another partition of the same frames on the same group: strips of 4 rows, a reset on the first frame
        g.render_frames(cam, scene.push_constants(0.0, 0, True), params, capi.RT_GROUP_STRIPS, 4, 5)
        g.sync()
        acc, cnt = g.context(n - 1).read_accum()
        assert bit_equal_pixels(acc, want[0]).all() and np.array_equal(cnt, want[1])
        # samples: 2 n frames, context i renders frames i and i + n; the reduction lands on context 0
        frames = 2 * n
        seq = _one_context(scene, W, H, range(frames), params)
        g.render_frames(cam, scene.push_constants(0.0, 0, True), params, capi.RT_GROUP_SAMPLES, 0, 2)
        g.reduce_samples()
        acc, cnt = g.read_total()
        assert np.array_equal(cnt, seq[1]) and (cnt == frames).all()
        assert np.abs(acc - seq[0]).max() <= 4e-7 * frames * max(1.0, float(np.abs(seq[0]).max()))
        g.sync()
        packed = np.abs(g.context(0).read_bgra8().astype(np.int32) - seq[2].astype(np.int32)).max()
        assert packed <= 1  # the image of the sum: a rounding boundary may fall the other way
        own, own_cnt = g.context(1).read_accum()  # the contexts' own accumulators are left as they are
        assert (own_cnt == 2).all() and not bit_equal_pixels(own, acc).all()
    finally:
        g.close()
