"""rt_render_frames: several frames of one camera per call must equal the same frames rendered one call each, bit
for bit - whether the library fuses them into one launch of the pooled kernel (a pixel's frames then follow each
other on one XCD, with its L2 as the point of coherence) or falls back to frame-by-frame rendering."""
import numpy as np
import pytest

from helpers import bit_equal_pixels
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, default_camera, make_params

pytestmark = pytest.mark.gpu


def _sequential(ctx, scene, cam, W, H, pcs, params):
    ctx.resize(W, H)
    rays = 0
    for pc in pcs:
        ctx.render_frame(cam, pc, params)
        rays += ctx.stats().rays
    acc, cnt = ctx.read_accum()
    return acc.copy(), cnt.copy(), ctx.read_bgra8().copy(), rays


def _same(a, b):
    return bit_equal_pixels(a[0], b[0]).all() and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    c.upload_scene(Scene(capi.RT_SCENE_CORNELL_TEAPOT))
    yield c
    c.close()


@pytest.mark.parametrize("W,H,strip,n", [(512, 288, (0, 1, 0), 3), (512, 288, (0, 1, 0), 18), (512, 288, (0, 1, 0), 35), (640, 600, (9, 2, 1), 4), (520, 291, (0, 1, 0), 2)])
def test_fused_frames_equal_one_call_each(ctx, W, H, strip, n):
    scene = Scene(capi.RT_SCENE_CORNELL_TEAPOT)
    cam = default_camera(W, H)
    params = make_params(2, 4, capi.RT_ACCEL_BVH, *strip, capi.RT_FLAG_POOLED)
    pcs = [scene.push_constants(0.0, 7 + 3 * k, k == 0) for k in range(n)]  # frame counts need not be consecutive
    want = _sequential(ctx, scene, cam, W, H, pcs, params)
    ctx.resize(W, H)
    ctx.render_frames(cam, pcs, params)
    st = ctx.stats()
    assert st.frames == (n if n <= 32 else n - 32) and st.pipeline == capi.RT_PIPELINE_POOLED  # really one launch (two for 35)
    acc, cnt = ctx.read_accum()
    got = (acc, cnt, ctx.read_bgra8())
    assert _same(got, want)
    owned = cnt > 0
    assert owned.any() and (cnt[owned] == n).all()
    if n <= 32:
        assert st.rays == want[3]
    # a second call goes on accumulating (no reset), and equals the same frames one by one
    more = [scene.push_constants(0.0, 100 + k, False) for k in range(2)]
    ctx.render_frames(cam, more, params)
    acc2, cnt2 = ctx.read_accum()
    bgra2 = ctx.read_bgra8()
    want2 = _sequential(ctx, scene, cam, W, H, pcs + more, params)
    assert _same((acc2, cnt2, bgra2), want2)


@pytest.mark.parametrize("W,H,spp,mb,n,rounds", [(512, 288, 1, 0, 16, 3), (1920, 1080, 2, 2, 4, 2), (1000, 500, 1, 1, 16, 2), (512, 288, 1, 0, 32, 2)])
def test_fused_frames_under_heavy_overlap(ctx, W, H, spp, mb, n, rounds):
    """Short pixels and many frames: frames of the same pixel are in flight at the same time all over the image, so the
    wait of a pixel's end for its block's previous frame (and the L2 as the one place the framebuffer is read
    from) is exercised at every pixel; repeated, because what goes wrong here goes wrong only sometimes."""
    scene = Scene(capi.RT_SCENE_CORNELL_TEAPOT)
    cam = default_camera(W, H)
    params = make_params(spp, mb, capi.RT_ACCEL_BVH, 0, 1, 0, capi.RT_FLAG_POOLED)
    pcs = [scene.push_constants(0.0, k, k == 0) for k in range(n)]
    want = _sequential(ctx, scene, cam, W, H, pcs, params)
    for _ in range(rounds):
        ctx.resize(W, H)
        ctx.render_frames(cam, pcs, params)
        assert ctx.stats().frames == n
        acc, cnt = ctx.read_accum()
        assert _same((acc, cnt, ctx.read_bgra8()), want)
        assert (cnt == n).all()


def test_frames_that_cannot_be_fused_are_rendered_one_by_one(ctx):
    scene = Scene(capi.RT_SCENE_CORNELL_TEAPOT)
    params = make_params(2, 3, capi.RT_ACCEL_BVH, 0, 1, 0, 0)
    for (W, H, vary_time, flags) in ((40, 30, False, 0), (512, 288, True, capi.RT_FLAG_POOLED), (512, 288, False, capi.RT_FLAG_SINGLE_KERNEL)):
        cam = default_camera(W, H)
        params.flags = flags
        pcs = [scene.push_constants(1.0 * k if vary_time else 0.0, k, k == 0) for k in range(3)]
        want = _sequential(ctx, scene, cam, W, H, pcs, params)
        ctx.resize(W, H)
        ctx.render_frames(cam, pcs, params)
        assert ctx.stats().frames == 1
        acc, cnt = ctx.read_accum()
        assert _same((acc, cnt, ctx.read_bgra8()), want)
    with pytest.raises(Exception):
        ctx.render_frames(default_camera(64, 64), [], params)


def test_sixteen_fused_frames_against_the_oracle(ctx):
    """The launch the benchmark times - sixteen frames of a still camera in one launch of the pooled kernel, library
    defaults (four shading waves compiled in, four-wide tree, tile order fed back) - directly against the CPU
    oracle's sixteen accumulated frames (flat loops, raytracer.comp:469-508), not only against frame-by-frame
    rendering: accumulated floats, counts and BGRA8 bytes bit for bit but for edge-tie pixels (at most 1e-4 of the
    image, one pixel at least), 144 image blocks so that every XCD owns several."""
    from helpers import oracle_frames
    scene = Scene(capi.RT_SCENE_CORNELL_TEAPOT)
    W, H, n = 128, 72, 16
    o = oracle_frames(scene, W, H, spp=1, max_bounces=4, frames=n, threads=16)
    cam = default_camera(W, H)
    pcs = [scene.push_constants(0.0, k, k == 0) for k in range(n)]
    for _ in range(2):  # the second launch runs on the tile order the first one measured
        ctx.resize(W, H)
        ctx.render_frames(cam, pcs, make_params(1, 4, capi.RT_ACCEL_BVH))
        st = ctx.stats()
        assert st.frames == n and st.pipeline == capi.RT_PIPELINE_POOLED and st.bvh_width == 4
        acc, cnt = ctx.read_accum()
        same = bit_equal_pixels(acc, o["accum"])
        assert (~same).sum() <= 1, f"{(~same).sum()} pixels differ from the oracle"
        assert (cnt == n).all() and np.array_equal(cnt, o["counts"])
        assert np.array_equal(ctx.read_bgra8()[same], o["bgra"][same])
        if same.all():
            assert st.rays == o["total_rays"]
