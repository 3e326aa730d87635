"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same scene,
camera and RNG seeds.

Bar (floating point, stated here): with the reference's flat search the accumulated
radiance, sample counts, BGRA8 bytes, primary-hit ids, per-pixel hash of ALL nearest-hit
ids and the ray count must be BIT-IDENTICAL to the oracle (math_mode 0).  With the BVH the
same holds for every pixel except documented edge-tie pixels (grazing hits whose
Moeller-Trumbore result lies outside the padded leaf box); their share must stay below
1e-4 of the image and everywhere else the bits must match.
"""
import numpy as np
import pytest

from helpers import (DEBUG_FLAGS, assert_identical, assert_identical_but_edge_ties, bit_equal_pixels, gpu_frames,
                     oracle_frames)
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Controller, Scene, default_camera, look_camera, make_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def star():
    return Scene(capi.RT_SCENE_CORNELL_STAR)


@pytest.fixture(scope="module")
def teapot():
    return Scene(capi.RT_SCENE_CORNELL_TEAPOT)


def test_flat_star_bit_exact(ctx, star):
    W, H = 96, 54
    o = oracle_frames(star, W, H, spp=2, max_bounces=6)
    g = gpu_frames(ctx, star, W, H, spp=2, max_bounces=6, accel=capi.RT_ACCEL_FLAT)
    assert_identical(g, o)


def test_bvh_star(ctx, star):
    W, H = 96, 54
    o = oracle_frames(star, W, H, spp=2, max_bounces=6)
    g = gpu_frames(ctx, star, W, H, spp=2, max_bounces=6, accel=capi.RT_ACCEL_BVH)
    assert_identical_but_edge_ties(g, o)


def test_flat_teapot_bit_exact(ctx, teapot):
    W, H = 64, 36
    o = oracle_frames(teapot, W, H, spp=1, max_bounces=3)
    g = gpu_frames(ctx, teapot, W, H, spp=1, max_bounces=3, accel=capi.RT_ACCEL_FLAT)
    assert_identical(g, o)


def test_bvh_teapot(ctx, teapot):
    W, H = 64, 36
    o = oracle_frames(teapot, W, H, spp=2, max_bounces=4)
    g = gpu_frames(ctx, teapot, W, H, spp=2, max_bounces=4, accel=capi.RT_ACCEL_BVH)
    assert_identical_but_edge_ties(g, o)


def test_accumulation_over_frames(ctx, star):
    """Frame accumulation (raytracer.comp:984-992): three frames, reset on the first."""
    W, H = 64, 36
    o = oracle_frames(star, W, H, spp=1, max_bounces=4, frames=3)
    g = gpu_frames(ctx, star, W, H, spp=1, max_bounces=4, frames=3, accel=capi.RT_ACCEL_FLAT)
    assert (g["counts"] == 3).all()
    assert_identical(g, o)


def test_controller_driven_session_accumulates_and_resets(ctx, star):
    """A session driven by the camera controller (main.cpp:283-290 loop order): frames accumulate
    while the camera rests, a movement key restarts the accumulation, and every frame equals the
    oracle's rendering of the same camera block and push constants."""
    from helpers import oracle_render
    W, H = 48, 27
    ctl = Controller()
    ctx.upload_scene(star)
    ctx.resize(W, H)
    params = make_params(1, 3, capi.RT_ACCEL_FLAT)
    inputs = []
    clock = 0.0
    for keys in (0, 0, 0, capi.RT_KEY_W, 0):
        clock += 0.02
        ctl.begin_frame(clock, keys)
        cam, pc = ctl.frame_inputs(star, W, H)
        ctx.render_frame(cam, pc, params)
        ctl.end_frame(clock)
        inputs.append((cam, pc))
    acc, cnt = ctx.read_accum()
    assert [pc.reset_frame_accumulation for _, pc in inputs] == [1, 0, 0, 1, 0]
    assert [pc.frameCount for _, pc in inputs] == [0, 1, 2, 3, 4]
    assert (cnt == 2).all()   # the frame with the W key and the one after it
    o = oracle_render(star, W, H, 1, 3, inputs)
    assert np.array_equal(cnt, o["counts"])
    assert bit_equal_pixels(acc, o["accum"]).all()


def _custom_scene(kind):
    """Small scenes that reach the branches the default scenes do not."""
    s = Scene(capi.RT_SCENE_EMPTY)
    white = s.add_material((0.8, 0.8, 0.8, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (0, 0, 0, 0), 0.6, 0.0, 1.5, 0.0)
    glass = s.add_material((1, 1, 1, 1), (1, 1, 1, 1), (1, 1, 1, 0.5), (0, 0, 0, 0), 0.1, 0.0, 1.5, 1.0)
    veil = s.add_material((0.9, 0.2, 0.2, 0.5), (0, 0, 0, 0), (1, 1, 1, 0.5), (0, 0, 0, 0), 0.4, 0.3, 1.5, 0.0)
    s.add_quad((-10, 0, -10), (-10, 0, 20), (16, 0, 20), (16, 0, -10), white)
    s.add_sphere((2.0, 1.0, 2.0), 1.0, glass)
    s.add_sphere((4.2, 0.8, 1.5), 0.8, veil)
    s.add_triangle((0, 0, 4), (5, 0, 4), (2.5, 4, 4), white)
    if kind == "ambient":
        s.add_light((0, 0, 0, 0), (0.231, 0.756, 0.945, 1.0), capi.RT_LIGHT_AMBIENT)
    elif kind == "directional":
        s.add_light((0.33, -0.67, 0.67, 0.0), (1, 1, 1, 5.0), capi.RT_LIGHT_DIRECTIONAL)
    elif kind == "mixed":
        s.add_light((0.33, -0.67, 0.67, 0.0), (1, 1, 1, 5.0), capi.RT_LIGHT_DIRECTIONAL)
        s.add_light((0, 0, 0, 0), (0.2, 0.2, 0.3, 1.0), capi.RT_LIGHT_AMBIENT)
        s.add_light((1, 3, 1, 0.3), (1, 0.9, 0.8, 2.0), capi.RT_LIGHT_POINT)  # falls through the shader's switch
    elif kind == "nolights":
        pass
    return s


@pytest.mark.parametrize("kind", ["ambient", "directional", "mixed", "nolights"])
@pytest.mark.parametrize("accel", [capi.RT_ACCEL_FLAT, capi.RT_ACCEL_BVH])
def test_light_types_transparency_and_transmission(ctx, kind, accel):
    s = _custom_scene(kind)
    W, H = 72, 40
    o = oracle_frames(s, W, H, spp=2, max_bounces=5)
    g = gpu_frames(ctx, s, W, H, spp=2, max_bounces=5, accel=accel)
    if accel == capi.RT_ACCEL_FLAT:
        assert_identical(g, o)
    else:
        assert_identical_but_edge_ties(g, o)
    assert np.isfinite(g["accum"]).all()


@pytest.mark.parametrize("preset,front", [(capi.RT_SCENE_PRESET1, (0, 0, -1)), (capi.RT_SCENE_PRESET1_TEAPOT, (0, 0, 1)),
                                          (capi.RT_SCENE_PRESET1_TEAPOT, (0, -0.2, -1))])
@pytest.mark.parametrize("accel", [capi.RT_ACCEL_FLAT, capi.RT_ACCEL_BVH])
def test_preset1_from_inside_the_scene(ctx, preset, front, accel):
    """createPreset1 (sphere-heavy: mirrors, nested dielectrics, an emissive sphere and triangles) and its
    variant with the teapot, seen from the origin towards the spheres (-z) and towards the teapot (+z)."""
    s = Scene(preset)
    W, H = 96, 54
    cam = look_camera((0.0, 0.5, 0.0), front, (0, 1, 0), 60.0, W, H)
    o = oracle_frames(s, W, H, spp=2, max_bounces=6, camera=cam)
    g = gpu_frames(ctx, s, W, H, spp=2, max_bounces=6, accel=accel, camera=cam)
    if accel == capi.RT_ACCEL_FLAT:
        assert_identical(g, o)
    else:
        assert_identical_but_edge_ties(g, o)
    assert len(np.unique(g["primary_ids"])) > 3  # the camera does see the scene


@pytest.mark.parametrize("skybox", [capi.RT_SKY_DAY, capi.RT_SKY_NIGHT, capi.RT_SKY_WHITE, capi.RT_SKY_BLACK])
def test_skybox_variants(ctx, star, skybox):
    """The shader's other skybox functions (raytracer.comp:520-569) behind RtParams.skybox: bit-identical
    to the oracle with the flat search, and the same bits from the pooled pipeline."""
    from helpers import DEBUG_FLAGS
    from raytracer_amd.renderer import look_camera
    W, H = 80, 45
    cam = look_camera((2.75, 2.75, -9.0), (0.15, 0.25, 1.0), (0.0, 1.0, 0.0), 70.0, W, H)  # box and plenty of sky
    o = oracle_frames(star, W, H, spp=2, max_bounces=5, camera=cam, skybox=skybox)
    g = gpu_frames(ctx, star, W, H, spp=2, max_bounces=5, accel=capi.RT_ACCEL_FLAT, camera=cam, skybox=skybox)
    assert_identical(g, o)
    assert (o["primary_ids"] < 0).mean() > 0.2
    b = gpu_frames(ctx, star, W, H, spp=2, max_bounces=5, accel=capi.RT_ACCEL_BVH, camera=cam, skybox=skybox)
    for fl in (capi.RT_FLAG_POOLED,):
        p = gpu_frames(ctx, star, W, H, spp=2, max_bounces=5, accel=capi.RT_ACCEL_BVH, camera=cam, skybox=skybox,
                       flags=DEBUG_FLAGS | fl)
        assert_identical(p, b)


@pytest.mark.parametrize("accel", [capi.RT_ACCEL_FLAT, capi.RT_ACCEL_BVH])
def test_scene_with_two_rebased_models(ctx, accel):
    """Two models in one scene (general GLB reader, indices rebased onto each model's vertices - the
    case scene.cpp:844-845 cannot express): the mesh path of the search against the oracle."""
    s = _custom_scene("directional")
    gold = s.add_material((1.0, 0.78, 0.34, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (0, 0, 0, 0), 0.2, 1.0, 1.5, 0.0)
    fl = capi.RT_MODEL_FULL_GLTF | capi.RT_MODEL_REBASE_INDICES
    s.add_model("star.glb", (1.0, 2.5, 1.0), 0.0, 20.0, 180.0, 0.1, gold, flags=fl)
    s.add_model("teapot.glb", (4.0, 1.5, 0.5), 90.0, 30.0, 0.0, 0.3, gold, flags=fl)
    assert s.view().total_meshes == 2
    W, H = 96, 54
    o = oracle_frames(s, W, H, spp=2, max_bounces=4)
    g = gpu_frames(ctx, s, W, H, spp=2, max_bounces=4, accel=accel)
    if accel == capi.RT_ACCEL_FLAT:
        assert_identical(g, o)
    else:
        assert_identical_but_edge_ties(g, o)
    n_prims = s.bvh_info()["prims"]
    assert (g["primary_ids"] >= n_prims - 6320).any()  # the second model is seen: its ids are the last 6320


@pytest.mark.parametrize("size", [(1, 1), (7, 3), (9, 17), (130, 5)])
def test_ragged_image_sizes(ctx, star, size):
    """Sizes that are not multiples of the 8x8 pixel tiles: every pixel written exactly once."""
    W, H = size
    o = oracle_frames(star, W, H, spp=1, max_bounces=3)
    g = gpu_frames(ctx, star, W, H, spp=1, max_bounces=3, accel=capi.RT_ACCEL_FLAT)
    assert_identical(g, o)


def test_empty_scene_is_all_sky(ctx):
    s = Scene(capi.RT_SCENE_EMPTY)
    g = gpu_frames(ctx, s, 16, 9, spp=2, max_bounces=3, accel=capi.RT_ACCEL_BVH)
    o = oracle_frames(s, 16, 9, spp=2, max_bounces=3)
    assert_identical(g, o)
    assert (g["primary_ids"] == -1).all() and g["total_rays"] == 16 * 9 * 2


def test_strip_partition_on_gpu_equals_whole_frame(ctx, star):
    W, H = 64, 45
    whole = gpu_frames(ctx, star, W, H, spp=1, max_bounces=4, accel=capi.RT_ACCEL_BVH)
    acc = np.zeros_like(whole["accum"])
    rays = 0
    for r in range(3):
        part = gpu_frames(ctx, star, W, H, spp=1, max_bounces=4, accel=capi.RT_ACCEL_BVH, strip=(4, 3, r))
        rows = [y for y in range(H) if (y // 4) % 3 == r]
        other = [y for y in range(H) if (y // 4) % 3 != r]
        acc[rows] = part["accum"][rows]
        assert (part["counts"][other] == 0).all() and (part["counts"][rows] == 1).all()
        rays += part["total_rays"]
    assert bit_equal_pixels(acc, whole["accum"]).all() and rays == whole["total_rays"]


def test_bvh_equals_flat_at_a_larger_size(ctx, teapot):
    """No oracle here (too slow at this size): the two search strategies of the device must
    agree with each other except on edge-tie pixels."""
    W, H = 480, 270
    f = gpu_frames(ctx, teapot, W, H, spp=2, max_bounces=4, accel=capi.RT_ACCEL_FLAT)
    b = gpu_frames(ctx, teapot, W, H, spp=2, max_bounces=4, accel=capi.RT_ACCEL_BVH)
    bad = assert_identical_but_edge_ties(b, f)
    # the documented edge-tie list: which pixels, and what the two searches saw there
    same = bit_equal_pixels(b["accum"], f["accum"]) & (b["path_hash"] == f["path_hash"])
    ties = [{"x": int(x), "y": int(y), "primary_id_bvh": int(b["primary_ids"][y, x]), "primary_id_flat": int(f["primary_ids"][y, x])}
            for y, x in zip(*np.nonzero(~same))]
    print(f"edge-tie pixels at {W}x{H}, 2 spp, 4 bounces, teapot scene: {bad} {ties}")
    import json, os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "edge_ties_480x270.json"), "w") as fh:
            json.dump({"width": W, "height": H, "spp": 2, "max_bounces": 4, "scene": "cornell_teapot", "pixels": ties}, fh)


def test_retired_pipeline_flag_is_refused(ctx, star):
    ctx.upload_scene(star)
    ctx.resize(8, 8)
    from raytracer_amd.renderer import RtError, default_camera
    with pytest.raises(RtError) as e:
        ctx.render_frame(default_camera(8, 8), star.push_constants(0.0, 0, True), make_params(1, 1, flags=capi.RT_FLAG_RESERVED_16))
    assert e.value.code == capi.RT_ERR_UNSUPPORTED


def test_unknown_skybox_is_refused(ctx, star):
    ctx.upload_scene(star)
    ctx.resize(8, 8)
    from raytracer_amd.renderer import RtError, default_camera
    with pytest.raises(RtError) as e:
        ctx.render_frame(default_camera(8, 8), star.push_constants(0.0, 0, True), make_params(1, 1, skybox=7))
    assert e.value.code == capi.RT_ERR_INVALID_ARGUMENT


def test_error_codes(ctx, star):
    from raytracer_amd.renderer import RtError, default_camera, make_params
    c = Context(0)
    try:
        cam = default_camera(8, 8)
        pc = star.push_constants(0.0, 0, True)
        with pytest.raises(RtError) as e:
            c.render_frame(cam, pc)
        assert e.value.code == capi.RT_ERR_NO_SCENE
        c.upload_scene(star)
        with pytest.raises(RtError) as e:
            c.render_frame(cam, pc)
        assert e.value.code == capi.RT_ERR_NO_FRAMEBUFFER
        c.resize(8, 8)
        with pytest.raises(RtError) as e:
            c.render_frame(cam, pc, make_params(spp=0))
        assert e.value.code == capi.RT_ERR_INVALID_ARGUMENT
        pc.total_triangles += 1  # loop bound differs from the uploaded scene
        with pytest.raises(RtError) as e:
            c.render_frame(cam, pc)
        assert e.value.code == capi.RT_ERR_INVALID_ARGUMENT
        with pytest.raises(RtError):
            Context(99)
    finally:
        c.close()


# ---------------------------------------------------------------------------- pipelines
# The library has two schedules of the same per-path program: the single persistent kernel
# (RT_FLAG_SINGLE_KERNEL; also what small frames get by default) and the pooled one
# (RT_FLAG_POOLED: persistent workgroups regrouping their paths through LDS queues; what larger
# frames get by default).  A pixel's arithmetic never depends on the schedule, so both must agree
# bit for bit — with each other at any size, and with the oracle.

PIPELINES = {"pooled": capi.RT_FLAG_POOLED}


@pytest.mark.parametrize("pipeline", sorted(PIPELINES))
def test_pipeline_flat_star_bit_exact_vs_oracle(ctx, star, pipeline):
    from helpers import DEBUG_FLAGS
    W, H = 96, 54
    o = oracle_frames(star, W, H, spp=2, max_bounces=6, frames=2)
    g = gpu_frames(ctx, star, W, H, spp=2, max_bounces=6, frames=2, accel=capi.RT_ACCEL_FLAT,
                   flags=DEBUG_FLAGS | PIPELINES[pipeline])
    assert_identical(g, o)


@pytest.mark.parametrize("pipeline", sorted(PIPELINES))
@pytest.mark.parametrize("kind", ["ambient", "directional", "mixed", "nolights"])
def test_pipeline_light_types_equal_single_kernel(ctx, pipeline, kind):
    from helpers import DEBUG_FLAGS
    s = _custom_scene(kind)
    W, H = 72, 40
    a = gpu_frames(ctx, s, W, H, spp=3, max_bounces=5, accel=capi.RT_ACCEL_BVH)
    b = gpu_frames(ctx, s, W, H, spp=3, max_bounces=5, accel=capi.RT_ACCEL_BVH, flags=DEBUG_FLAGS | PIPELINES[pipeline])
    assert_identical(b, a)


@pytest.mark.parametrize("pipeline", sorted(PIPELINES))
@pytest.mark.parametrize("size", [(1, 1), (9, 17), (130, 5)])
def test_pipeline_ragged_sizes_and_strips(ctx, star, pipeline, size):
    from helpers import DEBUG_FLAGS
    W, H = size
    fl = DEBUG_FLAGS | PIPELINES[pipeline]
    a = gpu_frames(ctx, star, W, H, spp=2, max_bounces=3, accel=capi.RT_ACCEL_BVH)
    b = gpu_frames(ctx, star, W, H, spp=2, max_bounces=3, accel=capi.RT_ACCEL_BVH, flags=fl)
    assert_identical(b, a)
    if H >= 5:
        a = gpu_frames(ctx, star, W, H, spp=1, max_bounces=3, accel=capi.RT_ACCEL_BVH, strip=(2, 2, 1))
        b = gpu_frames(ctx, star, W, H, spp=1, max_bounces=3, accel=capi.RT_ACCEL_BVH, strip=(2, 2, 1), flags=fl)
        assert_identical(b, a)


def test_c1_at_full_size_against_the_oracle(ctx, teapot):
    """BASELINE.json's CPU-runnable configuration in full: Cornell box + teapot.glb, 640x360, 1 spp,
    primary + 1 bounce.  Flat search (the reference's algorithm): every output bit-identical to the
    oracle; BVH search: the same but for documented edge-tie pixels."""
    from raytracer_amd.workloads import WORKLOADS
    _, W, H, spp, mb, _ = WORKLOADS["c1"]
    o = oracle_frames(teapot, W, H, spp=spp, max_bounces=mb, threads=16)
    g = gpu_frames(ctx, teapot, W, H, spp=spp, max_bounces=mb, accel=capi.RT_ACCEL_FLAT)
    assert_identical(g, o)
    b = gpu_frames(ctx, teapot, W, H, spp=spp, max_bounces=mb, accel=capi.RT_ACCEL_BVH)
    assert_identical_but_edge_ties(b, o)
    assert (o["primary_ids"] >= 0).mean() > 0.3  # the box fills a good part of the frame


@pytest.mark.parametrize("pipeline", sorted(PIPELINES))
@pytest.mark.parametrize("strip", [(0, 1, 0), (3, 2, 1)])
def test_tile_order_feedback_changes_the_schedule_only(ctx, teapot, pipeline, strip):
    """From the second frame of a configuration on, the 8x8 tiles are handed out dearest first as
    measured on the frame before; RT_FLAG_FIXED_TILE_ORDER keeps image order.  Three accumulated
    frames of a ragged size, whole and as one rank's strips, must not differ by a bit."""
    from helpers import DEBUG_FLAGS
    W, H = 130, 37
    fl = DEBUG_FLAGS | PIPELINES[pipeline]
    a = gpu_frames(ctx, teapot, W, H, spp=2, max_bounces=4, frames=3, accel=capi.RT_ACCEL_BVH, strip=strip,
                   flags=fl | capi.RT_FLAG_FIXED_TILE_ORDER)
    b = gpu_frames(ctx, teapot, W, H, spp=2, max_bounces=4, frames=3, accel=capi.RT_ACCEL_BVH, strip=strip, flags=fl)
    assert_identical(b, a)
    owned = a["counts"] > 0
    assert owned.any() and (a["counts"][owned] == 3).all()


@pytest.mark.parametrize("pipeline", [capi.RT_FLAG_POOLED, capi.RT_FLAG_SINGLE_KERNEL])
@pytest.mark.parametrize("kind", ["teapot", "mixed"])
def test_tree_widths_agree(ctx, teapot, kind, pipeline):
    """Both kernels walk the four-wide form of the tree unless told otherwise (RT_OPT_BVH_WIDTH) or its stack
    does not fit, and the pooled kernel reads it in 128-byte nodes or in 80-byte nodes with binary16 boxes
    (RT_OPT_BVH_NODE_FORMAT; by itself only on scenes beyond 4 MB): every form holds the same leaves behind boxes that
    contain them, so they must agree bit for bit - with each other, and (but for edge ties) with the oracle."""
    from helpers import DEBUG_FLAGS
    W, H = 120, 68
    s = teapot if kind == "teapot" else _custom_scene(kind)
    o = oracle_frames(s, W, H, spp=2, max_bounces=5)
    got = {}
    pooled = pipeline == capi.RT_FLAG_POOLED
    try:
        for width, fmt in ((2, 0), (4, 32), (4, 16), (4, 0)):
            ctx.set_option(capi.RT_OPT_BVH_WIDTH, width)
            ctx.set_option(capi.RT_OPT_BVH_NODE_FORMAT, fmt)
            got[width, fmt] = gpu_frames(ctx, s, W, H, spp=2, max_bounces=5, accel=capi.RT_ACCEL_BVH, flags=DEBUG_FLAGS | pipeline)
            st = got[width, fmt]["stats"]
            assert st.bvh_width == width
            # (the single kernel has the 128-byte form only; a scene this small gets it from the pooled kernel too unless asked)
            assert st.bvh_node_bytes == (64 if width == 2 else 80 if (fmt == 16 and pooled) else 128)
    finally:
        ctx.set_option(capi.RT_OPT_BVH_WIDTH, 0)
        ctx.set_option(capi.RT_OPT_BVH_NODE_FORMAT, 0)
    for key in ((4, 32), (4, 16), (4, 0)):
        assert_identical(got[key], got[2, 0])
    assert_identical_but_edge_ties(got[4, 16], o)
    with pytest.raises(Exception):
        ctx.set_option(capi.RT_OPT_BVH_WIDTH, 3)
    with pytest.raises(Exception):
        ctx.set_option(capi.RT_OPT_BVH_NODE_FORMAT, 8)


@pytest.mark.parametrize("kind", ["empty", "one_sphere", "one_triangle", "three"])
def test_pooled_kernel_on_trees_of_a_single_node(ctx, kind):
    """Scenes whose wide tree is one node with 0, 1 or 3 of its four places used: the pooled kernel must never enter
    an empty place, and must agree with the oracle's flat loops bit for bit."""
    from helpers import DEBUG_FLAGS
    s = Scene(capi.RT_SCENE_EMPTY)
    m = s.add_material((0.7, 0.6, 0.5, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (0, 0, 0, 0), 0.5, 0.0, 1.5, 0.0)
    if kind in ("one_sphere", "three"):
        s.add_sphere((0.0, 0.0, -3.0), 1.0, m)
    if kind in ("one_triangle", "three"):
        s.add_triangle((-2, -1, -4), (2, -1, -4), (0, 2, -4), m)
    if kind == "three":
        s.add_sphere((1.5, 0.5, -2.5), 0.4, m)
        s.add_light((0.33, -0.67, -0.67, 0.0), (1, 1, 1, 3.0), capi.RT_LIGHT_DIRECTIONAL)
    w = s.wide_bvh_info()
    assert w["nodes"] == 1 and w["stack_need"] <= 2
    W, H = 160, 120  # 19 200 pixels
    o = oracle_frames(s, W, H, spp=2, max_bounces=3)
    try:
        for fmt, node_bytes in ((0, 128), (16, 80)):  # both node formats: an empty place is a negative half extent in either
            ctx.set_option(capi.RT_OPT_BVH_NODE_FORMAT, fmt)
            g = gpu_frames(ctx, s, W, H, spp=2, max_bounces=3, accel=capi.RT_ACCEL_BVH, flags=DEBUG_FLAGS | capi.RT_FLAG_POOLED)
            assert g["stats"].pipeline == capi.RT_PIPELINE_POOLED and g["stats"].bvh_width == 4 and g["stats"].bvh_node_bytes == node_bytes
            assert_identical(g, o)
    finally:
        ctx.set_option(capi.RT_OPT_BVH_NODE_FORMAT, 0)


@pytest.mark.parametrize("workload", ["c2", "c3", "c4", "c5"])
def test_full_size_frames_agree_across_pipelines(ctx, workload):
    """BASELINE.json's full sizes (C2: 1920x1080, 4 spp x 20 bounces on the star scene; C3: 1920x1080,
    16 spp x 4 bounces on the teapot scene; C4: the same scene at 3840x2160, 4 spp x 20 bounces per frame;
    C5: 1.07 M triangles, 1920x1080, 8 spp x 20 bounces), too large for the oracle: the single kernel and the pooled
    schedule are independent orders of execution of the same 2 million pixels and must give the
    same accumulated floats, counts, BGRA8 bytes, per-pixel path hashes and ray count; two
    accumulated frames also check that frame N+1 sees frame N's accumulation."""
    from helpers import DEBUG_FLAGS
    from raytracer_amd.workloads import WORKLOADS, workload_camera
    preset, W, H, spp, mb, _ = WORKLOADS[workload]
    s = Scene(preset)
    cam = workload_camera(workload, W, H)
    a = gpu_frames(ctx, s, W, H, spp, mb, frames=2, accel=capi.RT_ACCEL_BVH, camera=cam, flags=DEBUG_FLAGS | capi.RT_FLAG_SINGLE_KERNEL)
    b = gpu_frames(ctx, s, W, H, spp, mb, frames=2, accel=capi.RT_ACCEL_BVH, camera=cam, flags=DEBUG_FLAGS | capi.RT_FLAG_POOLED)
    assert_identical(b, a)
    assert a["stats"].pipeline == capi.RT_PIPELINE_SINGLE and b["stats"].pipeline == capi.RT_PIPELINE_POOLED
    # left to itself the library runs a configuration's first three frames on the pooled schedule, the
    # next two on the single kernel, and from then on the faster of the two timed ones (and without the
    # hash the light queries are bounded): seven such frames, whatever ran them, are seven frames of the
    # single kernel
    a5 = gpu_frames(ctx, s, W, H, spp, mb, frames=7, accel=capi.RT_ACCEL_BVH, camera=cam, flags=capi.RT_FLAG_SINGLE_KERNEL)
    ctx.upload_scene(s)
    ctx.resize(W, H)
    used = []
    for f in range(7):
        ctx.render_frame(cam, s.push_constants(0.0, f, f == 0), make_params(spp, mb, capi.RT_ACCEL_BVH))
        used.append(ctx.stats().pipeline)
    assert used[:5] == [capi.RT_PIPELINE_POOLED] * 3 + [capi.RT_PIPELINE_SINGLE] * 2
    assert used[5] in (capi.RT_PIPELINE_POOLED, capi.RT_PIPELINE_SINGLE) and used[6] in (capi.RT_PIPELINE_POOLED, capi.RT_PIPELINE_SINGLE)
    acc, cnt = ctx.read_accum()
    assert bit_equal_pixels(acc, a5["accum"]).all() and np.array_equal(ctx.read_bgra8(), a5["bgra"])
    assert (cnt == 7).all()
    assert (a["counts"] == 2).all()
    assert np.isfinite(a["accum"]).all()
    # every pixel's BGRA8 byte is the rounded mean of its accumulated colour (raytracer.comp:992-995)
    mean = np.clip(a["accum"] / 2.0, 0.0, 1.0)
    expect = np.rint(mean * np.float32(255.0)).astype(np.uint32)
    packed = expect[..., 2] | (expect[..., 1] << 8) | (expect[..., 0] << 16) | (expect[..., 3] << 24)
    assert np.array_equal(packed, a["bgra"].view(np.uint32).reshape(H, W))


@pytest.mark.parametrize("pipeline", [capi.RT_FLAG_SINGLE_KERNEL, capi.RT_FLAG_POOLED])
@pytest.mark.parametrize("kind", ["teapot", "directional", "mixed"])
def test_bounded_light_queries_leave_the_image_unchanged(ctx, teapot, kind, pipeline):
    """Without RT_FLAG_PATH_HASH the single kernel and the pooled schedule bound their light-visibility searches (they stop
    at the first occluder and ignore anything beyond the light); the accumulated image must
    not change by a bit, only the nearest-hit ids those searches would have reported."""
    s = teapot if kind == "teapot" else _custom_scene(kind)
    W, H = 160, 90
    from helpers import DEBUG_FLAGS
    exact = gpu_frames(ctx, s, W, H, spp=3, max_bounces=4, frames=2, accel=capi.RT_ACCEL_BVH, flags=DEBUG_FLAGS | pipeline)
    fast = gpu_frames(ctx, s, W, H, spp=3, max_bounces=4, frames=2, accel=capi.RT_ACCEL_BVH, flags=capi.RT_FLAG_PRIMARY_IDS | pipeline)
    assert bit_equal_pixels(exact["accum"], fast["accum"]).all()
    assert np.array_equal(exact["counts"], fast["counts"]) and np.array_equal(exact["bgra"], fast["bgra"])
    assert np.array_equal(exact["primary_ids"], fast["primary_ids"]) and exact["total_rays"] == fast["total_rays"]


@pytest.mark.parametrize("kind", ["wall", "wall_lamp", "wall_lamp_floor"])
def test_pooled_queues_when_every_ray_ends_at_once(ctx, kind):
    """A 1080p frame of one or two triangles that fill the view: every search ends within its first burst of
    steps, so a workgroup's 1 392 paths arrive at the first-hit stage together and each sends a light query and
    a bounce ray back at once - the fullest the pooled kernel's search queue can get (its ring holds two rays per slot).
    The frame must complete (no abandoned-frame error from rt_sync / rt_read_*) and equal the single kernel's
    bit for bit, without the hash (bounded light queries, fused frames) as well as with it."""
    s = Scene(capi.RT_SCENE_EMPTY)
    white = s.add_material((0.8, 0.8, 0.8, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (0, 0, 0, 0), 0.7, 0.0, 1.5, 0.0)
    lamp = s.add_material((1, 1, 1, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (1.0, 0.9, 0.7, 12.0), 0.5, 0.0, 1.5, 0.0)
    s.add_triangle((-40.0, -30.0, 4.0), (45.0, -30.0, 4.0), (2.75, 60.0, 4.0), white)
    if kind != "wall":
        s.add_triangle((0.0, 5.0, 0.0), (1.0, 5.0, 0.0), (0.0, 5.0, 1.0), lamp)
    if kind == "wall_lamp_floor":
        s.add_triangle((-60.0, 0.0, -20.0), (2.75, 0.0, 90.0), (70.0, 0.0, -20.0), white)
    W, H = 1920, 1080
    for flags in (DEBUG_FLAGS, 0):
        a = gpu_frames(ctx, s, W, H, spp=4, max_bounces=4, frames=2, flags=flags | capi.RT_FLAG_SINGLE_KERNEL)
        b = gpu_frames(ctx, s, W, H, spp=4, max_bounces=4, frames=2, flags=flags | capi.RT_FLAG_POOLED)
        assert b["stats"].pipeline == capi.RT_PIPELINE_POOLED
        if flags:
            assert_identical(b, a)
        else:
            assert bit_equal_pixels(a["accum"], b["accum"]).all() and np.array_equal(a["bgra"], b["bgra"])
            assert a["total_rays"] == b["total_rays"]
    # the same frames in one fused launch
    cam = default_camera(W, H)
    ctx.render_frames(cam, [s.push_constants(0.0, f, f == 0) for f in range(2)], make_params(4, 4, capi.RT_ACCEL_BVH, 0, 1, 0, capi.RT_FLAG_POOLED))
    acc, cnt = ctx.read_accum()
    assert ctx.stats().frames == 2
    assert bit_equal_pixels(acc, a["accum"]).all() and (cnt == 2).all()
    # what the instrumented kernel saw waiting in a workgroup's search queue stays inside its ring
    ctx.render_frame(cam, s.push_constants(0.0, 2, False), make_params(4, 4, capi.RT_ACCEL_BVH, 0, 1, 0, capi.RT_FLAG_POOLED | capi.RT_FLAG_TRAVERSAL_STATS))
    st = ctx.stats()
    assert 0 < st.search_queue_peak <= st.search_queue_size * 3 // 4, (st.search_queue_peak, st.search_queue_size)
    # ... also with half the waves shading only, which is when the queue is fullest
    ctx.set_option(capi.RT_OPT_POOL_SHADER_WAVES, 8)
    try:
        b8 = gpu_frames(ctx, s, W, H, spp=4, max_bounces=4, frames=2, flags=capi.RT_FLAG_POOLED | capi.RT_FLAG_TRAVERSAL_STATS, upload=False)
    finally:
        ctx.set_option(capi.RT_OPT_POOL_SHADER_WAVES, 0)
    assert bit_equal_pixels(a["accum"], b8["accum"]).all()
    assert 0 < b8["stats"].search_queue_peak <= b8["stats"].search_queue_size * 3 // 4


def test_shading_waves_change_the_schedule_only(ctx, teapot):
    """RT_OPT_POOL_SHADER_WAVES: however many of a workgroup's waves only shade (none, one, five, the most the kernel
    accepts; the default runs on kernel instances of their own), and however small a shading job may be, the pooled
    kernel's outputs are the same bits - single frames and a fused launch."""
    W, H = 480, 270
    ref = gpu_frames(ctx, teapot, W, H, spp=3, max_bounces=4, frames=2, flags=DEBUG_FLAGS | capi.RT_FLAG_POOLED)
    cam = default_camera(W, H)
    pcs = [teapot.push_constants(0.0, f, f == 0) for f in range(4)]
    ctx.render_frames(cam, pcs, make_params(3, 4, capi.RT_ACCEL_BVH, 0, 1, 0, capi.RT_FLAG_POOLED))
    fused_ref = ctx.read_accum()[0].copy()
    try:
        for waves, job in ((9, 64), (1, 16), (5, 48), (8, 1)):
            ctx.set_option(capi.RT_OPT_POOL_SHADER_WAVES, waves)
            ctx.set_option(capi.RT_OPT_POOL_JOB_MIN, job)
            g = gpu_frames(ctx, teapot, W, H, spp=3, max_bounces=4, frames=2, flags=DEBUG_FLAGS | capi.RT_FLAG_POOLED, upload=False)
            assert_identical(g, ref)
            ctx.render_frames(cam, pcs, make_params(3, 4, capi.RT_ACCEL_BVH, 0, 1, 0, capi.RT_FLAG_POOLED))
            assert ctx.stats().frames == 4
            assert bit_equal_pixels(ctx.read_accum()[0], fused_ref).all(), (waves, job)
    finally:
        ctx.set_option(capi.RT_OPT_POOL_SHADER_WAVES, 0)
        ctx.set_option(capi.RT_OPT_POOL_JOB_MIN, 0)


@pytest.mark.parametrize("workload,W,H,fused", [("c3", 240, 135, False), ("c3", 240, 135, True), ("c2", 240, 135, True), ("c4", 240, 135, False),
                                                ("c5", 32, 18, False)])
def test_benchmark_sampling_against_the_oracle(ctx, workload, W, H, fused):
    """Every benchmark configuration's own scene, camera and sampling (C3: Cornell box + teapot.glb, 16 spp x 4 bounces;
    C2: the reference's default scene, 4 spp x 20 bounces; C4: the teapot scene, 4 spp x 20 bounces; C5: 1.07 M triangles,
    8 spp x 20 bounces), two accumulated frames of the still camera, at a size where the oracle's flat loops finish in
    seconds: the pooled kernel - frame by frame, or as one fused launch where the image has enough tiles for one -
    against the oracle: accumulated floats, counts, BGRA8 bytes (and the ray count where no pixel is an edge tie)."""
    from raytracer_amd.workloads import WORKLOADS, workload_camera
    preset, _, _, spp, mb, _ = WORKLOADS[workload]
    s = Scene(preset)
    n = 2
    cam = workload_camera(workload, W, H)
    o = oracle_frames(s, W, H, spp=spp, max_bounces=mb, frames=n, camera=cam, threads=16)
    ctx.upload_scene(s)
    ctx.resize(W, H)
    pcs = [s.push_constants(0.0, f, f == 0) for f in range(n)]
    params = make_params(spp, mb, capi.RT_ACCEL_BVH, 0, 1, 0, capi.RT_FLAG_POOLED)
    rays = 0
    if fused:
        ctx.render_frames(cam, pcs, params)
        assert ctx.stats().frames == n
        rays = ctx.stats().rays
    else:
        for pc in pcs:
            ctx.render_frame(cam, pc, params)
            rays += ctx.stats().rays
    st = ctx.stats()
    assert st.pipeline == capi.RT_PIPELINE_POOLED and st.bvh_width == 4 and st.bvh_node_bytes == (80 if workload == "c5" else 128)
    acc, cnt = ctx.read_accum()
    same = bit_equal_pixels(acc, o["accum"])
    assert (~same).sum() <= 3, f"{(~same).sum()} of {same.size} pixels differ from the oracle"
    assert np.array_equal(cnt, o["counts"]) and (cnt == n).all()
    assert np.array_equal(ctx.read_bgra8()[same], o["bgra"][same])
    if same.all():
        assert rays == o["total_rays"]
