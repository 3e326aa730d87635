"""GPU parity of what only scenes beyond 65 536 primitives exercise: the host builder's thread-pool path, the
four-wide collapse of a deep tree (its traversal can hold 44 stack entries on the 1.07 M-triangle scene), the
traversal stack's overflow into device memory (the pooled kernel keeps 8 entries per lane in LDS) and the launch
heuristics that switch at 16 MB of scene.  The 1.07 M-triangle teapot grid and a 36-teapot scene of 227 524
triangles (36 meshes with rebased indices) are compared with the CPU oracle's flat loops (raytracer.comp:408-508:
every ray against every primitive) at sizes the oracle affords, and the device's own flat search with its tree
search at a larger size.

Bar: every output bit-identical to the oracle (accumulated floats, counts, BGRA8, sample-0 primary ids, per-pixel
hash of all nearest-hit ids, ray count) except edge-tie pixels, at most 1e-4 of the image (one pixel at least)."""
import json
import os

import numpy as np
import pytest

from helpers import (DEBUG_FLAGS, assert_identical, assert_identical_but_edge_ties, bit_equal_pixels, gpu_frames,
                     oracle_frames)
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, look_camera, make_params
from raytracer_amd.workloads import WORKLOADS, workload_camera

pytestmark = pytest.mark.gpu

_uploaded = {}  # context handle -> the scene object it holds (an upload of these scenes builds their tree: seconds)


def frames_of(ctx, scene, *args, **kwargs):
    """gpu_frames, uploading the scene only if the context holds another."""
    key = ctx.handle.value
    fresh = _uploaded.get(key) is not scene
    _uploaded[key] = scene
    return gpu_frames(ctx, scene, *args, upload=fresh, **kwargs)


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def grid():
    s = Scene(WORKLOADS["c5"][0])
    assert s.bvh_info()["prims"] > 1_000_000 and s.wide_bvh_info()["stack_need"] > 32
    return s


def teapot_field(n):
    """n x n teapots over a floor under a lamp, each its own mesh (general index rebasing: the case
    scene.cpp:844-845 cannot express)."""
    s = Scene(capi.RT_SCENE_EMPTY)
    white = s.add_material((0.8, 0.8, 0.8, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (0, 0, 0, 0), 0.6, 0.0, 1.5, 0.0)
    gold = s.add_material((1.0, 0.78, 0.34, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (0, 0, 0, 0), 0.2, 1.0, 1.5, 0.0)
    lamp = s.add_material((1, 1, 1, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (1.0, 0.95, 0.8, 15.0), 0.5, 0.0, 1.5, 0.0)
    s.add_quad((20, -0.001, -10), (-20, -0.001, -10), (-20, -0.001, 30), (20, -0.001, 30), white)
    s.add_quad((6, 9, 2), (6, 9, 8), (0, 9, 8), (0, 9, 2), lamp)
    for gz in range(n):
        for gx in range(n):
            s.add_model("teapot.glb", (2.75 + 1.6 * (gx - (n - 1) / 2), 0.0, 1.0 + 1.6 * gz), 90.0, float((gx * 37 + gz * 101) % 360),
                        0.0, 0.2, gold, flags=capi.RT_MODEL_REBASE_INDICES)
    return s


@pytest.fixture(scope="module")
def grid_oracle(grid):
    W, H = 48, 27
    return oracle_frames(grid, W, H, spp=1, max_bounces=3, frames=2, camera=workload_camera("c5", W, H), threads=16)


@pytest.mark.parametrize("pipeline,width,node_format,node_bytes", [(capi.RT_FLAG_SINGLE_KERNEL, 2, 0, 64), (capi.RT_FLAG_POOLED, 2, 0, 64),
                                                                   (capi.RT_FLAG_POOLED, 4, 0, 80), (capi.RT_FLAG_POOLED, 4, 32, 128)])
def test_teapot_grid_against_the_oracle(ctx, grid, grid_oracle, pipeline, width, node_format, node_bytes):
    """1 068 084 primitives, the benchmark's camera, two accumulated frames: the single kernel (binary tree: the wide
    one's stack does not fit its LDS) and the pooled kernel on the binary tree and on the four-wide one in both node
    formats (left to itself it reads a scene of this size in 80-byte nodes with binary16 boxes) against the oracle's
    flat loops."""
    W, H = 48, 27
    try:
        ctx.set_option(capi.RT_OPT_BVH_WIDTH, width)
        ctx.set_option(capi.RT_OPT_BVH_NODE_FORMAT, node_format)
        g = frames_of(ctx, grid, W, H, spp=1, max_bounces=3, frames=2, camera=workload_camera("c5", W, H), flags=DEBUG_FLAGS | pipeline)
    finally:
        ctx.set_option(capi.RT_OPT_BVH_WIDTH, 0)
        ctx.set_option(capi.RT_OPT_BVH_NODE_FORMAT, 0)
    assert g["stats"].bvh_width == width and g["stats"].bvh_node_bytes == node_bytes
    assert g["stats"].pipeline == (capi.RT_PIPELINE_POOLED if pipeline == capi.RT_FLAG_POOLED else capi.RT_PIPELINE_SINGLE)
    assert_identical_but_edge_ties(g, grid_oracle)
    assert (grid_oracle["primary_ids"] >= 4).mean() > 0.3  # a good part of the frame sees teapots (ids 0-3 are floor and lamp)


def test_teapot_grid_flat_search_equals_the_oracle_bit_for_bit(ctx, grid, grid_oracle):
    W, H = 48, 27
    g = frames_of(ctx, grid, W, H, spp=1, max_bounces=3, frames=2, camera=workload_camera("c5", W, H), accel=capi.RT_ACCEL_FLAT)
    assert_identical(g, grid_oracle)


def test_teapot_grid_through_the_entry_point_that_allocates_nothing(ctx, grid, grid_oracle):
    """rt_render_frame_noalloc walks the binary tree on this scene (its fixed stack overflow area is too small for the
    wide one); no debug outputs there, so the accumulated floats, counts and BGRA8 bytes are what is compared."""
    W, H = 48, 27
    cam = workload_camera("c5", W, H)
    fb = Context(0)
    try:
        if _uploaded.get(ctx.handle.value) is not grid:
            ctx.upload_scene(grid)
            _uploaded[ctx.handle.value] = grid
        fb.resize(W, H)
        fb.sync()
        a, c, b = fb.device_buffers()
        rays = 0
        for f in range(2):
            ctx.render_frame_noalloc(cam, grid.push_constants(0.0, f, f == 0), make_params(1, 3, capi.RT_ACCEL_BVH, 0, 1, 0, capi.RT_FLAG_POOLED),
                                     a, c, b, W, H)
            rays += ctx.stats().rays
        assert ctx.stats().pipeline == capi.RT_PIPELINE_POOLED and ctx.stats().bvh_width == 2
        acc, cnt = fb.read_accum()
        same = bit_equal_pixels(acc, grid_oracle["accum"])
        assert (~same).sum() <= 1 and np.array_equal(cnt, grid_oracle["counts"])
        assert np.array_equal(fb.read_bgra8()[same], grid_oracle["bgra"][same])
        if same.all():
            assert rays == grid_oracle["total_rays"]
    finally:
        fb.close()


@pytest.fixture(scope="module")
def grid_flat_240(ctx, grid):
    W, H = 240, 135
    return frames_of(ctx, grid, W, H, spp=2, max_bounces=6, camera=workload_camera("c5", W, H), accel=capi.RT_ACCEL_FLAT)


@pytest.mark.parametrize("pipeline,node_format", [(capi.RT_FLAG_SINGLE_KERNEL, 0), (capi.RT_FLAG_POOLED, 0), (capi.RT_FLAG_POOLED, 32)])
def test_teapot_grid_tree_search_equals_flat_search_at_a_larger_size(ctx, grid, grid_flat_240, pipeline, node_format):
    """Too large for the oracle: the device's flat search (bit-identical to the oracle wherever that is checked) against
    its tree search, 240x135, 2 spp, 6 bounces, every output; the edge-tie pixels are listed."""
    W, H = 240, 135
    cam = workload_camera("c5", W, H)
    f = grid_flat_240
    try:
        ctx.set_option(capi.RT_OPT_BVH_NODE_FORMAT, node_format)
        b = frames_of(ctx, grid, W, H, spp=2, max_bounces=6, camera=cam, accel=capi.RT_ACCEL_BVH, flags=DEBUG_FLAGS | pipeline)
    finally:
        ctx.set_option(capi.RT_OPT_BVH_NODE_FORMAT, 0)
    if pipeline == capi.RT_FLAG_POOLED:
        assert b["stats"].bvh_node_bytes == (128 if node_format == 32 else 80)
    bad = assert_identical_but_edge_ties(b, f)
    same = bit_equal_pixels(b["accum"], f["accum"]) & (b["path_hash"] == f["path_hash"])
    ties = [{"x": int(x), "y": int(y), "primary_id_bvh": int(b["primary_ids"][y, x]), "primary_id_flat": int(f["primary_ids"][y, x])}
            for y, x in zip(*np.nonzero(~same))]
    print(f"edge-tie pixels at {W}x{H}, 2 spp, 6 bounces, 1.07 M-triangle scene: {bad} {ties}")
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        name = ("pooled" if pipeline == capi.RT_FLAG_POOLED else "single") + ("_nodes128" if node_format == 32 else "")
        with open(os.path.join(out, f"edge_ties_teapot_grid_{name}_240x135.json"), "w") as fh:
            json.dump({"width": W, "height": H, "spp": 2, "max_bounces": 6, "scene": "teapot_grid", "pixels": ties}, fh)


def test_teapot_grid_really_uses_the_stack_overflow(ctx, grid):
    """The frames above only mean something for the overflow area if rays get there: the instrumented pooled kernel
    counts the node visits that pushed beyond the 8 entries a lane keeps in LDS."""
    W, H = 240, 135
    cam = workload_camera("c5", W, H)
    for width in (4, 2):
        try:
            ctx.set_option(capi.RT_OPT_BVH_WIDTH, width)
            g = frames_of(ctx, grid, W, H, spp=2, max_bounces=6, camera=cam, flags=capi.RT_FLAG_POOLED | capi.RT_FLAG_TRAVERSAL_STATS)
        finally:
            ctx.set_option(capi.RT_OPT_BVH_WIDTH, 0)
        st = g["stats"]
        assert st.bvh_width == width and st.nodes_visited > 0
        assert st.stack_overflow_steps > 0, f"no traversal of the {width}-wide tree went beyond the LDS part of its stack"
        print(f"{width}-wide tree: {st.stack_overflow_steps} of {st.nodes_visited} node visits pushed into the device-memory overflow")


@pytest.mark.parametrize("pipeline", [capi.RT_FLAG_SINGLE_KERNEL, capi.RT_FLAG_POOLED])
def test_scene_of_36_meshes_beyond_the_builders_parallel_threshold(ctx, pipeline):
    """227 524 triangles in 36 meshes: above the 65 536 primitives from which the builder hands subtrees to a thread
    pool and above the 16 MB of scene from which the launch heuristics change, against the oracle's flat loops."""
    s = teapot_field(6)
    info = s.bvh_info()
    assert s.view().total_meshes == 36 and info["prims"] == 4 + 36 * 6320
    assert info["prims"] * 48 + info["nodes"] * 64 > 16 << 20
    W, H = 64, 36
    cam = look_camera((2.75, 3.2, -3.0), (0.0, -0.45, 1.0), (0.0, 1.0, 0.0), 55.0, W, H)
    o = oracle_frames(s, W, H, spp=1, max_bounces=4, camera=cam, threads=16)
    g = frames_of(ctx, s, W, H, spp=1, max_bounces=4, camera=cam, flags=DEBUG_FLAGS | pipeline)
    assert_identical_but_edge_ties(g, o)
    if pipeline == capi.RT_FLAG_POOLED:
        assert g["stats"].bvh_node_bytes == 80  # beyond 4 MB: binary16 boxes
    seen = o["primary_ids"][o["primary_ids"] >= 4]  # ids 0-3: floor and lamp; mesh m's triangles follow at 4 + 6320 m
    assert len(np.unique((seen - 4) // 6320)) > 10  # many of the meshes are seen
    f = frames_of(ctx, s, W, H, spp=1, max_bounces=4, camera=cam, accel=capi.RT_ACCEL_FLAT)
    assert_identical(f, o)
