"""GPU parity on exact-t ties: rays that hit two or more coincident primitives at the same t.

The reference's flat loops resolve such ties by test order (raytracer.comp:469-508): across
spheres, scene triangles and meshes the EARLIEST-tested primitive keeps the hit (`closest_so_far > t`
is strict: 478, 489, 500), while inside one mesh the LAST triangle wins (`t_r > closest_so_far`
rejects only larger t: 447).  The device folds both rules into one priority per primitive; here
every pixel of the frame (and every later path vertex on the floor) is such a tie, and the flat
search, the BVH (default leaves and one primitive per leaf, so that the coincident primitives sit
in different leaves with identical boxes) and both kernels must reproduce the oracle bit for bit:
primary-hit ids, the hash of all nearest-hit ids, accumulated radiance, counts, BGRA8, ray count.
The coincident primitives carry different materials (one of each pair emissive or coloured), so a
wrong winner changes the radiance, not only the id.
"""
import numpy as np
import pytest

from helpers import DEBUG_FLAGS, ArrayScene, gpu_frames, oracle_frames
from raytracer_amd import capi
from raytracer_amd.renderer import Context, Scene, look_camera
from test_gpu_parity import assert_identical

WALL = [(-40.0, -30.0, 4.0), (45.0, -30.0, 4.0), (2.75, 60.0, 4.0)]     # fills the default camera's view
FLOOR = [(-60.0, 0.0, -20.0), (2.75, 0.0, 90.0), (70.0, 0.0, -20.0)]     # y = 0 under the camera
W, H = 64, 36


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def _materials(s):
    white = s.add_material((0.8, 0.8, 0.8, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (0, 0, 0, 0), 0.7, 0.0, 1.5, 0.0)
    green = s.add_material((0.1, 0.8, 0.2, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (0, 0, 0, 0), 0.3, 0.2, 1.5, 0.0)
    lamp = s.add_material((1, 1, 1, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (1.0, 0.9, 0.7, 12.0), 0.5, 0.0, 1.5, 0.0)
    glass = s.add_material((1, 1, 1, 1), (0.9, 1, 0.9, 1), (1, 1, 1, 0.5), (0, 0, 0, 0), 0.05, 0.0, 1.5, 1.0)
    return white, green, lamp, glass


def _scene(kind):
    s = Scene(capi.RT_SCENE_EMPTY)
    white, green, lamp, glass = _materials(s)
    camera = None
    if kind == "scene_pair":
        # wall: white first (wins), green second; floor: white first (wins), its emissive twin is the
        # scene's light - hidden behind the winner, yet lighting the wall through the visibility test
        s.add_triangle(*WALL, white)
        s.add_triangle(*WALL, green)
        s.add_triangle(*FLOOR, white)
        s.add_triangle(*FLOOR, lamp)
        return s, camera
    if kind == "emitter_first":
        # the emissive twin is tested first and wins: every path that reaches the floor ends there
        s.add_triangle(*WALL, green)
        s.add_triangle(*WALL, white)
        s.add_triangle(*FLOOR, lamp)
        s.add_triangle(*FLOOR, white)
        return s, camera
    if kind == "sphere_pair":
        # the camera sits inside two identical huge spheres (first wins) around two identical small ones
        s.add_sphere((2.75, 2.75, -6.0), 40.0, white)
        s.add_sphere((2.75, 2.75, -6.0), 40.0, lamp)
        s.add_sphere((2.75, 2.75, 2.0), 2.5, glass)
        s.add_sphere((2.75, 2.75, 2.0), 2.5, green)
        s.add_sphere((9.0, 9.0, 0.0), 1.0, lamp)
        return s, camera
    s.add_sphere((2.75, 12.0, -2.0), 1.5, lamp)  # a sphere light above the camera for the mesh scenes
    verts = WALL + FLOOR
    if kind == "mesh_pair":
        # the wall three times and the floor twice inside ONE mesh: the last copy of each wins
        return ArrayScene(s, verts, [0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 0, 1, 2], [(0, 15, green)]), camera
    if kind == "scene_vs_mesh":
        # scene triangles coincide with mesh triangles: the triangle loop runs first and wins
        s.add_triangle(*WALL, white)
        s.add_triangle(*FLOOR, glass)
        return ArrayScene(s, verts, [0, 1, 2, 3, 4, 5], [(0, 6, green)]), camera
    if kind == "two_meshes":
        # the same triangles in two meshes (the second one twice): the earlier mesh wins
        return ArrayScene(s, verts, [0, 1, 2, 3, 4, 5, 0, 1, 2, 0, 1, 2, 3, 4, 5], [(0, 6, white), (6, 15, green)]), camera
    if kind == "all":
        s.add_triangle(*WALL, green)
        s.add_triangle(*WALL, white)
        s.add_triangle(*FLOOR, white)
        return ArrayScene(s, verts, [0, 1, 2, 0, 1, 2, 3, 4, 5, 3, 4, 5], [(0, 12, glass)]), camera
    raise ValueError(kind)


KINDS = ["scene_pair", "emitter_first", "sphere_pair", "mesh_pair", "scene_vs_mesh", "two_meshes", "all"]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_exact_ties_resolve_as_the_flat_loops_do(ctx, kind):
    s, cam = _scene(kind)
    spp, mb = 3, 5
    o = oracle_frames(s, W, H, spp=spp, max_bounces=mb, frames=2, camera=cam)
    assert (o["primary_ids"] >= 0).all()  # every pixel's primary ray hits (a tie)
    runs = [("flat", capi.RT_ACCEL_FLAT, capi.RT_FLAG_SINGLE_KERNEL, 4), ("flat pooled", capi.RT_ACCEL_FLAT, capi.RT_FLAG_POOLED, 4)]
    for leaf in (4, 1, 2):
        runs += [(f"bvh leaf {leaf}", capi.RT_ACCEL_BVH, capi.RT_FLAG_SINGLE_KERNEL, leaf),
                 (f"bvh pooled leaf {leaf}", capi.RT_ACCEL_BVH, capi.RT_FLAG_POOLED, leaf)]
    try:
        for label, accel, pipe, leaf in runs:
            ctx.set_option(capi.RT_OPT_BVH_MAX_LEAF, leaf)
            g = gpu_frames(ctx, s, W, H, spp=spp, max_bounces=mb, frames=2, accel=accel, camera=cam, flags=DEBUG_FLAGS | pipe)
            try:
                assert_identical(g, o)
            except AssertionError as e:
                raise AssertionError(f"{kind} / {label}: {e}") from e
    finally:
        ctx.set_option(capi.RT_OPT_BVH_MAX_LEAF, 4)


def test_the_tie_scenes_do_tie():
    """The scenes above are what they claim (CPU only): in the oracle's own id map the winner is the
    documented one (first scene primitive, last triangle of a mesh, earlier loop)."""
    # ids: spheres, then scene triangles, then mesh triangles by index / 3; the camera sees wall and floor
    want = {"scene_pair": {0, 2}, "emitter_first": {0, 2}, "sphere_pair": {0, 2},
            "mesh_pair": {1 + 4, 1 + 3},  # after the light sphere: the wall's last copy (triangle 4), the floor's (3)
            "scene_vs_mesh": {1, 2}, "two_meshes": {1 + 0, 1 + 1}, "all": {1, 3}}
    for kind in KINDS:
        s, cam = _scene(kind)
        o = oracle_frames(s, W, H, spp=1, max_bounces=1, camera=cam)
        assert set(np.unique(o["primary_ids"])) == want[kind], kind
