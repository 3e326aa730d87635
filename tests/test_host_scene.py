"""Host scene surface (the part of src/scene.cpp / src/tinygltf/loader.cpp / src/main.cpp the
hot path depends on): builders, placeholder rule, material clamps, light CDF, GLB loading,
camera block, BVH structure.  No GPU needed."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from raytracer_amd import capi
from raytracer_amd.renderer import RtError, Scene, default_camera, look_camera


def test_default_scene_counts_and_light_cdf():
    s = Scene(capi.RT_SCENE_CORNELL_STAR)
    v = s.view()
    # scene.cpp:350-695: 13 materials, 18 quads = 36 triangles, 4 spheres, star.glb = 60 triangles
    assert (v.total_spheres, v.total_triangles, v.total_meshes, v.n_materials) == (4, 36, 1, 13)
    assert (v.n_vertices, v.n_indices) == (180, 180)
    assert (v.meshes[0].index_start, v.meshes[0].index_end, v.meshes[0].material) == (0, 180, 8)
    # lights: the two triangles of the ceiling light quad (#14, #15), then the small blue sphere
    assert v.total_lights == 3 and v.lights_strength_sum == 42.0
    got = [(v.lights[i].type, v.lights[i].accumulated_str) for i in range(3)]
    assert got == [(capi.RT_LIGHT_TRIANGLE, 20.0), (capi.RT_LIGHT_TRIANGLE, 40.0), (capi.RT_LIGHT_SPHERE, 42.0)]
    assert [int(v.lights[i].pos_angle_aux[0]) for i in range(2)] == [14, 15]
    assert tuple(round(x, 6) for x in v.lights[2].pos_angle_aux) == (0.8, 0.15, 1.3, 0.15)


def test_placeholder_rule_and_empty_scene():
    s = Scene(capi.RT_SCENE_EMPTY)
    v = s.view()
    # scene.cpp:26-32: one placeholder per vector (not materials); totals stay 0
    assert (v.n_spheres, v.n_lights, v.n_triangles, v.n_meshes, v.n_vertices, v.n_indices, v.n_materials) == (1, 1, 1, 1, 1, 1, 0)
    assert (v.total_spheres, v.total_triangles, v.total_meshes, v.total_lights) == (0, 0, 0, 0)
    m = s.add_material((1, 1, 1, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (0, 0, 0, 0), 1.0, 0.0, 1.5, 0.0)
    s.add_sphere((0, 0, 0), 1.0, m)
    v = s.view()
    assert (v.n_spheres, v.total_spheres) == (1, 1)  # placeholder replaced, not appended to


def test_material_clamps():
    s = Scene(capi.RT_SCENE_EMPTY)
    i = s.add_material((2, -1, 0.5, 3), (0, 0, 0, 0), (1, 1, 1, 0.5), (5, 0.5, -2, 20.0), 0.0, 7.0, 1.0, -3.0)
    m = s.view().materials[i]
    assert tuple(m.albedo) == (1.0, 0.0, 0.5, 1.0)
    assert tuple(m.emission_color) == (1.0, 0.5, 0.0, 20.0)  # rgb clamped, strength kept (scene.cpp:740)
    assert m.roughness == np.float32(0.005) and m.metallic == 1.0 and m.trs_weight == 0.0
    assert m.ior == np.float32(1.00001)  # scene.cpp:744
    j = s.add_material((1, 1, 1, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (0, 0, 0, 0), 0.5, 0.0, -2.0, 0.0)
    assert s.view().materials[j].ior == 0.0


def test_quad_winding_and_emissive_registration():
    s = Scene(capi.RT_SCENE_EMPTY)
    lit = s.add_material((1, 1, 1, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (1, 1, 1, 4.0), 1.0, 0.0, 1.5, 0.0)
    s.add_quad((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), lit)
    v = s.view()
    t0, t1 = v.triangles[0], v.triangles[1]
    # scene.cpp:784-798: (v0,v1,v2) and (v2,v3,v0)
    assert (tuple(t0.v0), tuple(t0.v1), tuple(t0.v2)) == ((0, 0, 0), (1, 0, 0), (1, 1, 0))
    assert (tuple(t1.v0), tuple(t1.v1), tuple(t1.v2)) == ((1, 1, 0), (0, 1, 0), (0, 0, 0))
    assert tuple(t0.normal) == (0, 0, 1) and tuple(t1.normal) == (0, 0, 1)
    assert v.total_lights == 2 and v.lights[1].accumulated_str == 8.0 and v.lights_strength_sum == 8.0


def test_glb_loader_star_and_teapot():
    s = Scene(capi.RT_SCENE_EMPTY)
    m = s.add_material((1, 1, 1, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (0, 0, 0, 0), 1.0, 0.0, 1.5, 0.0)
    s.add_model("teapot.glb", (0, 0, 0), 0, 0, 0, 1.0, m)
    v = s.view()
    assert (v.n_vertices, v.n_indices) == (18960, 18960)  # 6320 unwelded triangles, u16 indices widened
    pos = np.ctypeslib.as_array(C.cast(v.vertices, C.POINTER(C.c_float)), shape=(v.n_vertices, 4))[:, :3]
    lo, hi = pos.min(axis=0), pos.max(axis=0)
    # identity transform: the file's own bounding box (node rotation ignored, as the reference loader does)
    assert np.allclose(lo, (-3.0, -2.0, -3.15), atol=2e-3) and np.allclose(hi, (3.434, 2.0, 0.0), atol=2e-3)
    idx = np.ctypeslib.as_array(v.indices, shape=(v.n_indices,))
    assert idx.max() == 18959 and len(np.unique(idx)) == 18960  # a permutation
    with pytest.raises(RtError) as e:
        s.add_model("does_not_exist.glb", (0, 0, 0), 0, 0, 0, 1.0, m)
    assert e.value.code == capi.RT_ERR_IO
    assert s.view().total_meshes == 2  # the failed model is still recorded, as an empty mesh (scene.cpp:812-846)


def test_model_transform_matches_numpy_float64():
    """T * yawPitchRoll(yaw,pitch,roll) * S applied to star.glb (scene.cpp:820-832)."""
    raw = Scene(capi.RT_SCENE_EMPTY)
    m = raw.add_material((1, 1, 1, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (0, 0, 0, 0), 1.0, 0.0, 1.5, 0.0)
    raw.add_model("star.glb", (0, 0, 0), 0, 0, 0, 1.0, m)
    v = raw.view()
    p0 = np.ctypeslib.as_array(C.cast(v.vertices, C.POINTER(C.c_float)), shape=(v.n_vertices, 4))[:, :3].astype(np.float64)
    s = Scene(capi.RT_SCENE_CORNELL_STAR)
    w = s.view()
    p1 = np.ctypeslib.as_array(C.cast(w.vertices, C.POINTER(C.c_float)), shape=(w.n_vertices, 4))[:, :3].astype(np.float64)
    yaw, pitch, roll, sc = math.radians(18.0), 0.0, math.radians(180.0), 0.1
    ch, sh, cp, sp, cb, sb = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
    R = np.array([[ch * cb + sh * sp * sb, -ch * sb + sh * sp * cb, sh * cp],
                  [sb * cp, cb * cp, -sp],
                  [-sh * cb + ch * sp * sb, sb * sh + ch * sp * cb, ch * cp]])
    want = (R @ (p0 * sc).T).T + np.array([3.5, 4.0, 3.1])
    assert np.allclose(p1, want, atol=1e-5)


def test_camera_block():
    cam = default_camera(1920, 1080)
    view = np.array(cam.view, np.float64).reshape(4, 4).T  # column-major
    vinv = np.array(cam.viewInv, np.float64).reshape(4, 4).T
    assert np.allclose(view @ vinv, np.eye(4), atol=1e-6)
    assert tuple(cam.position) == (2.75, 2.75, -6.0)
    assert cam.tanHalfFOV == np.float32(math.tan(math.radians(50.0) / 2.0))
    # looking down +z with up +y: camera-space -z maps to world +z
    d = vinv @ np.array([0, 0, -1, 0.0])
    assert np.allclose(d[:3], (0, 0, 1), atol=1e-7)
    cam2 = look_camera((0, 0, 0), (1, 0, 0), (0, 1, 0), 90.0, 100, 50)
    assert abs(cam2.tanHalfFOV - 1.0) < 1e-7


def test_bvh_structure_of_every_preset():
    for preset, prims in ((capi.RT_SCENE_CORNELL_STAR, 100), (capi.RT_SCENE_PRESET1, 15), (capi.RT_SCENE_CORNELL_TEAPOT, 6360)):
        info = Scene(preset).bvh_info()  # raises if the structural validation fails
        assert info["prims"] == prims and 1 <= info["depth"] <= 30


def test_preset1_with_the_teapot_the_reference_leaves_commented_out():
    """scene.cpp:333-343 defines a teapot for createPreset1 and comments its addModel call out;
    RT_SCENE_PRESET1_TEAPOT makes that call: everything else is createPreset1 unchanged."""
    plain, pot = Scene(capi.RT_SCENE_PRESET1), Scene(capi.RT_SCENE_PRESET1_TEAPOT)
    a, b = plain.arrays(), pot.arrays()
    for k in ("spheres", "materials", "lights", "triangles"):
        assert a[k].tobytes() == b[k].tobytes(), k
    vp, vt = plain.view(), pot.view()
    assert (vp.total_meshes, vt.total_meshes) == (0, 1)
    assert vt.n_vertices == 18960 and vt.n_indices == 18960
    start, end, material = np.frombuffer(b["meshes"].tobytes(), dtype=np.int32)[:3]
    assert (int(start), int(end)) == (0, 18960) and 0 <= int(material) < vt.n_materials
    info = pot.bvh_info()
    assert info["prims"] == 15 + 6320
    # pitch 90 about x, scale 1, moved to (0, -1, 10): the model's bounds sit around that point
    v = b["vertices"].view(np.float32).reshape(-1, 4)[:, :3]
    assert abs(float(v[:, 0].mean())) < 1.5 and 5.0 < float(v[:, 2].mean()) < 15.0


def test_scene_buffer_hashes_are_pinned():
    """Golden: sha256 of the seven buffers of the default scenes (tests/golden/scene_hashes.json)."""
    import hashlib
    import json
    path = os.path.join(os.path.dirname(__file__), "golden", "scene_hashes.json")
    want = json.load(open(path))
    for name, preset in (("cornell_star", capi.RT_SCENE_CORNELL_STAR), ("preset1", capi.RT_SCENE_PRESET1),
                         ("cornell_teapot", capi.RT_SCENE_CORNELL_TEAPOT)):
        arrays = Scene(preset).arrays()
        got = {k: hashlib.sha256(v.tobytes()).hexdigest() for k, v in arrays.items()}
        assert got == want[name], name


def test_parallel_bvh_build_lays_the_tree_out_as_the_serial_build_does(monkeypatch):
    """Scenes of 65 536 primitives or more build their subtrees on several threads; nodes and leaf
    order must come out exactly as from the serial depth-first build (RT_BVH_SERIAL forces that)."""
    import time
    s = Scene(capi.RT_SCENE_TEAPOT_GRID)
    t0 = time.perf_counter()
    parallel = s.bvh_hash()
    t1 = time.perf_counter()
    info = s.bvh_info()  # validates the structure
    monkeypatch.setenv("RT_BVH_SERIAL", "1")
    t2 = time.perf_counter()
    serial = s.bvh_hash()
    t3 = time.perf_counter()
    assert parallel == serial
    assert info["prims"] > 1_000_000
    print(f"1.07 M triangles: parallel build {t1 - t0:.2f} s, serial {t3 - t2:.2f} s")


def test_wide_tree_holds_the_binary_trees_leaves():
    """The four-wide form of the tree, which the pooled kernel walks: collapse + structural check (every leaf of
    the binary tree exactly once, every node reachable once, stack need and depth as stored; and for its 80-byte
    form with binary16 boxes: the same children, every box - a point of the node + relative centre -+ half extent, as
    real numbers - contains the 128-byte form's box, no subnormal halves, empty places not enterable)."""
    e = Scene(capi.RT_SCENE_EMPTY)
    assert e.wide_bvh_info()["nodes"] == 1
    for preset in (capi.RT_SCENE_CORNELL_STAR, capi.RT_SCENE_CORNELL_TEAPOT, capi.RT_SCENE_PRESET1_TEAPOT, capi.RT_SCENE_TEAPOT_GRID):
        s = Scene(preset)
        b, w = s.bvh_info(), s.wide_bvh_info()  # raises if the structural validation fails
        assert 0 < w["nodes"] <= b["nodes"]
        assert w["depth"] <= b["depth"]
        # a binary tree of depth d holds at most d - 1 entries; merging levels at most triples that
        assert w["depth"] - 1 <= w["stack_need"] <= 3 * (b["depth"] - 1)
