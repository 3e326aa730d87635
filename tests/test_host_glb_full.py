"""The general GLB reader (RT_MODEL_FULL_GLTF) and index rebasing (RT_MODEL_REBASE_INDICES): what
the reference's loader leaves out (src/tinygltf/loader.cpp:37-43 first mesh / first primitive only,
76-91 byteStride ignored, no node transforms; src/scene.cpp:844-845 indices not rebased).  The files
are written here, so the expected vertices are known exactly.  No GPU."""
import json
import struct

import numpy as np
import pytest

from raytracer_amd import capi
from raytracer_amd.renderer import RtError, Scene


GREY = ((0.7, 0.7, 0.7, 1.0), (0, 0, 0, 0), (1, 1, 1, 0), (0, 0, 0, 0), 0.5, 0.0, 1.5, 0.0)  # a plain diffuse material


def write_glb(path, doc, blob):
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    blob = blob + b"\0" * (-len(blob) % 4)
    total = 12 + 8 + len(js) + 8 + len(blob)
    with open(path, "wb") as f:
        f.write(struct.pack("<III", 0x46546C67, 2, total))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(blob), 0x004E4942) + blob)


def quat(axis, deg):
    a = np.radians(deg) / 2.0
    ax = np.asarray(axis, float) / np.linalg.norm(axis)
    return [float(v) for v in ax * np.sin(a)] + [float(np.cos(a))]


def quat_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def trs(t=(0, 0, 0), q=(0, 0, 0, 1), s=(1, 1, 1)):
    m = np.eye(4)
    m[:3, :3] = quat_matrix(q) @ np.diag(s)
    m[:3, 3] = t
    return m


@pytest.fixture()
def two_mesh_file(tmp_path):
    """Mesh 0: one primitive, positions interleaved with normals (byteStride 24), 8-bit indices.
    Mesh 1: two primitives, one with 16-bit indices and one without indices, plus a line primitive
    that must be skipped.  Node 0 (translation+rotation+scale) holds mesh 0 and has child node 1
    (a matrix) holding mesh 1; node 2 holds mesh 0 again under a translation only."""
    quad = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    inter = np.concatenate([quad, np.tile(np.float32([0, 0, 1]), (4, 1))], axis=1)  # pos | normal
    idx8 = np.array([0, 1, 2, 0, 2, 3], np.uint8)
    tri_a = np.array([[0, 0, 1], [2, 0, 1], [0, 2, 1], [2, 2, 1]], np.float32)
    idx16 = np.array([0, 1, 2, 2, 1, 3], np.uint16)
    tri_b = np.array([[5, 5, 5], [6, 5, 5], [5, 6, 5]], np.float32)  # non-indexed triangle
    line = np.array([[9, 9, 9], [8, 8, 8]], np.float32)
    parts, views = [], []

    def view(data, stride=None):
        off = sum(len(p) for p in parts)
        raw = data.tobytes()
        parts.append(raw + b"\0" * (-len(raw) % 4))
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(raw)}
        if stride:
            v["byteStride"] = stride
        views.append(v)
        return len(views) - 1

    v_inter, v_i8, v_a, v_i16, v_b, v_line = view(inter, 24), view(idx8), view(tri_a), view(idx16), view(tri_b), view(line)
    accessors = [
        {"bufferView": v_inter, "componentType": 5126, "count": 4, "type": "VEC3"},                    # 0 positions, strided
        {"bufferView": v_inter, "byteOffset": 12, "componentType": 5126, "count": 4, "type": "VEC3"},  # 1 normals
        {"bufferView": v_i8, "componentType": 5121, "count": 6, "type": "SCALAR"},                     # 2
        {"bufferView": v_a, "componentType": 5126, "count": 4, "type": "VEC3"},                        # 3
        {"bufferView": v_i16, "componentType": 5123, "count": 6, "type": "SCALAR"},                    # 4
        {"bufferView": v_b, "componentType": 5126, "count": 3, "type": "VEC3"},                        # 5
        {"bufferView": v_line, "componentType": 5126, "count": 2, "type": "VEC3"},                     # 6
    ]
    q0 = quat([0, 0, 1], 90.0)
    m1 = trs(t=(0, 0, 3), s=(2, 2, 2))
    doc = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0, 2]}],
        "nodes": [
            {"mesh": 0, "translation": [10, 0, 0], "rotation": q0, "scale": [2, 1, 1], "children": [1]},
            {"mesh": 1, "matrix": [float(v) for v in m1.T.reshape(-1)]},
            {"mesh": 0, "translation": [0, -4, 0]},
        ],
        "meshes": [
            {"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1}, "indices": 2}]},
            {"primitives": [{"attributes": {"POSITION": 3}, "indices": 4, "mode": 4},
                            {"attributes": {"POSITION": 6}, "mode": 1},
                            {"attributes": {"POSITION": 5}}]},
        ],
        "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": sum(len(p) for p in parts)}],
    }
    write_glb(tmp_path / "two.glb", doc, b"".join(parts))
    w0 = trs((10, 0, 0), q0, (2, 1, 1))
    w1 = w0 @ m1
    w2 = trs((0, -4, 0))

    def xf(m, pts):
        return (np.c_[pts.astype(np.float64), np.ones(len(pts))] @ m.T)[:, :3]

    verts = np.concatenate([xf(w0, quad), xf(w1, tri_a), xf(w1, tri_b), xf(w2, quad)])
    idx = np.concatenate([idx8.astype(np.uint32), idx16.astype(np.uint32) + 4, np.arange(3, dtype=np.uint32) + 8,
                          idx8.astype(np.uint32) + 11])
    return tmp_path, verts, idx


def model_arrays(scene):
    a = scene.arrays()
    return a["vertices"].view(np.float32).reshape(-1, 4), a["indices"].view(np.uint32), a["meshes"].view(np.uint32).reshape(-1, 3)


def test_general_reader_walks_nodes_strides_and_all_primitives(two_mesh_file):
    d, verts, idx = two_mesh_file
    s = Scene(capi.RT_SCENE_EMPTY, assets_dir=str(d))
    m = s.add_material(*GREY)
    s.add_model("two.glb", (0, 0, 0), 0, 0, 0, 1.0, m, flags=capi.RT_MODEL_FULL_GLTF)
    v, i, meshes = model_arrays(s)
    assert len(v) == 15 and len(i) == 21 and list(meshes[0][:2]) == [0, 21]
    np.testing.assert_allclose(v[:, :3], verts, atol=2e-5)
    assert np.array_equal(i, idx)


def test_reference_reader_semantics_on_such_files(two_mesh_file, tmp_path):
    """LoadModel (loader.cpp:37-43, 57-68, 76-91): mesh 0 / primitive 0 only, no node transform,
    8-bit indices refused, byteStride ignored - interleaved normals are read as positions, exactly
    as the reference would."""
    d, _, _ = two_mesh_file
    s = Scene(capi.RT_SCENE_EMPTY, assets_dir=str(d))
    m = s.add_material(*GREY)
    with pytest.raises(RtError) as e:
        s.add_model("two.glb", (0, 0, 0), 0, 0, 0, 1.0, m)  # its first primitive has 8-bit indices
    assert e.value.code == capi.RT_ERR_IO
    quad = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    inter = np.concatenate([quad, np.tile(np.float32([0, 0, 1]), (4, 1))], axis=1)
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    blob = inter.tobytes() + idx.tobytes()
    doc = {"asset": {"version": "2.0"}, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0, "translation": [5, 5, 5]}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1}]}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"},
                         {"bufferView": 1, "componentType": 5123, "count": 6, "type": "SCALAR"}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 96, "byteStride": 24},
                           {"buffer": 0, "byteOffset": 96, "byteLength": 12}],
           "buffers": [{"byteLength": len(blob)}]}
    write_glb(tmp_path / "strided.glb", doc, blob)
    s = Scene(capi.RT_SCENE_EMPTY, assets_dir=str(tmp_path))
    m = s.add_material(*GREY)
    s.add_model("strided.glb", (0, 0, 0), 0, 0, 0, 1.0, m)
    v, i, _ = model_arrays(s)
    assert len(v) == 4 and list(i) == [0, 1, 2, 0, 2, 3]
    assert v[:, :3].tolist() == [[0, 0, 0], [0, 0, 1], [1, 0, 0], [0, 0, 1]]  # pos0 nrm0 pos1 nrm1, untranslated
    s = Scene(capi.RT_SCENE_EMPTY, assets_dir=str(tmp_path))
    m = s.add_material(*GREY)
    s.add_model("strided.glb", (0, 0, 0), 0, 0, 0, 1.0, m, flags=capi.RT_MODEL_FULL_GLTF)
    v, _, _ = model_arrays(s)
    assert v[:, :3].tolist() == (quad + 5).tolist()


def test_rebased_indices_let_two_models_share_the_vertex_buffer(two_mesh_file):
    d, verts, idx = two_mesh_file
    s = Scene(capi.RT_SCENE_EMPTY, assets_dir=str(d))
    m = s.add_material(*GREY)
    fl = capi.RT_MODEL_FULL_GLTF | capi.RT_MODEL_REBASE_INDICES
    s.add_model("two.glb", (0, 0, 0), 0, 0, 0, 1.0, m, flags=fl)
    s.add_model("two.glb", (100, 0, 0), 0, 0, 0, 1.0, m, flags=fl)
    v, i, meshes = model_arrays(s)
    assert len(v) == 30 and len(i) == 42 and [list(r[:2]) for r in meshes] == [[0, 21], [21, 42]]
    assert np.array_equal(i[:21], idx) and np.array_equal(i[21:], idx + 15)
    np.testing.assert_allclose(v[15:, :3], verts + [100, 0, 0], atol=2e-5)
    # without the flag the second model's indices point into the first model's vertices (scene.cpp:844-845)
    s2 = Scene(capi.RT_SCENE_EMPTY, assets_dir=str(d))
    m2 = s2.add_material(*GREY)
    s2.add_model("two.glb", (0, 0, 0), 0, 0, 0, 1.0, m2, flags=capi.RT_MODEL_FULL_GLTF)
    s2.add_model("two.glb", (100, 0, 0), 0, 0, 0, 1.0, m2, flags=capi.RT_MODEL_FULL_GLTF)
    _, i2, _ = model_arrays(s2)
    assert np.array_equal(i2[21:], idx)
    # the BVH built over the rebased scene is structurally sound and holds both models' triangles
    info = s.bvh_info()
    assert info["prims"] == 14


def test_general_reader_agrees_with_the_reference_reader_on_the_shipped_assets():
    """star.glb and teapot.glb hold one mesh with one primitive; where their nodes carry no
    transform both readers must produce the same buffers."""
    for name in ("star.glb", "teapot.glb"):
        a, b = Scene(capi.RT_SCENE_EMPTY), Scene(capi.RT_SCENE_EMPTY)
        ma, mb = a.add_material(*GREY), b.add_material(*GREY)
        a.add_model(name, (0, 0, 0), 0, 0, 0, 1.0, ma)
        b.add_model(name, (0, 0, 0), 0, 0, 0, 1.0, mb, flags=capi.RT_MODEL_FULL_GLTF)
        va, ia, _ = model_arrays(a)
        vb, ib, _ = model_arrays(b)
        assert np.array_equal(ia, ib) and va.shape == vb.shape
        if not np.array_equal(va, vb):  # the file's node has a transform, which only the general reader applies
            assert not np.allclose(va, vb)


@pytest.mark.parametrize("breakage", ["index_out_of_range", "stride_too_small", "view_beyond_bin", "cycle", "no_triangles",
                                      "huge_count", "huge_count_strided", "negative_offset", "nan_count"])
def test_general_reader_rejects_malformed_files(tmp_path, breakage):
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    idx = np.array([0, 1, 2], np.uint16)
    blob = pos.tobytes() + idx.tobytes() + b"\0\0"
    views = [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 6}]
    accessors = [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"},
                 {"bufferView": 1, "componentType": 5123, "count": 3, "type": "SCALAR"}]
    nodes = [{"mesh": 0}]
    prim = {"attributes": {"POSITION": 0}, "indices": 1}
    if breakage == "index_out_of_range":
        blob = pos.tobytes() + np.array([0, 1, 7], np.uint16).tobytes() + b"\0\0"
    elif breakage == "stride_too_small":
        views[0]["byteStride"] = 8
    elif breakage == "view_beyond_bin":
        views[1]["byteLength"] = 600
    elif breakage == "cycle":
        nodes = [{"mesh": 0, "children": [1]}, {"children": [0]}]
    elif breakage == "no_triangles":
        prim["mode"] = 0
    elif breakage == "huge_count":          # count * element size would wrap a 64-bit product
        accessors[1]["count"] = 2 ** 63 + 3
    elif breakage == "huge_count_strided":  # (count - 1) * stride + element would wrap and pass a multiplied check
        accessors[0]["count"] = 2 ** 61 + 1
        views[0]["byteStride"] = 16
    elif breakage == "negative_offset":
        accessors[0]["byteOffset"] = -12
    elif breakage == "nan_count":
        accessors[1]["count"] = 1e400       # reads as infinity
    doc = {"asset": {"version": "2.0"}, "scenes": [{"nodes": [0]}], "nodes": nodes, "meshes": [{"primitives": [prim]}],
           "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": len(blob)}]}
    write_glb(tmp_path / "bad.glb", doc, blob)
    s = Scene(capi.RT_SCENE_EMPTY, assets_dir=str(tmp_path))
    m = s.add_material(*GREY)
    with pytest.raises(RtError) as e:
        s.add_model("bad.glb", (0, 0, 0), 0, 0, 0, 1.0, m, flags=capi.RT_MODEL_FULL_GLTF)
    assert e.value.code == capi.RT_ERR_IO
    # like the reference (scene.cpp:812-814) the scene still gets an (empty) mesh entry
    assert s.view().total_meshes == 1
    if breakage in ("huge_count", "huge_count_strided", "negative_offset", "nan_count"):
        # the reference-style reader (mesh 0 / primitive 0) refuses the same numbers
        s2 = Scene(capi.RT_SCENE_EMPTY, assets_dir=str(tmp_path))
        m2 = s2.add_material(*GREY)
        with pytest.raises(RtError) as e2:
            s2.add_model("bad.glb", (0, 0, 0), 0, 0, 0, 1.0, m2)
        assert e2.value.code == capi.RT_ERR_IO
