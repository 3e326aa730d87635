"""The product's own .glb reader (host/glb_loader.cpp::LoadModel) against the third-party parser the
reference uses (tinygltf; src/tinygltf/loader.cpp:12-94).  tests/golden/loader_tinygltf.npz holds what
tinygltf yields for assets/star.glb and assets/teapot.glb (mesh 0 / primitive 0, positions and
widened indices), written by tests/golden/make_loader_golden.py in the build container; where the
reference tree is present the fixture itself is re-derived and compared.  No GPU."""
import os
import shutil
import sys

import numpy as np
import pytest

from raytracer_amd import capi
from raytracer_amd.renderer import Scene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loader_tinygltf.npz")


def _loaded(model):
    """Vertices and indices Scene::addModel appends for `model` under the identity transform
    (position 0, angles 0, scale 1: every coordinate is x * 1 + 0, exact)."""
    s = Scene(capi.RT_SCENE_EMPTY)
    grey = s.add_material((0.7, 0.7, 0.7, 1.0), (0, 0, 0, 0), (1, 1, 1, 0), (0, 0, 0, 0), 0.5, 0.0, 1.5, 0.0)
    s.add_model(model + ".glb", (0.0, 0.0, 0.0), 0.0, 0.0, 0.0, 1.0, grey)
    a = s.arrays()
    verts = a["vertices"].view(np.float32).reshape(-1, 4)
    idx = a["indices"].view(np.uint32)
    mesh = a["meshes"].view(np.uint32).reshape(-1, 3)[-1]
    n_idx = int(mesh[1] - mesh[0])
    model_idx = idx[mesh[0]:mesh[1]]
    n_verts = int(model_idx.max()) + 1
    # the scene's vectors open with one placeholder element (scene.cpp:26-32); the model's vertices follow it
    return verts[len(verts) - n_verts:, :3], model_idx, n_idx


@pytest.mark.parametrize("model", ["star", "teapot"])
def test_glb_reader_equals_tinygltf(model):
    g = np.load(GOLDEN)
    pos, idx = g[model + "_positions"], g[model + "_indices"]
    verts, model_idx, n_idx = _loaded(model)
    assert n_idx == idx.size and np.array_equal(model_idx, idx)
    assert verts.shape == pos.shape
    # byte for byte (+ 0.0 maps a -0.0 the identity transform may have made to +0.0 on both sides)
    assert np.array_equal((verts + np.float32(0)).view(np.uint32), (pos + np.float32(0)).view(np.uint32))


@pytest.mark.skipif(not os.path.isfile("/root/reference/src/tinygltf/tiny_gltf.h") or shutil.which("g++") is None,
                    reason="needs the reference tree (tiny_gltf.h) and g++: build container only")
@pytest.mark.parametrize("model", ["star", "teapot"])
def test_fixture_is_what_tinygltf_yields(model):
    sys.path.insert(0, os.path.dirname(GOLDEN))
    import make_loader_golden
    pos, idx = make_loader_golden.dump(model)
    g = np.load(GOLDEN)
    assert np.array_equal(pos.view(np.uint32), g[model + "_positions"].view(np.uint32))
    assert np.array_equal(idx, g[model + "_indices"])
