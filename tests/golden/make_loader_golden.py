#!/usr/bin/env python3
"""Writes tests/golden/loader_tinygltf.npz: what the third-party parser the reference loads its
models with (tinygltf, vendored under the reference's src/tinygltf/) yields for the two asset files,
taken out the way the reference's LoadModel takes it (src/tinygltf/loader.cpp:37-91).  Needs the
reference tree (for tiny_gltf.h) and g++: run in the build container, `python tests/golden/make_loader_golden.py`."""
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def dump(model: str):
    """(positions float32 [n,3], indices uint32 [m]) of assets/<model>.glb through oracle/_ref/tinygltf_check."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, capture_output=True)
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, model + ".bin")
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "tinygltf_check"), os.path.join(ROOT, "assets", model + ".glb"), out],
                       check=True)
        raw = np.fromfile(out, np.uint8)
    nv, ni = (int(v) for v in raw[:8].view(np.uint32))
    pos = raw[8:8 + 12 * nv].view(np.float32).reshape(nv, 3).copy()
    idx = raw[8 + 12 * nv:8 + 12 * nv + 4 * ni].view(np.uint32).copy()
    assert raw.size == 8 + 12 * nv + 4 * ni
    return pos, idx


if __name__ == "__main__":
    data = {}
    for model in ("star", "teapot"):
        pos, idx = dump(model)
        data[model + "_positions"] = pos
        data[model + "_indices"] = idx
        print(model, pos.shape, idx.shape)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "loader_tinygltf.npz"), **data)
