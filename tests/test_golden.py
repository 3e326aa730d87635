"""Golden fixtures (tests/golden/, written by make_golden.py): the oracle must keep
reproducing them bit for bit (CPU), and so must the HIP path through the C ABI (GPU)."""
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
from helpers import bit_equal_pixels, gpu_frames, oracle_frames
from raytracer_amd import capi
from raytracer_amd.renderer import Scene

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = {
    "star_64x36_s2_b4_f2": (capi.RT_SCENE_CORNELL_STAR, 64, 36, 2, 4, 2),
    "teapot_48x27_s1_b3_f1": (capi.RT_SCENE_CORNELL_TEAPOT, 48, 27, 1, 3, 1),
    "preset1_64x36_s1_b6_f1": (capi.RT_SCENE_PRESET1, 64, 36, 1, 6, 1),
}


def check(out, g):
    assert np.array_equal(out["counts"], g["counts"])
    assert np.array_equal(out["primary_ids"], g["primary_ids"])
    assert np.array_equal(out["path_hash"], g["path_hash"])
    assert bit_equal_pixels(out["accum"], g["accum"]).all()
    assert np.array_equal(out["bgra"], g["bgra"])
    assert int(out["total_rays"]) == int(g["total_rays"])


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name):
    preset, w, h, spp, mb, frames = CASES[name]
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    check(oracle_frames(Scene(preset), w, h, spp, mb, frames=frames, threads=4), g)


def test_rng_streams_golden():
    lib = ob.load()
    want = json.load(open(os.path.join(GOLDEN, "rng_streams.json")))
    for key, rec in want.items():
        t, f, x, y = key.split(",")
        s = lib.rto_pixel_seed(float(t), int(f), int(x), int(y))
        assert s == rec["seed"]
        for b in rec["random_bits"]:
            r = np.float32(lib.rto_random_from(s))
            s = lib.rto_xorshift(s)
            assert int(r.view(np.uint32)) == b
    assert want["0.0,0,0,0"]["seed"] == 0 and set(want["0.0,0,0,0"]["random_bits"]) == {0}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_flat_reproduces_golden(name):
    from raytracer_amd.renderer import Context
    preset, w, h, spp, mb, frames = CASES[name]
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    ctx = Context(0)
    try:
        check(gpu_frames(ctx, Scene(preset), w, h, spp, mb, frames=frames, accel=capi.RT_ACCEL_FLAT), g)
    finally:
        ctx.close()
