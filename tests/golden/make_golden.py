#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the CPU oracle (oracle/rt_oracle.cpp,
math_mode 0) and the host scene builders.  The reference ships no fixtures of its own and its
shader cannot run in the build environment, so these vectors pin the oracle against drift and
give the GPU tests a committed target.  Re-run only when the arithmetic contract changes:

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_binding as ob  # noqa: E402
from helpers import oracle_frames  # noqa: E402
from raytracer_amd import capi  # noqa: E402
from raytracer_amd.renderer import Scene  # noqa: E402

CASES = {
    # name: (preset, width, height, spp, max_bounces, frames)
    "star_64x36_s2_b4_f2": (capi.RT_SCENE_CORNELL_STAR, 64, 36, 2, 4, 2),
    "teapot_48x27_s1_b3_f1": (capi.RT_SCENE_CORNELL_TEAPOT, 48, 27, 1, 3, 1),
    "preset1_64x36_s1_b6_f1": (capi.RT_SCENE_PRESET1, 64, 36, 1, 6, 1),
}


def main():
    hashes = {}
    for name, preset in (("cornell_star", capi.RT_SCENE_CORNELL_STAR), ("preset1", capi.RT_SCENE_PRESET1),
                         ("cornell_teapot", capi.RT_SCENE_CORNELL_TEAPOT)):
        arrays = Scene(preset).arrays()
        hashes[name] = {k: hashlib.sha256(v.tobytes()).hexdigest() for k, v in arrays.items()}
    json.dump(hashes, open(os.path.join(HERE, "scene_hashes.json"), "w"), indent=1, sort_keys=True)

    for name, (preset, w, h, spp, mb, frames) in CASES.items():
        out = oracle_frames(Scene(preset), w, h, spp, mb, frames=frames)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), accum=out["accum"], counts=out["counts"],
                            bgra=out["bgra"], primary_ids=out["primary_ids"], path_hash=out["path_hash"],
                            total_rays=np.int64(out["total_rays"]))
        print(name, "rays", out["total_rays"], "mean", out["accum"][..., :3].mean())

    lib = ob.load()
    streams = {}
    for (t, f, x, y) in ((0.0, 0, 1, 0), (0.0, 1, 0, 0), (2.5, 7, 959, 539), (0.0, 0, 0, 0)):
        s = lib.rto_pixel_seed(t, f, x, y)
        vals, bits = [], []
        for _ in range(8):
            r = np.float32(lib.rto_random_from(s))
            s = lib.rto_xorshift(s)
            vals.append(float(r))
            bits.append(int(r.view(np.uint32)))
        streams[f"{t},{f},{x},{y}"] = {"seed": int(lib.rto_pixel_seed(t, f, x, y)), "random_bits": bits}
    json.dump(streams, open(os.path.join(HERE, "rng_streams.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
