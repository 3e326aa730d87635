"""ctypes binding of the CPU oracle (oracle/_build/librt_oracle.so) for the tests, the smoke
test and the CPU-baseline leg of the benchmark.  Never imported by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "_build", "librt_oracle.so")


class RtoScene(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("spheres", "materials", "lights", "triangles", "vertices", "indices", "meshes")]


class RtoParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width", "height", "spp", "max_bounces", "math_mode", "threads", "row_begin",
                                         "row_end", "strip_rows", "strip_count", "strip_index", "skybox")]


class RtoOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("accum", "counts", "bgra", "primary_ids", "path_hash", "rays")]


_lib = None


def build():
    subprocess.run(["make", "-C", ORACLE_DIR], check=True, capture_output=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            build()
        lib = C.CDLL(ORACLE_LIB)
        lib.rto_render.restype = C.c_int
        lib.rto_render.argtypes = [C.POINTER(RtoScene), C.c_void_p, C.c_void_p, C.POINTER(RtoParams), C.POINTER(RtoOutputs)]
        lib.rto_math.restype = C.c_float
        lib.rto_math.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float]
        lib.rto_hash.restype = C.c_uint32
        lib.rto_hash.argtypes = [C.c_uint32]
        lib.rto_xorshift.restype = C.c_uint32
        lib.rto_xorshift.argtypes = [C.c_uint32]
        lib.rto_random_from.restype = C.c_float
        lib.rto_random_from.argtypes = [C.c_uint32]
        lib.rto_pixel_seed.restype = C.c_uint32
        lib.rto_pixel_seed.argtypes = [C.c_float, C.c_uint32, C.c_int, C.c_int]
        lib.rto_hit_scene.restype = C.c_int
        lib.rto_hit_scene.argtypes = [C.POINTER(RtoScene), C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                      C.POINTER(C.c_float)]
        _lib = lib
    return _lib


class OracleScene:
    """Holds numpy copies of the seven scene arrays (byte layouts of the shader's SSBOs)."""

    def __init__(self, arrays: dict):
        self.arrays = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
        self.c = RtoScene(*[self.arrays[k].ctypes.data for k in
                            ("spheres", "materials", "lights", "triangles", "vertices", "indices", "meshes")])


def render(scene: OracleScene, camera, pc, width, height, spp, max_bounces, math_mode=0, threads=8, accum=None,
           counts=None, strip=(0, 1, 0), rows=(0, 0), want_debug=True, skybox=0):
    """One oracle frame.  `camera` / `pc` are ctypes structs with the 336 B / 44 B layouts.
    Returns dict(accum, counts, bgra, primary_ids, path_hash, rays)."""
    lib = load()
    n = width * height
    if accum is None:
        accum = np.zeros((height, width, 4), np.float32)
    if counts is None:
        counts = np.zeros((height, width), np.int32)
    bgra = np.zeros((height, width, 4), np.uint8)
    ids = np.full((height, width), -1, np.int32) if want_debug else None
    ph = np.zeros((height, width), np.uint32) if want_debug else None
    rays = C.c_uint64(0)
    p = RtoParams(width, height, spp, max_bounces, math_mode, threads, rows[0], rows[1], strip[0], strip[1], strip[2], skybox)
    o = RtoOutputs(accum.ctypes.data, counts.ctypes.data, bgra.ctypes.data, ids.ctypes.data if want_debug else None,
                   ph.ctypes.data if want_debug else None, C.addressof(rays))
    rc = lib.rto_render(C.byref(scene.c), C.addressof(camera), C.addressof(pc), C.byref(p), C.byref(o))
    assert rc == 0, f"rto_render failed: {rc}"
    assert n == accum.shape[0] * accum.shape[1]
    return {"accum": accum, "counts": counts, "bgra": bgra, "primary_ids": ids, "path_hash": ph, "rays": rays.value}
