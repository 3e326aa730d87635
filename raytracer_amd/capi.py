"""ctypes view of the C ABI declared in include/rt_abi.h.

This module only loads ``lib/libraytracer_amd.so`` and mirrors its structs and
prototypes; it contains no rendering logic and no fallback: if the shared library is
missing, loading raises.  The library is built by ``raytracer_amd/csrc/Makefile``
(see ``__graft_entry__.build``).
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_LIB_PATH: load another build of the same ABI (A/B timing of two builds on one device)
LIB_PATH = os.environ.get("RT_LIB_PATH") or os.path.join(_HERE, "lib", "libraytracer_amd.so")
REPO_ROOT = os.path.dirname(_HERE)
ASSETS_DIR = os.path.join(REPO_ROOT, "assets") + os.sep

RT_ABI_VERSION = 2  # include/rt_abi.h
RT_OK = 0
RT_ERR_INVALID_ARGUMENT = -1
RT_ERR_HIP = -2
RT_ERR_NO_SCENE = -3
RT_ERR_NO_FRAMEBUFFER = -4
RT_ERR_UNSUPPORTED = -5
RT_ERR_IO = -6
RT_ERR_TOO_LARGE = -7

RT_ACCEL_FLAT = 0
RT_ACCEL_BVH = 1

RT_FLAG_PRIMARY_IDS = 1
RT_FLAG_PATH_HASH = 2
RT_FLAG_COUNT_RAYS = 4
RT_FLAG_TRAVERSAL_STATS = 8
RT_FLAG_RESERVED_16 = 16
RT_FLAG_POOLED = 32
RT_FLAG_SINGLE_KERNEL = 64
RT_FLAG_FIXED_TILE_ORDER = 128
RT_MODEL_FULL_GLTF, RT_MODEL_REBASE_INDICES = 1, 2
RT_SKY_GREY, RT_SKY_DAY, RT_SKY_NIGHT, RT_SKY_WHITE, RT_SKY_BLACK = 0, 1, 2, 3, 4
RT_PIPELINE_SINGLE, RT_PIPELINE_POOLED = 0, 2
RT_GROUP_STRIPS, RT_GROUP_SAMPLES = 0, 1
RT_OPT_EXIT_QUOTA_PCT, RT_OPT_REFILL_MIN, RT_OPT_POOL_BURST, RT_OPT_POOL_JOB_MIN = 1, 2, 3, 4
RT_OPT_TILE_CLASSES, RT_OPT_POOL_PER_CU, RT_OPT_BVH_MAX_LEAF, RT_OPT_POOL_SLOTS, RT_OPT_BVH_TREELET, RT_OPT_BVH_WIDTH = 5, 6, 7, 8, 9, 10
RT_OPT_POOL_SHADER_WAVES = 11
RT_OPT_BVH_NODE_FORMAT = 12

RT_SCENE_EMPTY = 0
RT_SCENE_CORNELL_STAR = 1
RT_SCENE_PRESET1 = 2
RT_SCENE_CORNELL_TEAPOT = 3
RT_SCENE_TEAPOT_GRID = 4
RT_SCENE_PRESET1_TEAPOT = 5

RT_LIGHT_AMBIENT, RT_LIGHT_SPHERE, RT_LIGHT_POINT, RT_LIGHT_DIRECTIONAL = 0, 1, 2, 3
RT_LIGHT_CONE, RT_LIGHT_AREA, RT_LIGHT_TRIANGLE = 4, 5, 6


class RtSphere(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("r", C.c_float), ("mat", C.c_int32), ("_pad", C.c_int32 * 3)]


class RtTriangle(C.Structure):
    _fields_ = [("v0", C.c_float * 3), ("_p0", C.c_float), ("v1", C.c_float * 3), ("_p1", C.c_float),
                ("v2", C.c_float * 3), ("_p2", C.c_float), ("normal", C.c_float * 3), ("mat", C.c_int32)]


class RtVertex(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("_pad", C.c_float)]


class RtMaterial(C.Structure):
    _fields_ = [("albedo", C.c_float * 4), ("subsurface", C.c_float * 4), ("specular_tint", C.c_float * 4),
                ("emission_color", C.c_float * 4), ("roughness", C.c_float), ("metallic", C.c_float),
                ("ior", C.c_float), ("trs_weight", C.c_float)]


class RtLight(C.Structure):
    _fields_ = [("pos_angle_aux", C.c_float * 4), ("color_str", C.c_float * 4), ("type", C.c_int32),
                ("accumulated_str", C.c_float), ("_pad", C.c_int32 * 2)]


class RtMeshInfo(C.Structure):
    _fields_ = [("index_start", C.c_uint32), ("index_end", C.c_uint32), ("material", C.c_int32)]


class RtCamera(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("viewInv", C.c_float * 16), ("proj", C.c_float * 16),
                ("projInv", C.c_float * 16), ("viewproj", C.c_float * 16), ("position", C.c_float * 3),
                ("tanHalfFOV", C.c_float)]


class RtPushConstants(C.Structure):
    _fields_ = [("time", C.c_float), ("frameCount", C.c_uint32), ("total_lights", C.c_int32),
                ("lights_strength_sum", C.c_float), ("world_up", C.c_float * 3),
                ("reset_frame_accumulation", C.c_uint32), ("total_spheres", C.c_int32),
                ("total_triangles", C.c_int32), ("total_meshes", C.c_int32)]


class RtParams(C.Structure):
    _fields_ = [("spp", C.c_int32), ("max_bounces", C.c_int32), ("accel", C.c_int32), ("strip_rows", C.c_int32),
                ("strip_count", C.c_int32), ("strip_index", C.c_int32), ("flags", C.c_uint32), ("skybox", C.c_int32)]


class RtStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64),
                ("total_rays", C.c_uint64), ("cycles_raygen", C.c_uint64), ("cycles_search", C.c_uint64),
                ("cycles_shade", C.c_uint64), ("lane_iterations", C.c_uint64), ("wave_iterations", C.c_uint64),
                ("cycles_shade_parts", C.c_uint64 * 5),
                ("kernel_ms", C.c_float), ("bvh_nodes", C.c_int32), ("bvh_depth", C.c_int32),
                ("total_prims", C.c_int32), ("pipeline", C.c_int32),
                ("pool_job_cycles", C.c_uint64 * 4), ("busy_lane_iterations", C.c_uint64), ("cycles_steps", C.c_uint64),
                ("bvh_width", C.c_int32), ("frames", C.c_int32),
                ("search_queue_peak", C.c_int32), ("search_queue_size", C.c_int32),
                ("bvh_node_bytes", C.c_int32), ("reserved0", C.c_int32), ("stack_overflow_steps", C.c_uint64)]


class RtSceneView(C.Structure):
    _fields_ = [("spheres", C.POINTER(RtSphere)), ("n_spheres", C.c_int32),
                ("materials", C.POINTER(RtMaterial)), ("n_materials", C.c_int32),
                ("lights", C.POINTER(RtLight)), ("n_lights", C.c_int32),
                ("triangles", C.POINTER(RtTriangle)), ("n_triangles", C.c_int32),
                ("vertices", C.POINTER(RtVertex)), ("n_vertices", C.c_int32),
                ("indices", C.POINTER(C.c_uint32)), ("n_indices", C.c_int32),
                ("meshes", C.POINTER(RtMeshInfo)), ("n_meshes", C.c_int32),
                ("total_spheres", C.c_int32), ("total_triangles", C.c_int32), ("total_meshes", C.c_int32),
                ("total_lights", C.c_int32), ("lights_strength_sum", C.c_float)]



class RtControllerState(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("front", C.c_float * 3), ("up", C.c_float * 3),
                ("yaw", C.c_float), ("pitch", C.c_float), ("roll", C.c_float), ("fov", C.c_float),
                ("last_frame", C.c_float), ("frame_count", C.c_int32),
                ("mouse_captured", C.c_int32), ("reset_accumulation", C.c_int32),
                ("accumulation_on", C.c_int32), ("should_close", C.c_int32)]


RT_KEY_W, RT_KEY_S, RT_KEY_A, RT_KEY_D, RT_KEY_SPACE, RT_KEY_LEFT_CONTROL, RT_KEY_LEFT_SHIFT = 1, 2, 4, 8, 16, 32, 64
RT_KEY_Q, RT_KEY_E, RT_KEY_F, RT_KEY_R, RT_KEY_X, RT_KEY_RELEASE_MOUSE, RT_KEY_K = 128, 256, 512, 1024, 2048, 4096, 8192


# name -> (restype, argtypes); every symbol include/rt_abi.h declares
_P = C.c_void_p
PROTOTYPES = {
    "rt_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "rt_destroy": (C.c_int, [_P]),
    "rt_upload_scene": (C.c_int, [_P, C.POINTER(RtSphere), C.c_int32, C.POINTER(RtMaterial), C.c_int32,
                                  C.POINTER(RtLight), C.c_int32, C.POINTER(RtTriangle), C.c_int32,
                                  C.POINTER(RtVertex), C.c_int32, C.POINTER(C.c_uint32), C.c_int32,
                                  C.POINTER(RtMeshInfo), C.c_int32]),
    "rt_resize": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "rt_bind_framebuffer": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P]),
    "rt_render_frame": (C.c_int, [_P, C.POINTER(RtCamera), C.POINTER(RtPushConstants), C.POINTER(RtParams)]),
    "rt_render_frames": (C.c_int, [_P, C.POINTER(RtCamera), C.POINTER(RtPushConstants), C.c_int32, C.POINTER(RtParams)]),
    "rt_render_frame_noalloc": (C.c_int, [_P, C.POINTER(RtCamera), C.POINTER(RtPushConstants), C.POINTER(RtParams), _P, _P, _P,
                                          C.c_int32, C.c_int32]),
    "rt_resolve": (C.c_int, [_P]),
    "rt_strip_staging_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "rt_strip_pack": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "rt_strip_unpack": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "rt_group_create": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(_P)]),
    "rt_group_destroy": (C.c_int, [_P]),
    "rt_group_size": (C.c_int, [_P]),
    "rt_group_context": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "rt_group_upload_scene_from": (C.c_int, [_P, _P]),
    "rt_group_resize": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "rt_group_render_frames": (C.c_int, [_P, C.POINTER(RtCamera), C.POINTER(RtPushConstants), C.POINTER(RtParams), C.c_int, C.c_int32, C.c_int32]),
    "rt_group_render_frame": (C.c_int, [_P, C.POINTER(RtCamera), C.POINTER(RtPushConstants), C.POINTER(RtParams), C.c_int, C.c_int32]),
    "rt_group_reduce_samples": (C.c_int, [_P]),
    "rt_group_read_total": (C.c_int, [_P, _P, _P]),
    "rt_group_sync": (C.c_int, [_P]),
    "rt_group_last_error": (C.c_char_p, [_P]),
    "rt_read_output_bgra8": (C.c_int, [_P, _P]),
    "rt_read_accum": (C.c_int, [_P, _P, _P]),
    "rt_read_debug": (C.c_int, [_P, _P, _P]),
    "rt_debug_math": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int32]),
    "rt_get_stats": (C.c_int, [_P, C.POINTER(RtStats)]),
    "rt_set_option": (C.c_int, [_P, C.c_int, C.c_int]),
    "rt_get_device_buffers": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "rt_get_stream": (_P, [_P]),
    "rt_sync": (C.c_int, [_P]),
    "rt_last_error": (C.c_char_p, [_P]),
    "rt_default_params": (None, [C.POINTER(RtParams)]),
    "rt_abi_version": (C.c_int, []),
    "rt_stats_size": (C.c_size_t, []),
    "rt_scene_create": (C.c_int, [C.c_int, C.c_char_p, C.POINTER(_P)]),
    "rt_scene_destroy": (C.c_int, [_P]),
    "rt_scene_add_material": (C.c_int, [_P, C.POINTER(RtMaterial)]),
    "rt_scene_add_sphere": (C.c_int, [_P, C.POINTER(C.c_float), C.c_float, C.c_int32]),
    "rt_scene_add_triangle": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32]),
    "rt_scene_add_quad": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                    C.POINTER(C.c_float), C.c_int32]),
    "rt_scene_add_light": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32]),
    "rt_scene_add_model": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_int32]),
    "rt_scene_add_model_ex": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float,
                                        C.c_float, C.c_int32, C.c_uint32]),
    "rt_scene_view": (C.c_int, [_P, C.POINTER(RtSceneView)]),
    "rt_upload_scene_from": (C.c_int, [_P, _P]),
    "rt_camera_look": (C.c_int, [C.POINTER(RtCamera), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                 C.POINTER(C.c_float), C.c_float, C.c_int32, C.c_int32]),
    "rt_camera_default": (C.c_int, [C.POINTER(RtCamera), C.c_int32, C.c_int32]),
    "rt_push_constants": (C.c_int, [C.POINTER(RtPushConstants), _P, C.c_float, C.c_uint32, C.c_int]),
    "rt_controller_create": (C.c_int, [C.POINTER(_P)]),
    "rt_controller_destroy": (C.c_int, [_P]),
    "rt_controller_get": (C.c_int, [_P, C.POINTER(RtControllerState)]),
    "rt_controller_set_plane_mode": (C.c_int, [_P, C.c_int]),
    "rt_controller_begin_frame": (C.c_int, [_P, C.c_double, C.c_uint32, C.POINTER(C.c_float)]),
    "rt_controller_mouse_button": (C.c_int, [_P, C.c_int]),
    "rt_controller_mouse_move": (C.c_int, [_P, C.c_double, C.c_double]),
    "rt_controller_scroll": (C.c_int, [_P, C.c_double]),
    "rt_controller_frame_inputs": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.POINTER(RtCamera), C.POINTER(RtPushConstants)]),
    "rt_controller_end_frame": (C.c_int, [_P, C.c_double]),
    "rt_scene_bvh_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_float)]),
    "rt_scene_wide_bvh_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rt_scene_bvh_hash": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
}

_lib = None


def _share_torch_hip_runtime() -> None:
    """A process can drive the GPU through ONE HIP runtime.  PyTorch's wheel carries its own copy (torch/lib/libamdhip64.so,
    asked for by that file name), this library asks for the system's by its soname (libamdhip64.so.7): whichever is loaded
    first serves the other only if it was torch's, because torch's request does not match the system copy's name - and a
    second runtime in the process finds no GPU ("No HIP GPUs are available").  So where torch is installed but not
    imported yet, its copy is loaded first; bench.py and parallel.py, which use both, then work in either import order."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    """Load libraytracer_amd.so (once) and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`); there is no fallback path")
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    older_build = "RT_LIB_PATH" in os.environ  # an A/B build of another revision (tools/ab_build.sh) may lack newer entry points
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if not older_build:
                raise
            # say which entry point the other build lacks when it is called, not an AttributeError from ctypes
            def missing(*_args, _name=name):
                raise RuntimeError(f"{LIB_PATH} (RT_LIB_PATH) does not export {_name}: it was built from an older revision")
            setattr(lib, name, missing)
            continue
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
