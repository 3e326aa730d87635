"""Thin object wrappers over the C ABI (include/rt_abi.h): ``Scene`` mirrors the
reference's Scene builder surface (src/scene.hpp:8-37) and ``Context`` stands where
``RaytracingApp``'s buffer/dispatch code stood (src/main.cpp:1222-1606).  All work
happens in libraytracer_amd.so; these classes only marshal arguments.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import capi


class RtError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        super().__init__(f"{where} failed with status {code}" + (f": {detail}" if detail else ""))
        self.code = code


def _f3(v: Sequence[float]):
    return (C.c_float * 3)(*[float(x) for x in v])


def _f4(v: Sequence[float]):
    return (C.c_float * 4)(*[float(x) for x in v])


class Scene:
    """Host scene (the seven vectors + counters of src/scene.hpp:10-21)."""

    def __init__(self, preset: int = capi.RT_SCENE_CORNELL_STAR, assets_dir: Optional[str] = None):
        self._lib = capi.load()
        self._h = C.c_void_p()
        rc = self._lib.rt_scene_create(int(preset), (assets_dir or capi.ASSETS_DIR).encode(), C.byref(self._h))
        if rc != capi.RT_OK:
            if self._h:
                self._lib.rt_scene_destroy(self._h)
                self._h = C.c_void_p()
            raise RtError(rc, "rt_scene_create", "asset missing or malformed" if rc == capi.RT_ERR_IO else "")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def _chk(self, rc, where):
        if rc < 0:
            raise RtError(rc, where)
        return rc

    # builder calls: src/scene.cpp:716-846
    def add_material(self, albedo, subsurface, specular_tint, emission_color, roughness, metallic, ior, trs_weight) -> int:
        m = capi.RtMaterial(_f4(albedo), _f4(subsurface), _f4(specular_tint), _f4(emission_color), roughness, metallic,
                            ior, trs_weight)
        return self._chk(self._lib.rt_scene_add_material(self._h, C.byref(m)), "rt_scene_add_material")

    def add_sphere(self, pos, r, mat):
        self._chk(self._lib.rt_scene_add_sphere(self._h, _f3(pos), float(r), int(mat)), "rt_scene_add_sphere")

    def add_triangle(self, v0, v1, v2, mat):
        self._chk(self._lib.rt_scene_add_triangle(self._h, _f3(v0), _f3(v1), _f3(v2), int(mat)), "rt_scene_add_triangle")

    def add_quad(self, v0, v1, v2, v3, mat):
        self._chk(self._lib.rt_scene_add_quad(self._h, _f3(v0), _f3(v1), _f3(v2), _f3(v3), int(mat)), "rt_scene_add_quad")

    def add_light(self, pos_angle_aux, color_str, light_type):
        self._chk(self._lib.rt_scene_add_light(self._h, _f4(pos_angle_aux), _f4(color_str), int(light_type)),
                  "rt_scene_add_light")

    def add_model(self, file_name, pos, pitch, yaw, roll, scale, material, flags: int = 0):
        """Scene::addModel; `flags` (capi.RT_MODEL_*) select the general GLB reader and index rebasing."""
        if flags:
            self._chk(self._lib.rt_scene_add_model_ex(self._h, file_name.encode(), _f3(pos), float(pitch), float(yaw),
                                                      float(roll), float(scale), int(material), int(flags)),
                      "rt_scene_add_model_ex")
            return
        self._chk(self._lib.rt_scene_add_model(self._h, file_name.encode(), _f3(pos), float(pitch), float(yaw),
                                               float(roll), float(scale), int(material)), "rt_scene_add_model")

    def view(self) -> capi.RtSceneView:
        v = capi.RtSceneView()
        self._chk(self._lib.rt_scene_view(self._h, C.byref(v)), "rt_scene_view")
        return v

    def arrays(self) -> dict:
        """Copies of the seven vectors as numpy byte-exact structured views (for hashing / the oracle)."""
        v = self.view()

        def grab(ptr, n, size):
            if n == 0 or not ptr:
                return np.zeros(0, np.uint8)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * size,)).copy()

        return {
            "spheres": grab(v.spheres, v.n_spheres, 32), "materials": grab(v.materials, v.n_materials, 80),
            "lights": grab(v.lights, v.n_lights, 48), "triangles": grab(v.triangles, v.n_triangles, 64),
            "vertices": grab(v.vertices, v.n_vertices, 16), "indices": grab(v.indices, v.n_indices, 4),
            "meshes": grab(v.meshes, v.n_meshes, 12),
        }

    def push_constants(self, time: float = 0.0, frame_count: int = 0, reset: bool = False) -> capi.RtPushConstants:
        pc = capi.RtPushConstants()
        self._chk(self._lib.rt_push_constants(C.byref(pc), self._h, float(time), int(frame_count), 1 if reset else 0),
                  "rt_push_constants")
        return pc

    def bvh_info(self) -> dict:
        n, d, p, s = C.c_int32(), C.c_int32(), C.c_int32(), C.c_float()
        self._chk(self._lib.rt_scene_bvh_info(self._h, C.byref(n), C.byref(d), C.byref(p), C.byref(s)), "rt_scene_bvh_info")
        return {"nodes": n.value, "depth": d.value, "prims": p.value, "sah_cost": s.value}

    def wide_bvh_info(self) -> dict:
        n, d, k = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self._lib.rt_scene_wide_bvh_info(self._h, C.byref(n), C.byref(d), C.byref(k)), "rt_scene_wide_bvh_info")
        return {"nodes": n.value, "depth": d.value, "stack_need": k.value}

    def bvh_hash(self) -> int:
        h = C.c_uint64()
        self._chk(self._lib.rt_scene_bvh_hash(self._h, C.byref(h)), "rt_scene_bvh_hash")
        return h.value


def default_camera(width: int, height: int) -> capi.RtCamera:
    cam = capi.RtCamera()
    rc = capi.load().rt_camera_default(C.byref(cam), int(width), int(height))
    if rc != capi.RT_OK:
        raise RtError(rc, "rt_camera_default")
    return cam


def look_camera(pos, front, up, fov_degrees: float, width: int, height: int) -> capi.RtCamera:
    cam = capi.RtCamera()
    rc = capi.load().rt_camera_look(C.byref(cam), _f3(pos), _f3(front), _f3(up), float(fov_degrees), int(width), int(height))
    if rc != capi.RT_OK:
        raise RtError(rc, "rt_camera_look")
    return cam


class Controller:
    """The application's camera controller and frame seeding policy (RaytracingApp's camera state,
    processInput, mouse callbacks, updateCamera, showFPS: src/main.cpp:250-273, 283-288, 402-565,
    1424-1442, 2156-2174) driven by explicit events instead of a window."""

    def __init__(self):
        self._lib = capi.load()
        self._h = C.c_void_p()
        rc = self._lib.rt_controller_create(C.byref(self._h))
        if rc != capi.RT_OK:
            raise RtError(rc, "rt_controller_create")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.rt_controller_destroy(h)

    def _chk(self, rc, what):
        if rc != capi.RT_OK:
            raise RtError(rc, what)

    @property
    def state(self) -> capi.RtControllerState:
        st = capi.RtControllerState()
        self._chk(self._lib.rt_controller_get(self._h, C.byref(st)), "rt_controller_get")
        return st

    def set_plane_mode(self, on: bool) -> None:
        self._chk(self._lib.rt_controller_set_plane_mode(self._h, 1 if on else 0), "rt_controller_set_plane_mode")

    def begin_frame(self, now_seconds: float, keys: int = 0) -> float:
        dt = C.c_float()
        self._chk(self._lib.rt_controller_begin_frame(self._h, float(now_seconds), int(keys), C.byref(dt)), "rt_controller_begin_frame")
        return dt.value

    def mouse_button(self, left_pressed: bool = True) -> None:
        self._chk(self._lib.rt_controller_mouse_button(self._h, 1 if left_pressed else 0), "rt_controller_mouse_button")

    def mouse_move(self, x: float, y: float) -> None:
        self._chk(self._lib.rt_controller_mouse_move(self._h, float(x), float(y)), "rt_controller_mouse_move")

    def scroll(self, yoffset: float) -> None:
        self._chk(self._lib.rt_controller_scroll(self._h, float(yoffset)), "rt_controller_scroll")

    def frame_inputs(self, scene: "Scene", width: int, height: int):
        cam, pc = capi.RtCamera(), capi.RtPushConstants()
        self._chk(self._lib.rt_controller_frame_inputs(self._h, scene._h, int(width), int(height), C.byref(cam), C.byref(pc)),
                  "rt_controller_frame_inputs")
        return cam, pc

    def end_frame(self, now_seconds: float) -> None:
        self._chk(self._lib.rt_controller_end_frame(self._h, float(now_seconds)), "rt_controller_end_frame")


def make_params(spp: int = 5, max_bounces: int = 20, accel: int = capi.RT_ACCEL_BVH, strip_rows: int = 0,
                strip_count: int = 1, strip_index: int = 0, flags: int = 0, skybox: int = capi.RT_SKY_GREY) -> capi.RtParams:
    return capi.RtParams(int(spp), int(max_bounces), int(accel), int(strip_rows), int(strip_count), int(strip_index),
                         int(flags), int(skybox))


class Context:
    """One GPU context: device scene buffers, framebuffer, stream."""

    def __init__(self, device: int = 0):
        self._lib = capi.load()
        self._h = C.c_void_p()
        rc = self._lib.rt_create(int(device), C.byref(self._h))
        if rc != capi.RT_OK:
            raise RtError(rc, "rt_create", (self._lib.rt_last_error(None) or b"").decode())
        self.width = self.height = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, where):
        if rc != capi.RT_OK:
            raise RtError(rc, where, (self._lib.rt_last_error(self._h) or b"").decode())

    @property
    def handle(self):
        return self._h

    def upload_scene(self, scene: Scene):
        self._chk(self._lib.rt_upload_scene_from(self._h, scene.handle), "rt_upload_scene_from")

    def upload_arrays(self, arrays: dict, n_spheres: int, n_lights: int, n_triangles: int, n_meshes: int):
        """rt_upload_scene with explicit element counts on byte arrays in the shader's buffer layouts
        (as Scene.arrays() returns them): for scenes the builder calls cannot express."""
        a = {k: np.ascontiguousarray(v, dtype=np.uint8) for k, v in arrays.items()}
        self._keep = a

        def ptr(name, ctype):
            return C.cast(a[name].ctypes.data_as(C.c_void_p), C.POINTER(ctype))

        self._chk(self._lib.rt_upload_scene(
            self._h, ptr("spheres", capi.RtSphere), int(n_spheres), ptr("materials", capi.RtMaterial), a["materials"].size // 80,
            ptr("lights", capi.RtLight), int(n_lights), ptr("triangles", capi.RtTriangle), int(n_triangles),
            ptr("vertices", capi.RtVertex), a["vertices"].size // 16, ptr("indices", C.c_uint32), a["indices"].size // 4,
            ptr("meshes", capi.RtMeshInfo), int(n_meshes)), "rt_upload_scene")

    def set_option(self, option: int, value: int):
        self._chk(self._lib.rt_set_option(self._h, int(option), int(value)), "rt_set_option")

    def resize(self, width: int, height: int):
        self._chk(self._lib.rt_resize(self._h, int(width), int(height)), "rt_resize")
        self.width, self.height = int(width), int(height)

    def bind_framebuffer(self, width: int, height: int, accum_ptr: int, counts_ptr: int, bgra_ptr: int):
        self._chk(self._lib.rt_bind_framebuffer(self._h, int(width), int(height), C.c_void_p(accum_ptr),
                                                C.c_void_p(counts_ptr), C.c_void_p(bgra_ptr)), "rt_bind_framebuffer")
        self.width, self.height = int(width), int(height)

    def render_frame(self, camera: capi.RtCamera, pc: capi.RtPushConstants, params: Optional[capi.RtParams] = None):
        self._chk(self._lib.rt_render_frame(self._h, C.byref(camera), C.byref(pc), C.byref(params) if params else None),
                  "rt_render_frame")

    def render_frames(self, camera: capi.RtCamera, pcs, params: Optional[capi.RtParams] = None):
        """Consecutive frames of one camera (pcs: a sequence of RtPushConstants), fused into one launch where possible."""
        arr = (capi.RtPushConstants * len(pcs))(*pcs)
        self._chk(self._lib.rt_render_frames(self._h, C.byref(camera), arr, len(pcs), C.byref(params) if params else None),
                  "rt_render_frames")

    def render_frame_noalloc(self, camera, pc, params, accum_ptr: int, counts_ptr: int, bgra_ptr: int, width: int, height: int):
        self._chk(self._lib.rt_render_frame_noalloc(self._h, C.byref(camera), C.byref(pc), C.byref(params) if params else None,
                                                    C.c_void_p(accum_ptr), C.c_void_p(counts_ptr), C.c_void_p(bgra_ptr),
                                                    int(width), int(height)), "rt_render_frame_noalloc")

    def strip_pack(self, strip_rows: int, strip_count: int, strip_index: int, staging_ptr: int):
        self._chk(self._lib.rt_strip_pack(self._h, int(strip_rows), int(strip_count), int(strip_index), C.c_void_p(staging_ptr)),
                  "rt_strip_pack")

    def strip_unpack(self, strip_rows: int, strip_count: int, skip_index: int, gathered_ptr: int):
        self._chk(self._lib.rt_strip_unpack(self._h, int(strip_rows), int(strip_count), int(skip_index), C.c_void_p(gathered_ptr)),
                  "rt_strip_unpack")

    def resolve(self):
        self._chk(self._lib.rt_resolve(self._h), "rt_resolve")

    def sync(self):
        self._chk(self._lib.rt_sync(self._h), "rt_sync")

    def read_bgra8(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self._chk(self._lib.rt_read_output_bgra8(self._h, out.ctypes.data_as(C.c_void_p)), "rt_read_output_bgra8")
        return out

    def read_accum(self):
        acc = np.empty((self.height, self.width, 4), dtype=np.float32)
        cnt = np.empty((self.height, self.width), dtype=np.int32)
        self._chk(self._lib.rt_read_accum(self._h, acc.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p)),
                  "rt_read_accum")
        return acc, cnt

    def read_debug(self):
        ids = np.empty((self.height, self.width), dtype=np.int32)
        ph = np.empty((self.height, self.width), dtype=np.uint32)
        self._chk(self._lib.rt_read_debug(self._h, ids.ctypes.data_as(C.c_void_p), ph.ctypes.data_as(C.c_void_p)),
                  "rt_read_debug")
        return ids, ph

    def debug_math(self, fn: int, x: np.ndarray, y: Optional[np.ndarray] = None) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), dtype=np.float32)
        out = np.empty_like(x)
        self._chk(self._lib.rt_debug_math(self._h, int(fn), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p),
                                          out.ctypes.data_as(C.c_void_p), int(x.size)), "rt_debug_math")
        return out

    def stats(self) -> capi.RtStats:
        st = capi.RtStats()
        self._chk(self._lib.rt_get_stats(self._h, C.byref(st)), "rt_get_stats")
        return st

    def device_buffers(self):
        a, c, b = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(self._lib.rt_get_device_buffers(self._h, C.byref(a), C.byref(c), C.byref(b)), "rt_get_device_buffers")
        return a.value, c.value, b.value

    @property
    def stream(self) -> int:
        return int(self._lib.rt_get_stream(self._h) or 0)


class Group:
    """n devices driven from one host thread (rt_group_*): a context per device and an RCCL communicator."""

    def __init__(self, devices: Sequence[int]):
        self._lib = capi.load()
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = self._lib.rt_group_create(len(devices), arr, C.byref(self._h))
        if rc != capi.RT_OK:
            raise RtError(rc, "rt_group_create", (self._lib.rt_last_error(None) or b"").decode())
        self.width = self.height = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rt_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, where):
        if rc != capi.RT_OK:
            raise RtError(rc, where, (self._lib.rt_group_last_error(self._h) or b"").decode())

    def __len__(self):
        return int(self._lib.rt_group_size(self._h))

    def context(self, index: int) -> "Context":
        """A borrowed view of device `index`'s context (do not close it)."""
        h = C.c_void_p()
        self._chk(self._lib.rt_group_context(self._h, int(index), C.byref(h)), "rt_group_context")
        c = Context.__new__(Context)
        c._lib, c._h, c.width, c.height = self._lib, h, self.width, self.height
        c.close = lambda: None
        return c

    def upload_scene(self, scene: Scene):
        self._chk(self._lib.rt_group_upload_scene_from(self._h, scene.handle), "rt_group_upload_scene_from")

    def resize(self, width: int, height: int):
        self._chk(self._lib.rt_group_resize(self._h, int(width), int(height)), "rt_group_resize")
        self.width, self.height = int(width), int(height)

    def render_frame(self, camera, pc, params, mode: int, strip_rows: int = 0):
        self._chk(self._lib.rt_group_render_frame(self._h, C.byref(camera), C.byref(pc), C.byref(params) if params else None,
                                                  int(mode), int(strip_rows)), "rt_group_render_frame")

    def render_frames(self, camera, pc, params, mode: int, strip_rows: int, frames: int):
        self._chk(self._lib.rt_group_render_frames(self._h, C.byref(camera), C.byref(pc), C.byref(params) if params else None,
                                                   int(mode), int(strip_rows), int(frames)), "rt_group_render_frames")

    def reduce_samples(self):
        self._chk(self._lib.rt_group_reduce_samples(self._h), "rt_group_reduce_samples")

    def read_total(self):
        acc = np.empty((self.height, self.width, 4), dtype=np.float32)
        cnt = np.empty((self.height, self.width), dtype=np.int32)
        self._chk(self._lib.rt_group_read_total(self._h, acc.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p)),
                  "rt_group_read_total")
        return acc, cnt

    def sync(self):
        self._chk(self._lib.rt_group_sync(self._h), "rt_group_sync")
