"""Pins the CPU oracle with hand-derived known answers.  The reference ships no tests,
golden images or vectors, so these (and tests/golden/) are what pins the restatement of
shaders/raytracer.comp.  Everything here is derived independently of the oracle's code:
integer arithmetic re-done with Python ints, geometry solved by hand, float64 references.
"""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_binding as ob
from raytracer_amd import capi
from raytracer_amd.renderer import Scene

M32 = 0xFFFFFFFF


def py_hash(x):  # raytracer.comp:193-200 in Python integers
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return x


def py_xorshift(s):  # raytracer.comp:202-207
    s ^= (s << 13) & M32
    s ^= s >> 17
    s ^= (s << 5) & M32
    return s


def test_hash_and_xorshift_known_values():
    lib = ob.load()
    assert lib.rto_hash(0) == 0  # hash(0) = 0: the seed-0 quirk of pixel (0,0) at time < 1, frame 0
    for x in (1, 2, 1920, 0xDEADBEEF, 0xFFFFFFFF, 123456789):
        assert lib.rto_hash(x) == py_hash(x)
    # xorshift32 (13,17,5) from state 1: the published sequence
    seq = [270369, 67634689, 2647435461, 307599695]
    s = 1
    for want in seq:
        mine = py_xorshift(s)  # the same step in Python integers
        s = lib.rto_xorshift(s)
        assert s == want and mine == want
    assert lib.rto_xorshift(0) == 0  # state 0 is a fixed point


def test_random_is_state_over_two_pow_32_and_can_reach_one():
    lib = ob.load()
    for s in (1, 12345, 0x80000000, 0xCAFEBABE):
        nxt = py_xorshift(s)
        want = np.float32(np.float32(nxt) / np.float32(4294967296.0))
        assert np.float32(lib.rto_random_from(s)) == want
    # xorshift is a bijection: invert it to get the state whose successor is 0xFFFFFFFF; as
    # binary32 that successor rounds up to 2^32, so random() == 1.0
    def unshift_left(y, k):
        x = y
        for _ in range(32 // k + 1):
            x = y ^ ((x << k) & M32)
        return x

    def unshift_right(y, k):
        x = y
        for _ in range(32 // k + 1):
            x = y ^ (x >> k)
        return x

    found = unshift_left(unshift_right(unshift_left(0xFFFFFFFF, 5), 17), 13)
    assert py_xorshift(found) == 0xFFFFFFFF
    assert lib.rto_random_from(found) == 1.0
    assert lib.rto_random_from(0) == 0.0


def test_pixel_seed_formula_and_its_quirks():
    lib = ob.load()
    assert lib.rto_pixel_seed(0.0, 0, 0, 0) == 0  # RNG stuck at 0 for this pixel
    assert lib.rto_pixel_seed(0.99, 0, 0, 0) == 0  # uint(time) truncates
    for (t, f, x, y) in ((0.0, 0, 5, 7), (3.7, 11, 100, 200), (12.0, 59, 1919, 1079)):
        want = py_hash((int(t) * 1920) & M32) ^ py_hash(f) ^ py_hash((x + y * 1920) & M32)
        assert lib.rto_pixel_seed(t, f, x, y) == want
    # stride 1920 is hard-coded: pixel (1920+k, y) shares its seed with (k, y+1)
    assert lib.rto_pixel_seed(0.0, 3, 1920 + 5, 10) == lib.rto_pixel_seed(0.0, 3, 5, 11)


def _mini_scene():
    s = Scene(capi.RT_SCENE_EMPTY)
    m = s.add_material((1, 1, 1, 1), (0, 0, 0, 0), (1, 1, 1, 0.5), (0, 0, 0, 0), 1.0, 0.0, 1.5, 0.0)
    return s, m


def _hit(scene, orig, dir):
    lib = ob.load()
    osc = ob.OracleScene(scene.arrays())
    pc = scene.push_constants(0.0, 0, True)
    t = C.c_float(0)
    o = (C.c_float * 3)(*orig)
    d = (C.c_float * 3)(*dir)
    pid = lib.rto_hit_scene(C.byref(osc.c), C.addressof(pc), o, d, C.byref(t))
    return pid, t.value


def test_ray_triangle_known_answer():
    s, m = _mini_scene()
    s.add_triangle((-1, -1, 0), (1, -1, 0), (0, 1, 0), m)
    pid, t = _hit(s, (0, 0, -1), (0, 0, 1))
    assert pid == 0 and t == 1.0
    pid, t = _hit(s, (0, 0, -3), (0, 0, 2))  # direction need not be unit: t is in units of |dir|
    assert pid == 0 and t == 1.5
    assert _hit(s, (5, 5, -1), (0, 0, 1))[0] == -1  # outside
    assert _hit(s, (0, 0, -1), (1, 0, 0))[0] == -1  # parallel: |a| < 1e-6
    assert _hit(s, (0, 0, 0.004), (0, 0, -1))[0] == -1  # t = 0.004 is not inside (0.005, inf)
    pid, t = _hit(s, (0, 0, 1), (0, 0, -1))  # two-sided
    assert pid == 0 and t == 1.0


def test_sphere_roots_known_answer():
    s, m = _mini_scene()
    s.add_sphere((0, 0, 5), 1.0, m)
    pid, t = _hit(s, (0, 0, 0), (0, 0, 1))
    assert pid == 0 and t == 4.0
    pid, t = _hit(s, (0, 0, 5), (0, 0, 1))  # from the centre: near root negative, far root taken
    assert pid == 0 and t == 1.0
    assert _hit(s, (0, 2, 0), (0, 0, 1))[0] == -1


def test_tie_rules_first_triangle_wins_last_mesh_triangle_wins():
    # two identical scene triangles: strict `closest_so_far > t` keeps the FIRST (raytracer.comp:489)
    s, m = _mini_scene()
    s.add_triangle((-1, -1, 0), (1, -1, 0), (0, 1, 0), m)
    s.add_triangle((-1, -1, 0), (1, -1, 0), (0, 1, 0), m)
    assert _hit(s, (0, 0, -1), (0, 0, 1)) == (0, 1.0)
    # the same pair inside ONE mesh: `t_r > closest_so_far` rejects only larger t, so the LAST wins (447)
    arrays = s.arrays()
    verts = np.zeros((3, 4), np.float32)
    verts[:, :3] = [(-1, -1, 0), (1, -1, 0), (0, 1, 0)]
    arrays["vertices"] = verts.view(np.uint8).reshape(-1)
    arrays["indices"] = np.array([0, 1, 2, 0, 1, 2], np.uint32).view(np.uint8)
    mesh = np.zeros(3, np.uint32)
    mesh[:] = (0, 6, 0)
    arrays["meshes"] = mesh.view(np.uint8)
    osc = ob.OracleScene(arrays)
    pc = s.push_constants(0.0, 0, True)
    pc.total_triangles = 0
    pc.total_meshes = 1
    lib = ob.load()
    t = C.c_float(0)
    pid = lib.rto_hit_scene(C.byref(osc.c), C.addressof(pc), (C.c_float * 3)(0, 0, -1), (C.c_float * 3)(0, 0, 1), C.byref(t))
    assert (pid, t.value) == (1, 1.0)  # S + T + index/3 with S = T = 0: the second mesh triangle
    # a scene triangle and a mesh triangle at the same t: the earlier loop (triangles) wins (500)
    pc.total_triangles = 2
    pid = lib.rto_hit_scene(C.byref(osc.c), C.addressof(pc), (C.c_float * 3)(0, 0, -1), (C.c_float * 3)(0, 0, 1), C.byref(t))
    assert pid == 0


@pytest.mark.parametrize("fn,ref,lo,hi,max_ulp", [
    (0, np.sin, -6.3, 6.3, 2.0), (1, np.cos, -6.3, 6.3, 2.0), (3, np.arccos, -1.0, 1.0, 3.0),
    (5, np.log2, 1e-6, 100.0, 3.0), (6, np.exp2, -20.0, 20.0, 2.0)])
def test_polynomial_math_accuracy_vs_float64(fn, ref, lo, hi, max_ulp):
    """The explicit binary32 evaluations stay within a few ulp of the true function, i.e. well
    inside the precision Vulkan allows a driver for these GLSL built-ins."""
    lib = ob.load()
    rng = np.random.default_rng(7)
    x = rng.uniform(lo, hi, 4000).astype(np.float32)
    got = np.array([lib.rto_math(fn, 0, float(v), 0.0) for v in x], np.float64)
    want = ref(x.astype(np.float64))
    ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    if fn in (0, 1):  # near zeros of sin/cos measure absolute error against the argument's own rounding
        ulp = np.maximum(ulp, np.spacing(np.float32(1.0)) * 0.5)
    if fn == 5:       # near log2(1) = 0 likewise
        ulp = np.maximum(ulp, np.spacing(np.float32(0.25)))
    assert np.max(np.abs(got - want) / ulp) <= max_ulp


def test_tan_of_acos_and_pow_accuracy():
    lib = ob.load()
    rng = np.random.default_rng(8)
    c = rng.uniform(0.01, 0.999, 3000).astype(np.float32)
    got = np.array([lib.rto_math(2, 0, lib.rto_math(3, 0, float(v), 0.0), 0.0) for v in c], np.float64)
    want = np.tan(np.arccos(c.astype(np.float64)))
    assert np.max(np.abs(got - want) / np.abs(want)) < 2e-5  # tan(acos(c)) is ill-conditioned near c -> 1
    x = rng.uniform(0.0, 1.0, 3000).astype(np.float32)
    for e in (5.0, 1.0 / 2.2):
        got = np.array([lib.rto_math(4, 0, float(v), e) for v in x], np.float64)
        want = x.astype(np.float64) ** np.float64(np.float32(e))
        # exp2(y*log2(x)): the exponent carries |y*log2 x| * 2^-24 of error, as GLSL allows for pow
        assert np.max(np.abs(got - want) / np.maximum(want, 1e-30)) < 1e-5
    assert lib.rto_math(4, 0, 0.0, 5.0) == 0.0 and lib.rto_math(4, 0, 1.0, 5.0) == 1.0
    assert math.isnan(lib.rto_math(4, 0, -0.5, 5.0))  # GLSL: pow undefined for x < 0 -> NaN via log2
    assert math.isnan(lib.rto_math(3, 0, 1.0000001, 0.0))


# ------------------------------------------------------------------ skybox variants (raytracer.comp:520-577)
def _sky_image(skybox, front=(0.0, 0.0, 1.0), W=32, H=18):
    from helpers import oracle_frames
    from raytracer_amd import capi
    from raytracer_amd.renderer import Scene, look_camera
    s = Scene(capi.RT_SCENE_EMPTY)
    cam = look_camera((0.0, 0.0, 0.0), front, (0.0, 1.0, 0.0), 50.0, W, H)
    return oracle_frames(s, W, H, spp=1, max_bounces=1, camera=cam, skybox=skybox)["accum"]


def test_constant_skyboxes():
    """Empty scene, one sample: every pixel is gamma(sky) (raytracer.comp:978-988)."""
    import numpy as np
    from raytracer_amd import capi
    assert (_sky_image(capi.RT_SKY_WHITE)[..., :3] == 1.0).all()
    assert (_sky_image(capi.RT_SKY_BLACK)[..., :3] == 0.0).all()
    grey = _sky_image(capi.RT_SKY_GREY)[..., :3]
    assert np.allclose(grey, 0.3 ** (1 / 2.2), atol=2e-6) and (grey == grey[0, 0, 0]).all()


def test_day_sky_gradient_and_sun_disc():
    """skybox_color_day (522-540): mix(horizon, zenith, 0.5*(dir.y+1)) - whiter towards the zenith, the
    blue horizon colour's ordering of channels below it - and a white disc around the sun direction."""
    import numpy as np
    from raytracer_amd import capi
    img = _sky_image(capi.RT_SKY_DAY)[..., :3]
    col = img[:, 16]                          # one image column, top row first
    assert (np.diff(col[:, 0]) < 0).all()     # red falls from the top of the image to the bottom
    assert (col[:, 0] < col[:, 1]).all() and (col[:, 1] < col[:, 2]).all()  # 0.231 < 0.756 < 0.945 mixed with white
    a = 0.5 * (0.0 + 1.0)                      # the centre rows look along the horizon: dir.y ~ 0
    mid = (img[8, 16] + img[9, 16]) / 2
    expect = (np.array([0.231, 0.756, 0.945]) * (1 - a) + a) ** (1 / 2.2)
    assert np.allclose(mid, expect, atol=0.01)
    sun = _sky_image(capi.RT_SKY_DAY, front=(-0.33, 0.67, -0.67), W=33, H=33)[..., :3]
    assert (sun[16, 16] == 1.0).all() and sun[0, 0, 0] < 1.0


def test_night_sky_is_dark_with_a_moon():
    """skybox_color_night (543-561): zenith 0.005, horizon 0, moon colour (0.9, 0.9, 0.8) inside the disc."""
    import numpy as np
    from raytracer_amd import capi
    img = _sky_image(capi.RT_SKY_NIGHT)[..., :3]
    assert img.max() <= 0.005 ** (1 / 2.2) + 1e-6 and (np.diff(img[:, 5, 0]) < 0).all()
    moon = _sky_image(capi.RT_SKY_NIGHT, front=(0.33, 0.67, -0.67), W=33, H=33)[..., :3]
    assert np.allclose(moon[16, 16], np.array([0.9, 0.9, 0.8]) ** (1 / 2.2), atol=2e-6)
