"""Device arithmetic against the oracle's binary32 evaluation: bit-identical results for
every primitive the path tracer builds on (no FMA contraction, IEEE division and sqrt,
the explicit polynomial sin/cos/tan/acos/asin/log2/exp2/pow, the RNG's float conversion)."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from raytracer_amd.renderer import Context

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def oracle_eval(fn, x, y):
    lib = ob.load()
    return np.array([lib.rto_math(fn, 0, float(a), float(b)) for a, b in zip(x, y)], dtype=np.float32)


def samples(lo, hi, n, rng):
    v = rng.uniform(lo, hi, n).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, np.inf, -np.inf, np.nan, 1e-30, 1e-40, 3.1415927, 1.5707964,
                        6.2831855, 0.70710677, 2.0, 1e9, 2e9], dtype=np.float32)
    return np.concatenate([v, special])


@pytest.mark.parametrize("fn,lo,hi", [(0, -10, 10), (1, -10, 10), (2, -3.2, 3.2), (3, -1.001, 1.001), (7, -1.001, 1.001),
                                       (5, 0, 100), (6, -130, 130)])
def test_unary(ctx, fn, lo, hi):
    rng = np.random.default_rng(fn)
    x = samples(lo, hi, 20000, rng)
    y = np.zeros_like(x)
    g = ctx.debug_math(fn, x, y)
    o = oracle_eval(fn, x, y)
    nan_both = np.isnan(g) & np.isnan(o)
    same = (g.view(np.uint32) == o.view(np.uint32)) | nan_both
    assert same.all(), f"fn {fn}: {np.count_nonzero(~same)} mismatches, first x={x[~same][:5]} g={g[~same][:5]} o={o[~same][:5]}"


def test_pow(ctx):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(0, 1, 20000), rng.uniform(0, 50, 2000), [0.0, 1.0, -1.0, np.inf, np.nan]]).astype(np.float32)
    for e in (5.0, 1.0 / 2.2):
        y = np.full_like(x, np.float32(e))
        g = ctx.debug_math(4, x, y)
        o = oracle_eval(4, x, y)
        same = (g.view(np.uint32) == o.view(np.uint32)) | (np.isnan(g) & np.isnan(o))
        assert same.all(), f"pow(x,{e}): {np.count_nonzero(~same)} mismatches x={x[~same][:5]} g={g[~same][:5]} o={o[~same][:5]}"


def test_basic_ieee(ctx):
    rng = np.random.default_rng(2)
    x = rng.uniform(-100, 100, 50000).astype(np.float32)
    y = rng.uniform(-100, 100, 50000).astype(np.float32)
    assert np.array_equal(ctx.debug_math(8, x, y).view(np.uint32), (x / y).view(np.uint32)), "division is not correctly rounded"
    ax = np.abs(x)
    assert np.array_equal(ctx.debug_math(9, ax, y).view(np.uint32), np.sqrt(ax).view(np.uint32)), "sqrt is not correctly rounded"
    two_step = (x * y).astype(np.float32) + x
    assert np.array_equal(ctx.debug_math(11, x, y).view(np.uint32), two_step.view(np.uint32)), "x*y+x was contracted to an FMA"
    inv = np.float32(1.0) / np.sqrt((x * x).astype(np.float32) + (y * y).astype(np.float32))
    assert np.array_equal(ctx.debug_math(12, x, y).view(np.uint32), inv.view(np.uint32))


def test_rng_float(ctx):
    lib = ob.load()
    states = np.array([1, 2, 0xffffffff, 0x80000000, 12345, 0xdeadbeef, 0x7fffffff], dtype=np.uint32)
    g = ctx.debug_math(10, states.view(np.float32))
    o = np.array([lib.rto_random_from(int(s)) for s in states], dtype=np.float32)
    assert np.array_equal(g.view(np.uint32), o.view(np.uint32))


def test_short_reciprocal_and_root_against_the_host(ctx):
    """g_rcp / g_sqrt / g_rcp_sqrt (hardware estimate + one fused Newton round inside a range,
    the compiler's IEEE expansion outside) on random bit patterns of every exponent and on the
    special values: the host's correctly rounded 1/x, sqrt(x), 1/sqrt(x)."""
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 1 << 32, 300000, dtype=np.uint64).astype(np.uint32)
    x = np.concatenate([bits.view(np.float32), samples(0, 1, 10, rng),
                        np.array([2.0 ** -126, 2.0 ** -127, 2.0 ** 126, 2.0 ** 127, 2.0 ** -101, 2.0 ** -102, 3.0e38], dtype=np.float32)])
    y = np.zeros_like(x)
    with np.errstate(all="ignore"):
        want = {13: np.float32(1.0) / x, 9: np.sqrt(x), 14: np.float32(1.0) / np.sqrt(x)}
    for fn, w in want.items():
        g = ctx.debug_math(fn, x, y)
        same = (g.view(np.uint32) == w.view(np.uint32)) | (np.isnan(g) & np.isnan(w))
        assert same.all(), f"fn {fn}: {np.count_nonzero(~same)} mismatches, x={x[~same][:5]} g={g[~same][:5]} want={w[~same][:5]}"


@pytest.mark.parametrize("fn", [100, 101, 102])
def test_short_forms_equal_the_ieee_expansions_on_every_bit_pattern(ctx, fn):
    """All 2^32 inputs on the device: the short forms and the compiler's IEEE expansions agree
    (element i of the probe counts the disagreements on patterns i, i + n, ...)."""
    n = 1 << 20
    z = np.zeros(n, dtype=np.float32)
    bad = ctx.debug_math(fn, z, z)
    assert float(bad.sum()) == 0.0, f"fn {fn}: {int(bad.sum())} bit patterns differ"


def test_the_sweep_sees_differences(ctx):
    """Control for the sweeps above: without its range guard the short reciprocal differs from 1/x on
    every subnormal, on every |x| >= 2^126 (subnormal reciprocals) and on the infinities."""
    n = 1 << 20
    z = np.zeros(n, dtype=np.float32)
    bad = ctx.debug_math(103, z, z).astype(np.float64).sum()
    subnormals = 2 * ((1 << 23) - 1)      # both signs, zero excluded (1/0 = inf either way)
    huge = 2 * 2 * (1 << 23)              # biased exponents 253 and 254
    assert subnormals + huge <= bad <= subnormals + huge + 4, bad
