"""Direct sweeps of the device-side math primitives (halide_amd/csrc/hlmi_device_math.h) against the oracle's restatement
(oracle/oracle_common.h), bit for bit, over the whole float range — SURVEY.md §8 row a10.  In the pipelines these functions only
ever see the remap table's, the tone curve's and nl_means' operand ranges; a transcription slip outside those would pass every
pipeline test.  Reference: /root/reference/src/IROperator.cpp:847-966 (halide_log / halide_exp), :1616-1643 (fast_exp),
src/CodeGen_LLVM.cpp:3925-3941 (pow's select chain); semantics pinned on the oracle side by tests/test_oracle_primitives.py."""
import ctypes as C

import numpy as np
import pytest

N = 1 << 24


def _special():
    f = np.float32
    v = [0.0, -0.0, 1.0, -1.0, 2.0, 0.5, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.1754944e-38, 1.1754942e-38, 3.4028235e38,
         -3.4028235e38, 88.72284, 88.72283, -87.33655, -87.33654, -103.97208, 0.6931472, 127.0, -126.0, 255.0, 3.0, -3.0, 7.0]
    return np.array(v, f)


def _all_exponents(rng, n):
    """n floats whose bit patterns are uniform over all of float32: every exponent, denormals, both signs, NaNs and infinities"""
    return rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)


def _device(hl, fn, x, y=None, z=None):
    f = hl.lib.hlmi_debug_math
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    ptr = lambda a: None if a is None else np.ascontiguousarray(a, np.float32).ctypes.data
    ya = None if y is None else np.ascontiguousarray(y, np.float32)
    za = None if z is None else np.ascontiguousarray(z, np.float32)
    assert f(fn, x.ctypes.data, ptr(ya), ptr(za), out.ctypes.data, x.size) == 0
    return out


def _oracle_vec(oracle, name, *arrays):
    fn = getattr(oracle._lib, name)
    fn.restype = None
    fn.argtypes = [C.c_void_p] * (len(arrays) + 1) + [C.c_size_t]
    arrays = [np.ascontiguousarray(a, np.float32) for a in arrays]
    out = np.empty_like(arrays[0])
    fn(*[a.ctypes.data for a in arrays], out.ctypes.data, arrays[0].size)
    return out


def _int_cast_defined(t):
    """halide_exp / fast_exp convert floor(t / ln 2) to int32 (src/IROperator.cpp:929, :1621).  Outside int32's range — and for
    NaN — that conversion is undefined in the reference's IR (LLVM fptosi: poison; x86 happens to give INT_MIN, gfx950 saturates),
    so the sweep holds the two sides to each other only where it is defined: finite t with |t / ln 2| < 2^31."""
    with np.errstate(invalid="ignore", over="ignore"):
        return np.isfinite(t) & (np.abs(t.astype(np.float64)) < 1.4e9)


def _same(got, want):
    g, w = got.view(np.uint32), want.view(np.uint32)
    nan = np.isnan(got) & np.isnan(want)          # any NaN equals any NaN (payloads are not part of the contract)
    bad = (g != w) & ~nan
    return int(np.count_nonzero(bad)), bad


@pytest.mark.gpu
@pytest.mark.parametrize("fn,name", [(0, "oracle_halide_exp_v"), (1, "oracle_halide_log_v"), (3, "oracle_fast_exp_v")])
def test_unary_primitives_over_the_whole_float_range(hl, oracle, fn, name):
    rng = np.random.default_rng(100 + fn)
    dense = {0: rng.uniform(-110.0, 95.0, N // 2), 1: np.exp(rng.uniform(-104.0, 89.0, N // 2)), 3: rng.uniform(-110.0, 95.0, N // 2)}[fn]
    with np.errstate(over="ignore"):   # the top of log's dense range rounds to +inf in float32: a wanted special value
        x = np.concatenate([_special(), _all_exponents(rng, N // 2), dense.astype(np.float32)])
    if fn != 1:
        x = x[_int_cast_defined(x)]
    got, want = _device(hl, fn, x), _oracle_vec(oracle, name, x)
    n, bad = _same(got, want)
    assert n == 0, (n, x[bad][:8], got[bad][:8], want[bad][:8])


@pytest.mark.gpu
def test_pow_select_chain_and_magnitudes(hl, oracle):
    """x > 0: exp(log(x) y); y == 0 -> 1; x == 0 -> 0; negative base: sign by the parity of an integer exponent, NaN otherwise."""
    rng = np.random.default_rng(7)
    sx, sy = _special(), _special()
    gx, gy = np.meshgrid(sx, sy)
    xs = [gx.ravel(), _all_exponents(rng, N // 4), rng.uniform(0.0, 1.0, N // 4).astype(np.float32),        # the tone curve's domain
          -rng.uniform(0.0, 8.0, N // 4).astype(np.float32), rng.uniform(0.0, 40.0, N // 4).astype(np.float32)]
    ys = [gy.ravel(), _all_exponents(rng, N // 4), rng.uniform(0.2, 4.0, N // 4).astype(np.float32),
          rng.integers(-9, 10, N // 4).astype(np.float32), rng.uniform(-6.0, 6.0, N // 4).astype(np.float32)]
    ys[3][::3] += np.float32(0.5)                                                                         # a third of them non-integers
    x, y = np.concatenate(xs).astype(np.float32), np.concatenate(ys).astype(np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        t = _oracle_vec(oracle, "oracle_halide_log_v", np.abs(x)) * y      # the argument pow hands to exp
    keep = _int_cast_defined(t)
    x, y = x[keep], y[keep]
    assert x.size > N // 2
    got, want = _device(hl, 2, x, y), _oracle_vec(oracle, "oracle_halide_pow_v", x, y)
    n, bad = _same(got, want)
    assert n == 0, (n, x[bad][:8], y[bad][:8], got[bad][:8], want[bad][:8])


@pytest.mark.gpu
def test_lerp_in_the_librarys_canonical_form(hl, oracle):
    rng = np.random.default_rng(9)
    a, b = rng.uniform(-4, 4, N // 4).astype(np.float32), rng.uniform(-4, 4, N // 4).astype(np.float32)
    w = rng.uniform(0, 1, N // 4).astype(np.float32)
    w[:1024] = np.repeat(np.array([0.0, 0.25, 0.5, 0.75, 1.0, 0.125, 0.875, 0.375], np.float32), 128)
    got, want = _device(hl, 4, a, b, w), _oracle_vec(oracle, "oracle_lerp_v", a, b, w)
    n, bad = _same(got, want)
    assert n == 0, (n, hl.canon_fma(), oracle.get_canon())


def test_vector_entry_points_of_the_oracle_equal_the_scalar_ones(oracle, each_canon):
    """CPU: the array forms the sweeps above use are the scalar restatements, element by element."""
    x = np.concatenate([_special(), np.linspace(-100, 90, 997).astype(np.float32)])
    for name, scalar in (("oracle_halide_exp_v", oracle.halide_exp), ("oracle_halide_log_v", oracle.halide_log), ("oracle_fast_exp_v", oracle.fast_exp)):
        got = _oracle_vec(oracle, name, x)
        want = np.array([scalar(float(v)) for v in x], np.float32)
        assert _same(got, want)[0] == 0, name
    y = np.resize(np.array([2.0, 0.5, 0.0, -3.0, 2.5], np.float32), x.size)
    got = _oracle_vec(oracle, "oracle_halide_pow_v", x, y)
    want = np.array([oracle.halide_pow(float(a), float(b)) for a, b in zip(x, y)], np.float32)
    assert _same(got, want)[0] == 0
