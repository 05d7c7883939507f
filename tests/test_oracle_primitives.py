"""Pin the oracle's arithmetic primitives against the known-answer bounds the reference's own tests
state for them (paths relative to /root/reference):
  test/correctness/math.cpp:28-47,328-334      exp on [0,20], log on [1,1e6], pow on [-10,10]x[-4,4]:
                                               |err| < 1e-4 or relative error < 1.25e-6 vs libm
  test/correctness/vector_math.cpp:566-648     mantissa error vs libm: log <= 8, exp <= 32, pow <= 64,
                                               fast_exp <= 64 for a in [0.5, 8.5), b likewise
"""
import math

import numpy as np


def _rel_equal(a, b):
    if a == b or (math.isnan(a) and math.isnan(b)):
        return True
    if abs(b - a) < 1e-4:
        return True
    rel = abs((b - a) / a) if abs(a) > abs(b) else abs((b - a) / b)
    return rel < 1.25e-6


def _mantissa(x):
    return int(np.float32(x).view(np.uint32)) & 0x007FFFFF


def test_exp_log_pow_math_cpp_ranges(oracle):
    # math.cpp samples 256 points uniformly over each range
    for i in range(256):
        x = np.float32(0 + 20 * i / 256)
        assert _rel_equal(oracle.halide_exp(float(x)), float(np.exp(np.float32(x), dtype=np.float32)))
        x = np.float32(1 + (1000000 - 1) * i / 256)
        assert _rel_equal(oracle.halide_log(float(x)), float(np.log(x, dtype=np.float32)))
    for i in range(256):
        x = np.float32(-10 + 20 * i / 256)
        y = np.float32(-4 + 8 * i / 256)
        with np.errstate(all="ignore"):
            want = float(np.power(x, y, dtype=np.float32))
        got = oracle.halide_pow(float(x), float(y))
        assert _rel_equal(got, want), (x, y, got, want)


def test_mantissa_error_bounds_vector_math_cpp(oracle):
    rng = np.random.default_rng(0)
    vals = (rng.random(320 * 16) * 0.0625 * 256 + 1.0).astype(np.float32)  # A(dis*0.0625+1.0) scaled over u8..float
    worst = dict(log=0, exp=0, pow=0, fast_exp=0)
    for i in range(len(vals) - 1):
        a = np.float32(vals[i] * np.float32(0.5))
        b = np.float32(vals[i + 1] * np.float32(0.5))
        worst["log"] = max(worst["log"], abs(_mantissa(oracle.halide_log(float(a))) - _mantissa(np.log(a))))
        ce = np.exp(b, dtype=np.float32)
        if np.isfinite(ce):
            worst["exp"] = max(worst["exp"], abs(_mantissa(oracle.halide_exp(float(b))) - _mantissa(ce)))
            worst["fast_exp"] = max(worst["fast_exp"], abs(_mantissa(oracle.fast_exp(float(b))) - _mantissa(ce)))
        cp = np.power(a, np.float32(b / np.float32(16.0)), dtype=np.float32)
        worst["pow"] = max(worst["pow"],
                           abs(_mantissa(oracle.halide_pow(float(a), float(b / np.float32(16.0)))) - _mantissa(cp)))
    assert worst["log"] <= 8, worst
    assert worst["exp"] <= 32, worst
    assert worst["pow"] <= 64, worst
    assert worst["fast_exp"] <= 64, worst


def test_constants_match_their_defining_expressions():
    # halide_exp: one_over_ln2 = 1.0f / logf(2.0f) (src/IROperator.cpp:927); fast_exp folds 1/logf(2.0)
    ln2 = np.log(np.float32(2.0), dtype=np.float32)
    assert ln2.view(np.uint32) == 0x3F317218
    assert (np.float32(1.0) / ln2).view(np.uint32) == 0x3FB8AA3B
    assert np.float32(1.0 / float(ln2)).view(np.uint32) == 0x3FB8AA3B
    # x / 65535.0f -> x * fold(1/65535.0f): float fold and double fold agree
    assert (np.float32(1.0) / np.float32(65535.0)).view(np.uint32) == np.float32(1.0 / 65535.0).view(np.uint32)


def test_remap_lut_shape_and_symmetry(oracle):
    lut = oracle.ll_remap_lut(8, 1.0 / 7.0)
    assert lut.shape == (3585,)
    assert lut[1792] == 0.0
    assert np.array_equal(lut[1793:], -lut[:1792][::-1])  # odd function, evaluated with the same |fx|
    # against a float64 evaluation of alpha*fx*exp(-fx^2/2)
    fx = (np.arange(-1792, 1793) / 256.0)
    ref = (1.0 / 7.0) * fx * np.exp(-fx * fx / 2)
    assert np.max(np.abs(lut - ref)) < 2e-7
