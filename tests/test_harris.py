"""harris: Harris corner response, f32 planar -> f32 (reference: /root/reference/apps/harris/harris_generator.cpp:7-62;
driver shape filter.cpp:24-26).  Sums left to right as written; GPU == oracle bit for bit."""
import numpy as np
import pytest


def _img(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = 0.5 + 0.3 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 0.2 * ((xx // 16 + yy // 16) % 2)
    return np.clip(np.stack([base, base * 0.9, base[::-1] * 0.8]) + rng.normal(0, 0.02, (3, h, w)), 0, 1).astype(np.float32)


def test_oracle_against_float64_reference(oracle):
    inp = _img(40, 30, 1)
    got = oracle.harris(inp)
    g = 0.299 * inp[0].astype(np.float64) + 0.587 * inp[1] + 0.114 * inp[2]
    iy = (-g[:-2, :-2] + g[2:, :-2] - 2 * g[:-2, 1:-1] + 2 * g[2:, 1:-1] - g[:-2, 2:] + g[2:, 2:]) / 12
    ix = (-g[:-2, :-2] + g[:-2, 2:] - 2 * g[1:-1, :-2] + 2 * g[1:-1, 2:] - g[2:, :-2] + g[2:, 2:]) / 12

    def s3(f):
        return sum(f[dy:f.shape[0] - 2 + dy, dx:f.shape[1] - 2 + dx] for dy in range(3) for dx in range(3))
    sxx, syy, sxy = s3(ix * ix), s3(iy * iy), s3(ix * iy)
    ref = (sxx * syy - sxy * sxy - 0.04 * (sxx + syy) ** 2)[1:-1, 1:-1]     # output starts at (3, 3)
    assert got.shape == ref.shape and np.max(np.abs(got - ref)) < 1e-6


def _run(hl, inp, out_min=(3, 3), out_size=None, in_min=None):
    a = hl.Buffer(inp)
    if in_min:
        a.set_min(*in_min, 0)
    ow, oh = out_size if out_size else (inp.shape[2] - 6, inp.shape[1] - 6)
    o = hl.Buffer(np.zeros((oh, ow), np.float32)).set_min(*out_min)
    hl.harris(a, o)
    return o.numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1536, 2560), (7, 7), (70, 22), (71, 23), (333, 201)])
def test_hip_matches_oracle_bit_for_bit(hl, oracle, w, h):
    inp = _img(w, h, seed=w + h)
    got, want = _run(hl, inp), oracle.harris(inp)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got != want)} of {got.size} differ"


@pytest.mark.gpu
def test_hip_tight_input_and_out_of_bounds(hl, oracle):
    inp = _img(60, 40, seed=5)
    got = _run(hl, inp, out_min=(12, 7), out_size=(56, 36), in_min=(10, 5))        # exactly output grown by 2
    assert np.array_equal(got.view(np.uint32), oracle.harris(inp, (12, 7), (56, 36), (10, 5)).view(np.uint32))
    with pytest.raises(hl.HalideError) as e:
        _run(hl, inp, out_min=(11, 7), out_size=(56, 36), in_min=(10, 5))
    assert e.value.code == -4


def test_bounds_query(hl):
    q = hl.Buffer.bounds_query(np.float32, 3)
    o = hl.Buffer(np.zeros((20, 30), np.float32)).set_min(3, 3)
    hl.harris(q, o)
    assert q.mins == [1, 1, 0] and q.extents == [34, 24, 3]
