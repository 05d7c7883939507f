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



def test_oracle_matches_float32_numpy_restatement_bit_for_bit(oracle, canon0):
    """Second reading of the generator (:7-62), array at a time in float32: same operator order, one rounding each."""
    f32 = np.float32
    inp = _img(57, 43, 2)
    h, w = inp.shape[1:]
    g = (f32(0.299) * inp[0] + f32(0.587) * inp[1]) + f32(0.114) * inp[2]
    a, b, c, d = f32(-1.0) / f32(12), f32(1.0) / f32(12), f32(-2.0) / f32(12), f32(2.0) / f32(12)
    G = lambda dx, dy: g[1 + dy:h - 1 + dy, 1 + dx:w - 1 + dx]            # gray(x + dx, y + dy) on the interior
    iy = ((((G(-1, -1) * a + G(-1, 1) * b) + G(0, -1) * c) + G(0, 1) * d) + G(1, -1) * a) + G(1, 1) * b
    ix = ((((G(-1, -1) * a + G(1, -1) * b) + G(-1, 0) * c) + G(1, 0) * d) + G(-1, 1) * a) + G(1, 1) * b

    def s3(f):                                                            # sum3x3 (:7-11): x outer, y inner, left to right
        hh, ww = f.shape
        F = lambda dx, dy: f[1 + dy:hh - 1 + dy, 1 + dx:ww - 1 + dx]
        acc = F(-1, -1)
        for dx, dy in [(-1, 0), (-1, 1), (0, -1), (0, 0), (0, 1), (1, -1), (1, 0), (1, 1)]:
            acc = acc + F(dx, dy)
        return acc
    sxx, syy, sxy = s3(ix * ix), s3(iy * iy), s3(ix * iy)
    trace = sxx + syy
    want = (sxx * syy - sxy * sxy) - (f32(0.04) * trace) * trace         # output region starts at (2, 2)
    got = oracle.harris(inp, out_origin=(2, 2), out_size=(w - 4, h - 4))
    assert got.dtype == np.float32 and want.dtype == np.float32
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got != want)} differ"


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
