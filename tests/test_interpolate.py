"""interpolate: alpha-weighted pull-push pyramid, 10 levels, f32 RGBA planar -> RGB
(reference: /root/reference/apps/interpolate/interpolate_generator.cpp:20-77).  The oracle evaluates every level on the box
its readers touch; it is cross-checked against a naive evaluator in which every Func is a memoised pure function on Z^2
(no boxes at all), so a wrong region or a missing clamp shows up as a difference.  GPU == oracle bit for bit."""
import functools

import numpy as np
import pytest

F = np.float32


def _img(w, h, seed, holes=True):
    rng = np.random.default_rng(seed)
    rgb = rng.random((3, h, w), dtype=np.float32)
    alpha = rng.random((h, w), dtype=np.float32)
    if holes:
        alpha[rng.random((h, w)) < 0.4] = 0.0        # the app's purpose: fill transparent holes from their surroundings
    alpha[0, 0] = 1.0
    return np.concatenate([rgb, alpha[None]]).astype(np.float32)


def naive_interpolate(inp):
    _, H, W = inp.shape
    L = 10
    fd = lambda a, b: a // b   # floor division, as Halide's

    @functools.lru_cache(maxsize=None)
    def down(l, x, y, c):
        if l == 0:
            xx, yy = min(max(x, 0), W - 1), min(max(y, 0), H - 1)
            a = inp[3, yy, xx]
            return F(inp[c, yy, xx] * a) if c < 3 else F(a)

        def prev(px, py):
            if l == 4:
                px, py = min(max(px, 0), W // 8), min(max(py, 0), H // 8)
            return down(l - 1, px, py, c)

        def downx(px, py):
            return F(F(F(prev(2 * px - 1, py) + F(F(2.0) * prev(2 * px, py))) + prev(2 * px + 1, py)) * F(0.25))
        return F(F(F(downx(x, 2 * y - 1) + F(F(2.0) * downx(x, 2 * y))) + downx(x, 2 * y + 1)) * F(0.25))

    @functools.lru_cache(maxsize=None)
    def interp(l, x, y, c):
        if l == L - 1:
            return down(l, x, y, c)

        def upx(px, py):
            return F(F(interp(l + 1, fd(px, 2), py, c) + interp(l + 1, fd(px + 1, 2), py, c)) * F(0.5))
        up = F(F(upx(x, fd(y, 2)) + upx(x, fd(y + 1, 2))) * F(0.5))
        alpha = F(F(1.0) - down(l, x, y, 3))
        return F(down(l, x, y, c) + F(alpha * up))

    out = np.zeros((3, H, W), np.float32)
    with np.errstate(all="ignore"):
        for y in range(H):
            for x in range(W):
                for c in range(3):
                    out[c, y, x] = F(interp(0, x, y, c) / interp(0, x, y, 3))
    return out


@pytest.mark.parametrize("w,h", [(13, 9), (1, 1), (24, 17)])
def test_oracle_matches_naive_pure_function_evaluator(oracle, w, h, canon0):
    inp = _img(w, h, seed=w * 31 + h)
    got, want = oracle.interpolate(inp), naive_interpolate(inp)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got != want)} of {got.size} differ"


def test_oracle_boxes(oracle):
    bi, bd = oracle.interpolate_boxes(1536, 2560)
    assert list(bi[0]) == [0, 1535, 0, 2559] and list(bi[1]) == [0, 768, 0, 1280]
    assert bd[9].tolist() == bi[9].tolist()
    assert bd[3][0] == 0 and bd[3][2] == 0 and bd[3][1] <= max(bi[3][1], 1536 // 8)      # the clamp in front of level 4
    assert bd[8][0] == -1 and bd[7][0] == -3 and bd[4][0] == -31 and bd[2][0] == -1 and bd[1][0] == -3


def test_oracle_fills_holes_and_keeps_opaque_pixels(oracle):
    inp = _img(64, 48, seed=2)
    out = oracle.interpolate(inp)
    assert np.isfinite(out).all()
    opaque = inp[3] == 1.0
    assert np.allclose(out[:, opaque], inp[:3, opaque], atol=1e-5)     # alpha = 1: the pixel's own colour


def _run(hl, inp):
    a = hl.Buffer(inp)
    o = hl.Buffer(np.zeros((3,) + inp.shape[1:], np.float32))
    hl.interpolate(a, o)
    return o.numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1536, 2560), (1, 1), (13, 9), (255, 257), (640, 480)])
def test_hip_matches_oracle_bit_for_bit(hl, oracle, w, h):
    inp = _img(w, h, seed=w + h)
    got, want = _run(hl, inp), oracle.interpolate(inp)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got.view(np.uint32) != want.view(np.uint32))} of {got.size} differ"


@pytest.mark.gpu
def test_hip_pinned_shapes(hl):
    inp = _img(32, 24, seed=1)
    with pytest.raises(hl.HalideError) as e:       # three channels in: input.dim(2).set_bounds(0, 4) (:23)
        hl.interpolate(hl.Buffer(inp[:3].copy()), hl.Buffer(np.zeros((3, 24, 32), np.float32)))
    assert e.value.code == -8
    with pytest.raises(hl.HalideError) as e:       # output must span the input's extent (:83-87)
        hl.interpolate(hl.Buffer(inp), hl.Buffer(np.zeros((3, 24, 30), np.float32)))
    assert e.value.code == -8


def test_bounds_query(hl):
    q = hl.Buffer.bounds_query(np.float32, 3)
    o = hl.Buffer(np.zeros((3, 20, 30), np.float32))
    hl.interpolate(q, o)
    assert q.extents == [30, 20, 4]
    Q = hl.Buffer.bounds_query                       # all-null queries (RunGen's): the output's shape is the request
    qi, qo = Q(np.float32, 3, extents=[50, 30, 4]), Q(np.float32, 3, extents=[60, 20, 3])
    hl.interpolate(qi, qo)
    assert qi.extents == [60, 20, 4] and qo.extents == [60, 20, 3]
    qi, qo = Q(np.float32, 3, extents=[50, 30, 4]), Q(np.float32, 3)
    hl.interpolate(qi, qo)
    assert qi.extents == [50, 30, 4] and qo.extents == [50, 30, 3]
