"""max_filter: max over a disc-like footprint (radius 26) of the edge-clamped input, f32 planar
(reference: /root/reference/apps/max_filter/max_filter_generator.cpp:14-53).  The oracle evaluates the reference's
log-slice construction literally; the GPU kernel evaluates its closed form.  Only comparisons are involved, so the two
must agree bit for bit."""
import numpy as np
import pytest


def _img(w, h, seed, ch=3):
    rng = np.random.default_rng(seed)
    return rng.random((ch, h, w), dtype=np.float32)


def _footprint():
    """h(dx) as the generator defines it (:47-53): rows |dy| <= h(dx) of column dx belong to the footprint."""
    lim = np.float32(26.25) * np.float32(26.25)
    h = {}
    for dx in range(-26, 27):
        n = sum(1 for dy in range(27) if np.float32(dx * dx + dy * dy) < lim)
        h[dx] = min(max(n, 0), 27)
    return h


def _brute(inp, ox, oy, ow, oh):
    h = _footprint()
    pad = 96
    big = np.pad(inp, ((0, 0), (pad, pad), (pad, pad)), mode="edge")
    want = np.full((inp.shape[0], oh, ow), -np.inf, np.float32)
    for dx in range(-26, 27):
        for dy in range(-h[dx], h[dx] + 1):
            want = np.maximum(want, big[:, pad + oy + dy:pad + oy + dy + oh, pad + ox + dx:pad + ox + dx + ow])
    return want


def test_oracle_is_the_footprint_max(oracle):
    """The literal log-slice evaluation equals a brute-force max over {(dx, dy): |dx| <= 26, |dy| <= h(dx)} of the
    edge-clamped input for every output row y >= 0 — including rows below and columns beside the input."""
    inp = _img(45, 38, 1)
    h = _footprint()
    assert h[0] == 27 and h[26] == 4 and h[-26] == 4
    ox, oy, ow, oh = -37, 0, 120, 80
    assert np.array_equal(oracle.max_filter(inp, out_origin=(ox, oy), out_size=(ow, oh)), _brute(inp, ox, oy, ow, oh))


def test_oracle_rows_above_the_image_miss_part_of_the_footprint(oracle):
    """As written, vert_log is only updated on rows -26 .. height-1 (generator :28): for output rows -26 .. -12 a sample
    that starts below row -26 holds row 0 alone, and the result is smaller than the footprint max.  Rows < -27 and rows
    >= -11 agree again."""
    inp = _img(40, 50, 2)
    ox, oy, ow, oh = 0, -40, 40, 40
    got, full = oracle.max_filter(inp, out_origin=(ox, oy), out_size=(ow, oh)), _brute(inp, ox, oy, ow, oh)
    assert np.all(got <= full)
    rows = sorted({int(y) + oy for y in np.argwhere(got != full)[:, 1]})
    assert rows and rows[0] >= -26 and rows[-1] <= -12


def test_oracle_tables(oracle):
    sfr, fh = oracle.max_filter_tables()
    assert list(sfr[:5]) == [0, 1, 2, 2, 3] and sfr[27] == 5 and sfr[15] == 4 and sfr[16] == 5
    h = _footprint()
    assert [min(v, 27) for v in fh] == [h[dx] for dx in range(-26, 27)]


def _run(hl, inp, out_min=None, out_size=None, in_min=None):
    a = hl.Buffer(inp)
    if in_min:
        a.set_min(*in_min, 0)
    ow, oh = out_size if out_size else (inp.shape[2], inp.shape[1])
    o = hl.Buffer(np.zeros((inp.shape[0], oh, ow), np.float32))
    if out_min:
        o.set_min(*out_min, 0)
    hl.max_filter(a, o)
    return o.numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1536, 2560), (1, 1), (7, 5), (64, 32), (65, 33), (333, 201), (27, 300)])
def test_hip_matches_oracle_bit_for_bit(hl, oracle, w, h):
    inp = _img(w, h, seed=w + h)
    got, want = _run(hl, inp), oracle.max_filter(inp)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got != want)} of {got.size} differ"


@pytest.mark.gpu
def test_hip_regions_outside_the_input_and_nonzero_x_min(hl, oracle):
    inp = _img(120, 90, seed=3)
    got = _run(hl, inp, out_min=(-40, -33), out_size=(230, 170), in_min=(5, 0))
    want = oracle.max_filter(inp, out_origin=(-40, -33), out_size=(230, 170), in_origin=(5, 0))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_hip_idempotent_on_constant_and_monotone(hl):
    """Size-independent properties: a constant image is a fixed point; the filter dominates its input and is monotone."""
    c = np.full((3, 100, 140), 0.375, np.float32)
    assert np.array_equal(_run(hl, c), c)
    a = _img(300, 200, 5)
    fa = _run(hl, a)
    assert np.all(fa >= a)
    b = np.maximum(a, _img(300, 200, 6) * 0.8)
    assert np.all(_run(hl, b) >= fa)


@pytest.mark.gpu
def test_hip_input_rows_must_start_at_zero(hl):
    inp = _img(32, 32, seed=0)
    with pytest.raises(hl.HalideError):
        _run(hl, inp, in_min=(0, 3))


def test_bounds_query(hl):
    q = hl.Buffer.bounds_query(np.float32, 3)
    o = hl.Buffer(np.zeros((3, 20, 30), np.float32)).set_min(4, 2, 0)
    hl.max_filter(q, o)
    assert q.mins == [4, 2, 0] and q.extents == [30, 20, 3]
