"""iir_blur: first-order IIR low pass down/up the columns, then along the rows, f32 planar
(reference: /root/reference/apps/iir_blur/iir_blur_generator.cpp:13-31, 146-156).  The recurrences are sequential by
definition; GPU == oracle bit for bit."""
import numpy as np
import pytest


def test_oracle_against_float64_recurrence(oracle):
    rng = np.random.default_rng(0)
    inp = rng.random((2, 9, 13), dtype=np.float32)
    got = oracle.iir_blur(inp, 0.3)

    def cols(a, alpha):                       # a: (H, W) -> blurred columns
        b = a.astype(np.float64).copy()
        for y in range(1, b.shape[0]):
            b[y] = (1 - alpha) * b[y - 1] + alpha * a[y]
        for y in range(b.shape[0] - 2, -1, -1):
            b[y] = (1 - alpha) * b[y + 1] + alpha * b[y]
        return b
    alpha = float(np.float32(0.3))
    ref = np.stack([cols(cols(ch, alpha).T, alpha).T for ch in inp])
    assert np.max(np.abs(got - ref)) < 1e-5



def test_oracle_matches_float32_recurrence_bit_for_bit(oracle, canon0):
    """The same recurrence (generator :18-31, :146-156) row by row in float32: (1 - alpha) is rounded once, every step is
    one multiply, one multiply, one add."""
    f32 = np.float32
    rng = np.random.default_rng(1)
    inp = rng.random((3, 23, 17), dtype=np.float32)
    alpha = f32(0.3)
    c1 = f32(1) - alpha

    def cols_T(a):                            # (H, W) -> blurred columns, transposed (W, H)
        b = a.copy()
        for y in range(1, b.shape[0]):
            b[y] = c1 * b[y - 1] + alpha * a[y]
        for y in range(b.shape[0] - 2, -1, -1):
            b[y] = c1 * b[y + 1] + alpha * b[y]
        return np.ascontiguousarray(b.T)
    want = np.stack([cols_T(cols_T(ch)) for ch in inp])
    got = oracle.iir_blur(inp, float(alpha))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got != want)} differ"


def test_oracle_constant_image_is_a_fixed_point(oracle):
    inp = np.full((3, 20, 30), 0.625, np.float32)
    assert np.array_equal(oracle.iir_blur(inp, 0.5), inp)     # (1-a) v + a v with a = 1/2 and v = 5/8 is exact


def _run(hl, inp, alpha):
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
    hl.iir_blur(a, alpha, o)
    return o.numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,c,alpha", [(1536, 2560, 3, 0.1), (1, 1, 3, 0.5), (64, 64, 1, 0.5), (65, 130, 3, 0.25), (200, 77, 2, 0.9),
                                            (1024, 70, 1, 0.3), (1088, 1024, 1, 0.6), (960, 1030, 2, 0.2), (1027, 1152, 1, 0.45)])
def test_hip_matches_oracle_bit_for_bit(hl, oracle, w, h, c, alpha):
    rng = np.random.default_rng(w + h)
    inp = rng.random((c, h, w), dtype=np.float32)
    got, want = _run(hl, inp, alpha), oracle.iir_blur(inp, alpha)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got != want)} of {got.size} differ"


@pytest.mark.gpu
def test_hip_output_shape_is_pinned_to_the_input(hl):
    inp = np.zeros((3, 16, 16), np.float32)
    with pytest.raises(hl.HalideError) as e:
        hl.iir_blur(hl.Buffer(inp), 0.5, hl.Buffer(np.zeros((3, 16, 15), np.float32)))
    assert e.value.code == -8


def test_bounds_query(hl):
    q = hl.Buffer.bounds_query(np.float32, 3)
    hl.iir_blur(q, 0.5, hl.Buffer(np.zeros((3, 20, 30), np.float32)))
    assert q.extents == [30, 20, 3]
    # RunGen's queries pass EVERY buffer without a host (tools/RunGen.h:1212-1250): the output's shape is then the request, as
    # in Halide's bounds inference; only a query that names no shape at all falls back to the generator's estimates
    Q = hl.Buffer.bounds_query
    qi, qo = Q(np.float32, 3, extents=[96, 64, 3]), Q(np.float32, 3, extents=[40, 24, 3])
    hl.iir_blur(qi, 0.5, qo)
    assert qi.extents == [40, 24, 3] and qo.extents == [40, 24, 3]
    qi, qo = Q(np.float32, 3, extents=[96, 64, 3]), Q(np.float32, 3)
    hl.iir_blur(qi, 0.5, qo)
    assert qi.extents == [96, 64, 3] and qo.extents == [96, 64, 3]
    qi, qo = Q(np.float32, 3), Q(np.float32, 3)
    hl.iir_blur(qi, 0.5, qo)
    assert qi.extents == [1536, 2560, 3] and qo.extents == [1536, 2560, 3]
