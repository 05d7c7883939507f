"""conv_layer: 3x3 conv + bias + ReLU, f32, layout (c, x, y, n) (reference:
/root/reference/apps/conv_layer/conv_layer_generator.cpp:21-27, layouts :35-50).
Exact path: k-ordered fma chain == the oracle bit for bit (f32 matrix cores).  bf16 path: tolerance."""
import numpy as np
import pytest


def _data(n, h, w, ci, co, seed):
    rng = np.random.default_rng(seed)
    inp = rng.uniform(-1, 1, (n, h + 2, w + 2, ci)).astype(np.float32)   # halide [CI, W+2, H+2, N]
    filt = rng.uniform(-1, 1, (ci, 3, 3, co)).astype(np.float32)         # halide [CO, kx, ky, CI]
    bias = rng.uniform(-1, 1, co).astype(np.float32)
    return inp, filt, bias


def test_oracle_against_float64_einsum(oracle):
    inp, filt, bias = _data(2, 9, 11, 32, 128, 0)
    got = oracle.conv_layer(inp, filt, bias)
    win = np.lib.stride_tricks.sliding_window_view(inp.astype(np.float64), (3, 3), axis=(1, 2))  # n,y,x,ci,ky,kx
    ref = np.einsum("nyxikl,iklo->nyxo", win, filt.astype(np.float64)) + bias.astype(np.float64)
    ref = np.maximum(ref, 0)
    assert np.max(np.abs(got - ref)) < 2e-5
    assert (got >= 0).all()


def test_oracle_is_a_k_ordered_fma_chain(oracle):
    """Spot-check the canonical rounding: acc = fma(f, x, acc) in RDom order (ci fastest, kx, ky), from bias."""
    import math
    inp, filt, bias = _data(1, 3, 3, 32, 128, 1)
    got = oracle.conv_layer(inp, filt, bias)
    for (y, x, c) in [(0, 0, 0), (2, 1, 77), (1, 2, 127)]:
        acc = np.float32(bias[c])
        for ky in range(3):
            for kx in range(3):
                for ci in range(32):
                    # exact fma via float64: product of two f32 is exact in f64; the sum rounds once to f32
                    # (double rounding is harmless here: f64 has > 2*24+2 bits)
                    acc = np.float32(float(filt[ci, ky, kx, c]) * float(inp[0, y + ky, x + kx, ci]) + float(acc))
        assert got[0, y, x, c] == max(acc, np.float32(0))


def _run(hl, fn, inp, filt, bias):
    n, hp, wp, ci = inp.shape
    co = bias.shape[0]
    out = np.zeros((n, hp - 2, wp - 2, co), np.float32)
    bi, bf, bb, bo = hl.Buffer(inp), hl.Buffer(filt), hl.Buffer(bias), hl.Buffer(out)
    fn(bi, bf, bb, bo)
    return bo.numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w,ci,co", [(5, 80, 100, 128, 128), (16, 56, 56, 128, 128), (1, 1, 1, 32, 128),
                                         (2, 7, 9, 64, 256), (3, 5, 13, 96, 128)])
def test_hip_f32_exact_matches_oracle(hl, oracle, n, h, w, ci, co):
    inp, filt, bias = _data(n, h, w, ci, co, seed=n + h + w)
    got = _run(hl, hl.conv_layer, inp, filt, bias)
    want = oracle.conv_layer(inp, filt, bias)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
        f"{np.count_nonzero(got != want)} of {got.size} differ, max abs {np.max(np.abs(got - want))}"


@pytest.mark.gpu
def test_hip_rejects_non_dense_layout(hl):
    inp, filt, bias = _data(1, 4, 4, 32, 128, 0)
    bi, bf, bb = hl.Buffer(inp), hl.Buffer(filt), hl.Buffer(bias)
    bo = hl.Buffer(np.zeros((1, 4, 4, 128), np.float32)).set_min(0, 1, 0, 0)
    with pytest.raises(hl.HalideError) as e:
        hl.conv_layer(bi, bf, bb, bo)
    assert e.value.code == -8


def test_bounds_query_reports_reference_shapes(hl):
    q = [hl.Buffer.bounds_query(np.float32, 4), hl.Buffer.bounds_query(np.float32, 4),
         hl.Buffer.bounds_query(np.float32, 1), hl.Buffer.bounds_query(np.float32, 4)]
    hl.conv_layer(*q)
    assert q[0].extents == [128, 102, 82, 5] and q[1].extents == [128, 3, 3, 128]
    assert q[2].extents == [128] and q[3].extents == [128, 100, 80, 5]
