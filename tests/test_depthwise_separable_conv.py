"""depthwise_separable_conv: zero-padded depthwise FWxFH conv + pointwise 1x1 conv + bias + ReLU, f32
(reference: /root/reference/apps/depthwise_separable_conv/depthwise_separable_conv_generator.cpp:24-75;
driver shape process.cpp:13).  Canonical rounding: fma chains in RDom order; GPU == oracle bit for bit."""
import numpy as np
import pytest


def _data(n, h, w, ci, co, cm=1, fw=3, fh=3, seed=0):
    rng = np.random.default_rng(seed)
    ic = ci * cm
    inp = rng.uniform(-1, 1, (n, h, w, ci)).astype(np.float32)     # halide [CI, W, H, N]
    dw = rng.uniform(-1, 1, (fh, fw, ic, cm)).astype(np.float32)   # halide [CM, IC, FW, FH]
    pw = rng.uniform(-1, 1, (ic, co)).astype(np.float32)           # halide [CO, IC]
    bias = rng.uniform(-1, 1, co).astype(np.float32)
    return inp, dw, pw, bias


def test_oracle_against_float64_reference(oracle):
    inp, dw, pw, bias = _data(2, 9, 11, 8, 5, seed=1)
    got = oracle.depthwise_separable_conv(inp, dw, pw, bias)
    pad = np.pad(inp.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    win = np.lib.stride_tricks.sliding_window_view(pad, (3, 3), axis=(1, 2))        # n, y, x, ci, ky, kx
    mid = np.einsum("nyxikl,kli->nyxi", win, dw[..., 0].astype(np.float64))          # CM = 1: ic == ci
    ref = np.maximum(mid @ pw.astype(np.float64) + bias.astype(np.float64), 0)
    assert np.max(np.abs(got - ref)) < 1e-5 and (got >= 0).all()


def test_oracle_is_an_ordered_fma_chain(oracle):
    inp, dw, pw, bias = _data(1, 4, 5, 4, 3, seed=2)
    got = oracle.depthwise_separable_conv(inp, dw, pw, bias)
    for (y, x, c) in [(0, 0, 0), (3, 4, 2), (2, 1, 1)]:
        mid = []
        for d in range(4):
            acc = np.float32(0)
            for ry in range(3):
                for rx in range(3):
                    yy, xx = y + ry - 1, x + rx - 1
                    v = float(inp[0, yy, xx, d]) if (0 <= yy < 4 and 0 <= xx < 5) else 0.0
                    acc = np.float32(float(dw[ry, rx, d, 0]) * v + float(acc))   # exact product, one rounding
            mid.append(acc)
        acc = np.float32(bias[c])
        for rc in range(4):
            acc = np.float32(float(pw[rc, c]) * float(mid[rc]) + float(acc))
        assert got[0, y, x, c] == max(acc, np.float32(0))


def _run(hl, inp, dw, pw, bias):
    n, h, w, ci = inp.shape
    out = np.zeros((n, h, w, pw.shape[1]), np.float32)
    bufs = [hl.Buffer(a) for a in (inp, dw, pw, bias)]
    bo = hl.Buffer(out)
    hl.depthwise_separable_conv(*bufs, bo)
    return bo.numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w,ci,co,cm,fw,fh", [(4, 112, 112, 32, 16, 1, 3, 3), (1, 1, 1, 32, 16, 1, 3, 3), (2, 7, 45, 8, 5, 1, 3, 3),
                                                 (1, 9, 33, 16, 24, 1, 5, 3), (3, 5, 13, 64, 32, 1, 1, 1)])
def test_hip_matches_oracle_bit_for_bit(hl, oracle, n, h, w, ci, co, cm, fw, fh):
    inp, dw, pw, bias = _data(n, h, w, ci, co, cm, fw, fh, seed=n + h + w)
    got = _run(hl, inp, dw, pw, bias)
    want = oracle.depthwise_separable_conv(inp, dw, pw, bias)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
        f"{np.count_nonzero(got != want)} of {got.size} differ, max abs {np.max(np.abs(got - want))}"


@pytest.mark.gpu
def test_hip_64_bit_addressing_path_matches_oracle(hl, oracle, monkeypatch):
    """Buffers too large for 32-bit offsets take dsc_fused_t<.., A32 = false> (long strides, 64-bit index chains): the same kernel
    body selected by HLMI_DSC_NO_A32=1 at the driver's MobileNet shape, bit for bit against the oracle like the default path."""
    monkeypatch.setenv("HLMI_DSC_NO_A32", "1")
    inp, dw, pw, bias = _data(4, 112, 112, 32, 16, 1, 3, 3, seed=11)
    got = _run(hl, inp, dw, pw, bias)
    want = oracle.depthwise_separable_conv(inp, dw, pw, bias)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got != want)} of {got.size} differ"


@pytest.mark.gpu
def test_hip_channel_multiplier_two(hl, oracle):
    """CM = 2: depthwise_filter's second dimension is indexed by the INTERMEDIATE channel (generator :57-62), so it
    has IC = 2 CI entries and stride CM."""
    inp, dw, pw, bias = _data(2, 6, 10, 4, 6, cm=2, seed=9)
    got = _run(hl, inp, dw, pw, bias)
    assert np.array_equal(got, oracle.depthwise_separable_conv(inp, dw, pw, bias))


@pytest.mark.gpu
def test_hip_input_that_starts_after_zero_is_out_of_bounds(hl):
    inp, dw, pw, bias = _data(1, 4, 4, 8, 4)
    bi = hl.Buffer(inp).set_min(0, 1, 0, 0)   # x starts at 1: clamp(x, 0, max) reads x = 0
    with pytest.raises(hl.HalideError) as e:
        hl.depthwise_separable_conv(bi, hl.Buffer(dw), hl.Buffer(pw), hl.Buffer(bias), hl.Buffer(np.zeros((1, 4, 4, 4), np.float32)))
    assert e.value.code == -4


def test_bounds_query_reports_the_generator_estimates(hl):
    q = [hl.Buffer.bounds_query(np.float32, d) for d in (4, 4, 2, 1, 4)]
    hl.depthwise_separable_conv(*q)
    assert q[0].extents == [32, 112, 112, 4] and q[1].extents == [1, 32, 3, 3]
    assert q[2].extents == [16, 32] and q[3].extents == [16] and q[4].extents == [16, 112, 112, 4]
