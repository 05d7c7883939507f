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


def test_bf16_oracle_rounds_operands_to_nearest_even(oracle):
    """1 + 2^-8 is the midpoint between the bf16 neighbours 1 and 1 + 2^-7: ties go to the even mantissa (1);
    1 + 3*2^-8 is the midpoint between 1 + 2^-7 (odd) and 1 + 2^-6 (even)."""
    ci, co = 64, 128
    inp = np.zeros((1, 3, 3, ci), np.float32)
    filt = np.zeros((ci, 3, 3, co), np.float32)
    bias = np.zeros(co, np.float32)
    inp[0, 1, 1, 0] = 1.0 + 2.0 ** -8
    inp[0, 1, 1, 1] = 1.0 + 3 * 2.0 ** -8
    filt[0, 1, 1, 0] = 1.0
    filt[1, 1, 1, 1] = 1.0
    out, mag = oracle.conv_layer_bf16(inp, filt, bias)
    assert out[0, 0, 0, 0] == 1.0 and out[0, 0, 0, 1] == 1.0 + 2.0 ** -6
    assert mag[0, 0, 0, 0] == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w,ci,co", [(16, 56, 56, 128, 128), (5, 80, 100, 128, 128), (1, 1, 1, 64, 128),
                                         (2, 7, 9, 64, 256), (3, 5, 13, 192, 128)])
def test_hip_bf16_matches_oracle_within_accumulation_tolerance(hl, oracle, n, h, w, ci, co):
    """bf16 operands (same rounding as the oracle), f32 accumulation on the matrix cores in hardware order:
    |gpu - oracle| <= 2e-6 * (|bias| + sum |products|) + 1e-6 — about 16 ulp of the accumulation scale for
    K = 9 CI <= 1728 terms."""
    inp, filt, bias = _data(n, h, w, ci, co, seed=3 * n + h + w)
    got = _run(hl, hl.conv_layer_bf16, inp, filt, bias)
    want, mag = oracle.conv_layer_bf16(inp, filt, bias)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    tol = 2e-6 * mag.astype(np.float64) + 1e-6
    assert (err <= tol).all(), f"{np.count_nonzero(err > tol)} of {got.size} beyond tolerance, worst ratio {np.max(err / tol):.2f}"
    # and it must really be a bf16 computation: far from the exact f32 result, close to the bf16 oracle
    exact = oracle.conv_layer(inp, filt, bias)
    assert np.max(np.abs(got - exact)) > 10 * np.max(err)


@pytest.mark.gpu
def test_hip_bf16_lane_mapping_with_asymmetric_operands(hl, oracle):
    """A = one-hot pixels, B asymmetric in (ci, co): catches a swapped row/column or k-half mapping exactly
    (all values are small integers: bf16 and f32 accumulation are exact)."""
    n, h, w, ci, co = 1, 4, 8, 64, 128
    inp = np.zeros((n, h + 2, w + 2, ci), np.float32)
    rng = np.random.default_rng(0)
    for y in range(h + 2):
        for x in range(w + 2):
            inp[0, y, x, rng.integers(0, ci)] = float(rng.integers(1, 4))
    filt = ((np.arange(ci)[:, None, None, None] * 3 + np.arange(3)[None, :, None, None] * 5 +
             np.arange(3)[None, None, :, None] * 7 + np.arange(co)[None, None, None, :]) % 17 - 8).astype(np.float32)
    bias = (np.arange(co) % 5).astype(np.float32)
    got = _run(hl, hl.conv_layer_bf16, inp, filt, bias)
    want = oracle.conv_layer(inp, filt, bias)
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_hip_bf16_filter_image_follows_the_filter_contents(hl, oracle):
    """The re-ordered bf16 image of the filter is cached per (filter allocation, version): a resident filter is
    re-ordered once; a filter the caller rewrites and marks host_dirty (the reference's protocol for changed inputs,
    src/runtime/HalideRuntime.h:1699-1702) must be re-read; so must a new allocation that lands on the same address."""
    inp, filt, bias = _data(2, 6, 10, 64, 128, seed=8)
    bi, bf, bb = hl.Buffer(inp), hl.Buffer(filt), hl.Buffer(bias)

    def run():
        out = np.zeros((2, 6, 10, 128), np.float32)
        bo = hl.Buffer(out)
        hl.conv_layer_bf16(bi, bf, bb, bo)
        return bo.numpy().copy()
    first, again = run(), run()                   # second call: cached image
    assert np.array_equal(first, again)
    want, mag = oracle.conv_layer_bf16(inp, filt, bias)
    assert (np.abs(first - want) <= 2e-6 * mag + 1e-6).all()
    filt[...] = -filt                              # rewrite the SAME host array, tell the library
    bf.set_host_dirty()
    changed = run()
    want2, mag2 = oracle.conv_layer_bf16(inp, filt, bias)
    assert (np.abs(changed - want2) <= 2e-6 * mag2 + 1e-6).all() and not np.array_equal(changed, first)
    # a different filter buffer of the same size, allocated after the first one was freed (same address is likely)
    bf.device_free()
    filt3 = (filt * np.float32(0.5)).astype(np.float32)
    bf = hl.Buffer(filt3)
    third = run()
    want3, mag3 = oracle.conv_layer_bf16(inp, filt3, bias)
    assert (np.abs(third - want3) <= 2e-6 * mag3 + 1e-6).all()


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("wrapped", [False, True])
def test_hip_bf16_filter_images_are_safe_across_threads_and_streams(hl, oracle, wrapped):
    """the cache of re-ordered filters must not hand a thread an image that another thread, on another
    stream, is re-filling or evicting.  Four host threads with a stream each call conv_layer_bf16 with TEN different
    filters in random order (the cache holds eight: entries are evicted and re-filled while other streams still read
    them); wrapped = the filters are wrapped device pointers (version 0, what every torch tensor is): never cached, their
    image lives in the calling stream's scratch arena."""
    import ctypes
    import threading
    hip = hl.hip_runtime()
    n, h, w, ci, co = 2, 12, 20, 64, 128
    rng = np.random.default_rng(21)
    nf = 10
    inputs = [rng.uniform(-1, 1, (n, h + 2, w + 2, ci)).astype(np.float32) for _ in range(4)]
    filts = [rng.uniform(-1, 1, (ci, 3, 3, co)).astype(np.float32) for _ in range(nf)]
    bias = rng.uniform(-1, 1, co).astype(np.float32)
    want = [[oracle.conv_layer_bf16(inputs[t], filts[f], bias) for f in range(nf)] for t in range(4)]
    owned = [hl.Buffer(f).copy_to_device() for f in filts]          # the filters live on the device
    if wrapped:
        fbufs = [hl.Buffer.wrap_device(hl.lib.halide_hip_get_device_ptr(None, b.ptr), np.float32, (co, 3, 3, ci)) for b in owned]
    else:
        fbufs = owned
    bb = hl.Buffer(bias).copy_to_device()
    errors = []

    def worker(t):
        try:
            stream = ctypes.c_void_p()
            assert hip.hipStreamCreateWithFlags(ctypes.byref(stream), 1) == 0
            hl.set_stream(stream.value)
            r = np.random.default_rng(100 + t)
            bi = hl.Buffer(inputs[t])
            for rep in range(40):
                f = int(r.integers(0, nf))
                bo = hl.Buffer(np.zeros((n, h, w, co), np.float32))
                hl.conv_layer_bf16(bi, fbufs[f], bb, bo)
                got = bo.numpy()
                ref, mag = want[t][f]
                if not (np.abs(got - ref) <= 2e-6 * mag + 1e-6).all():
                    errors.append(f"thread {t} rep {rep} filter {f}: result is not this filter's")
                bo.device_free()
            hl.set_stream(None)
            assert hip.hipStreamSynchronize(stream) == 0
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {t}: {e!r}")

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]


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
    # the same query with shapes on the (still host-less) buffers, as RunGen passes them: the shapes are taken, not the estimates
    Q = hl.Buffer.bounds_query
    q = [Q(np.float32, 4, extents=[64, 20, 12, 2]), Q(np.float32, 4, extents=[128, 3, 3, 64]), Q(np.float32, 1, extents=[128]),
         Q(np.float32, 4, extents=[128, 18, 10, 2])]
    hl.conv_layer(*q)
    assert [b.extents for b in q] == [[64, 20, 12, 2], [128, 3, 3, 64], [128], [128, 18, 10, 2]]
    q = [Q(np.float32, 4), Q(np.float32, 4), Q(np.float32, 1), Q(np.float32, 4, extents=[256, 30, 20, 3])]   # only the output shaped
    hl.conv_layer(*q)
    assert q[0].extents == [128, 32, 22, 3] and q[1].extents == [256, 3, 3, 128] and q[2].extents == [256]
