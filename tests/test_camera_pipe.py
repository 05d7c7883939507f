"""camera_pipe: raw Bayer u16 -> RGB u8 (reference: /root/reference/apps/camera_pipe/camera_pipe_generator.cpp).
Integer pipeline => bit-exact u8; parameters as in apps/camera_pipe/process.cpp:44-65 and CMakeLists (3700 2.0 50 1.0)."""
import numpy as np
import pytest

M3200 = np.array([[1.6697, -0.2693, -0.4004, -42.4346], [-0.3576, 1.0615, 1.5949, -37.1158],
                  [-0.2175, -1.8751, 6.9640, -26.6970]], np.float32)   # process.cpp:44-46
M7000 = np.array([[2.2997, -0.4478, 0.1706, -39.0923], [-0.3826, 1.5906, -0.2080, -25.4311],
                  [-0.0888, -0.7344, 2.2832, -20.0826]], np.float32)   # process.cpp:48-50
PARAMS = dict(color_temp=3700.0, gamma=2.0, contrast=50.0, sharpen=1.0, black=25, white=1023)


def _raw(w, h, seed, kind="scene"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.integers(0, 1024, (h, w), dtype=np.uint16)
    if kind == "full":
        return rng.integers(0, 65536, (h, w), dtype=np.uint16)  # exercises the u16 wrap paths of the demosaic
    yy, xx = np.mgrid[0:h, 0:w]
    scene = (np.sin(xx / 45.0 + seed) + np.cos(yy / 31.0) + 2.2) / 4.4 * 900 + 40
    raw = scene + rng.normal(0, 10, (h, w))
    raw[rng.random((h, w)) < 0.002] = 1023  # hot pixels
    return raw.clip(0, 1023).astype(np.uint16)


def test_oracle_setup_values(oracle):
    m, curve, s = oracle.camera_pipe_setup(M3200, M7000, 3700.0, 2.0, 50.0, 1.0, 25, 1023)
    assert s == 32 and m.shape == (3, 4)
    # Q8.8 of the kelvin-interpolated matrix, computed independently in float64 (coefficients are far from .5 ulp ties)
    alpha = (1.0 / 3700 - 1.0 / 3200) / (1.0 / 7000 - 1.0 / 3200)
    ref = np.trunc((M3200.astype(np.float64) * alpha + M7000.astype(np.float64) * (1 - alpha)) * 256).astype(np.int16)
    assert np.array_equal(m, ref)
    assert curve[:26].max() == 0 and curve[1023] > 250 and np.all(np.diff(curve.astype(int)) >= 0)
    # float64 evaluation of the same curve agrees to within 1 code value
    x = np.arange(1024)
    xf = np.clip((x - 25) / (1023 - 25), 0, 1)
    g = xf ** 0.5
    b = 2 - 2 ** 0.5
    a = 2 - 2 * b
    z = np.where(g > 0.5, 1 - (a * (1 - g) ** 2 + b * (1 - g)), a * g * g + b * g)
    ref = np.where(x <= 25, 0, np.clip(z * 255 + 0.5, 0, 255).astype(np.uint8))
    assert np.max(np.abs(curve.astype(int) - ref.astype(int))) <= 1


def test_oracle_flat_gray_scene(oracle):
    raw = np.full((80, 100), 500, np.uint16)
    out = oracle.camera_pipe(raw, M3200, M7000, 3700.0, 2.0, 50.0, 1.0, 25, 1023, 64, 32)
    for c in range(3):
        assert np.all(out[c] == out[c, 0, 0])  # flat in, flat out (sharpening of a constant is the constant)


def _run(hl, raw, out_w, out_h, p=PARAMS, in_min=None, out_min=None):
    bi, b3, b7 = hl.Buffer(raw), hl.Buffer(M3200.copy()), hl.Buffer(M7000.copy())
    bo = hl.Buffer(np.zeros((3, out_h, out_w), np.uint8))
    if in_min:
        bi.set_min(*in_min)
    if out_min:
        bo.set_min(*out_min)
    hl.camera_pipe(bi, b3, b7, p["color_temp"], p["gamma"], p["contrast"], p["sharpen"], p["black"], p["white"], bo)
    return bo.numpy()


def _oracle(oracle, raw, out_w, out_h, p=PARAMS):
    return oracle.camera_pipe(raw, M3200, M7000, p["color_temp"], p["gamma"], p["contrast"], p["sharpen"], p["black"],
                              p["white"], out_w, out_h)


@pytest.mark.gpu
@pytest.mark.parametrize("iw,ih,kind", [(2592, 1968, "scene"), (2592, 1968, "uniform"), (160, 120, "full"), (96, 72, "scene"),
                                        (57, 49, "uniform")])
def test_hip_matches_oracle(hl, oracle, iw, ih, kind):
    raw = _raw(iw, ih, seed=iw + ih, kind=kind)
    ow, oh = ((iw - 32) // 32) * 32, ((ih - 24) // 32) * 32   # process.cpp:34
    if ow <= 0:
        ow, oh = iw - 22, ih - 18                               # largest output the footprint allows
    got, want = _run(hl, raw, ow, oh), _oracle(oracle, raw, ow, oh)
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} of {got.size} differ"


@pytest.mark.gpu
@pytest.mark.parametrize("p", [dict(color_temp=3200.0, gamma=1.0, contrast=0.0, sharpen=0.0, black=0, white=1023),
                               dict(color_temp=7000.0, gamma=2.2, contrast=100.0, sharpen=4.0, black=64, white=900),
                               dict(color_temp=5000.0, gamma=1.8, contrast=25.0, sharpen=7.97, black=25, white=1023)])
def test_hip_parameter_sweep(hl, oracle, p):
    raw = _raw(200, 150, seed=5)
    got, want = _run(hl, raw, 160, 120, p), _oracle(oracle, raw, 160, 120, p)
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_hip_odd_output_size_and_out_of_bounds(hl, oracle):
    raw = _raw(120, 90, seed=9)
    assert np.array_equal(_run(hl, raw, 97, 71), _oracle(oracle, raw, 97, 71))
    with pytest.raises(hl.HalideError) as e:
        _run(hl, raw, 99, 71)   # needs input x up to 99 + 21 = 120 > 119
    assert e.value.code == -4


def test_bounds_query(hl):
    q = hl.Buffer.bounds_query(np.uint16, 2)
    m3, m7 = hl.Buffer(M3200.copy()), hl.Buffer(M7000.copy())
    out = hl.Buffer(np.zeros((3, 1920, 2560), np.uint8))
    hl.camera_pipe(q, m3, m7, 3700.0, 2.0, 50.0, 1.0, 25, 1023, out)
    assert q.mins == [10, 6] and q.extents == [2572, 1932]
