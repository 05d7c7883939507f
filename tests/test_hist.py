"""hist: histogram equalisation of the luma, u8 planar (reference: /root/reference/apps/hist/hist_generator.cpp:13-56).
The histogram is integer (exact, order-free: wave-private LDS histograms + a shuffle scan on the GPU); the pointwise
float part rounds once per operator.  GPU == oracle bit for bit."""
import numpy as np
import pytest


def _img(w, h, seed, kind="scene"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.integers(0, 256, (3, h, w), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = 90 + 60 * np.sin(xx / 23.0) * np.cos(yy / 17.0)
    img = np.stack([base, base * 0.8 + 20, base[::-1] * 0.6 + 40]) + rng.normal(0, 12, (3, h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def test_oracle_cdf_is_the_luma_histogram_prefix_sum(oracle):
    inp = _img(61, 47, 1)
    _, cdf = oracle.hist(inp, return_cdf=True)
    f = inp.astype(np.float32)
    y = (np.float32(0.299) * f[0] + np.float32(0.587) * f[1]) + np.float32(0.114) * f[2]
    hist = np.bincount(np.clip(y, 0, 255).astype(np.int32).ravel(), minlength=256)
    assert np.array_equal(cdf, np.cumsum(hist)) and cdf[-1] == 61 * 47


def test_oracle_flat_image_maps_to_full_scale(oracle):
    inp = np.full((3, 8, 8), 100, np.uint8)
    out = oracle.hist(inp)
    assert (out == 255).all()   # every pixel is in the top cdf bin: eq = 255, chroma offsets are 0 for gray input



def _numpy_hist(inp):
    """Second reading of the generator (:13-56), array at a time in float32, one rounding per operator as written."""
    f32 = np.float32
    r, g, b = (inp[c].astype(f32) for c in range(3))
    y = (f32(0.299) * r + f32(0.587) * g) + f32(0.114) * b
    cr = (r - y) * f32(0.713) + f32(128)
    cb = (b - y) * f32(0.564) + f32(128)
    bins = np.clip(y, f32(0), f32(255)).astype(np.int32)                # cast<int>(clamp(Y, 0, 255))
    cdf = np.cumsum(np.bincount(bins.ravel(), minlength=256)).astype(np.int32)
    scale = f32(255.0) / f32(inp.shape[1] * inp.shape[2])
    eq = np.clip(cdf[bins].astype(f32) * scale, f32(0), f32(255))
    red = np.clip(eq + (cr - f32(128)) * f32(1.4), f32(0), f32(255))
    green = np.clip((eq - f32(0.343) * (cb - f32(128))) - f32(0.711) * (cr - f32(128)), f32(0), f32(255))
    blue = np.clip(eq + f32(1.765) * (cb - f32(128)), f32(0), f32(255))
    return np.stack([red, green, blue]).astype(np.uint8)                # u8(): truncation of values in [0, 255]


@pytest.mark.parametrize("kind,seed", [("scene", 5), ("uniform", 6)])
def test_oracle_matches_numpy_restatement(oracle, kind, seed, canon0):
    inp = _img(90, 70, seed, kind)
    got, want = oracle.hist(inp), _numpy_hist(inp)
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} of {got.size} differ"


def _run(hl, inp, out_min=None, out_size=None):
    a = hl.Buffer(inp)
    ow, oh = out_size if out_size else (inp.shape[2], inp.shape[1])
    o = hl.Buffer(np.zeros((3, oh, ow), np.uint8))
    if out_min:
        o.set_min(*out_min, 0)
    hl.hist(a, o)
    return o.numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,kind", [(1536, 2560, "scene"), (1536, 2560, "uniform"), (1, 1, "scene"), (7, 5, "uniform"), (64, 16, "scene"),
                                      (333, 201, "scene"), (1024, 3, "uniform")])
def test_hip_matches_oracle_bit_for_bit(hl, oracle, w, h, kind):
    inp = _img(w, h, seed=w + h, kind=kind)
    got, want = _run(hl, inp), oracle.hist(inp)
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} of {got.size} differ"


@pytest.mark.gpu
def test_hip_crop_uses_the_histogram_of_the_whole_input(hl, oracle):
    inp = _img(200, 120, seed=3)
    got = _run(hl, inp, out_min=(36, 10), out_size=(64, 40))
    assert np.array_equal(got, oracle.hist(inp, out_origin=(36, 10), out_size=(64, 40)))
    got = _run(hl, inp, out_min=(37, 11), out_size=(61, 39))   # unaligned: scalar path
    assert np.array_equal(got, oracle.hist(inp, out_origin=(37, 11), out_size=(61, 39)))


@pytest.mark.gpu
def test_hip_input_must_start_at_zero(hl):
    inp = _img(32, 32, seed=0)
    a = hl.Buffer(inp).set_min(4, 0, 0)
    with pytest.raises(hl.HalideError) as e:
        hl.hist(a, hl.Buffer(np.zeros((3, 8, 8), np.uint8)).set_min(4, 0, 0))
    assert e.value.code == -4
