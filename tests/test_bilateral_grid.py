"""bilateral_grid (BASELINE configs[1]): f32 1920x1080, s_sigma=8, r_sigma=0.1.
Reference algorithm: /root/reference/apps/bilateral_grid/bilateral_grid_generator.cpp:14-67.
Parity bar: GPU == oracle bit-for-bit (<= 1 ulp is the stated tolerance; the kernels keep the oracle's
operation order so the test demands equality).  Oracle <-> real Halide is unpinned (no golden in the reference)."""
import numpy as np
import pytest

f32 = np.float32


def naive_bilateral_grid(inp, r_sigma):
    """Independent lazy restatement straight from the generator text (pure functions on Z^3, f32 scalars)."""
    import functools
    H, W = inp.shape
    s = 8
    inv_r = f32(1.0) / f32(r_sigma)

    def clamp01(v):
        return max(min(v, f32(1.0)), f32(0.0))

    @functools.lru_cache(maxsize=None)
    def hist_cell(x, y):
        h = {}
        for ry in range(s):
            for rx in range(s):
                px, py = min(max(x * s + rx - s // 2, 0), W - 1), min(max(y * s + ry - s // 2, 0), H - 1)
                val = clamp01(f32(inp[py, px]))
                zi = int(val * inv_r + f32(0.5))
                a = h.get(zi, (f32(0.0), f32(0.0)))
                h[zi] = (a[0] + val, a[1] + f32(1.0))
        return h

    def hist(x, y, z, c):
        return hist_cell(x, y).get(z, (f32(0.0), f32(0.0)))[c]

    def blur(f):
        # 5 taps along the axis chosen by the caller via the shift function
        def b(vals):
            return (((vals[0] + vals[1] * f32(4)) + vals[2] * f32(6)) + vals[3] * f32(4)) + vals[4]
        return b
    b5 = blur(None)

    @functools.lru_cache(maxsize=None)
    def blurz(x, y, z, c):
        return b5([hist(x, y, z + d, c) for d in (-2, -1, 0, 1, 2)])

    @functools.lru_cache(maxsize=None)
    def blurx(x, y, z, c):
        return b5([blurz(x + d, y, z, c) for d in (-2, -1, 0, 1, 2)])

    @functools.lru_cache(maxsize=None)
    def blury(x, y, z, c):
        return b5([blurx(x, y + d, z, c) for d in (-2, -1, 0, 1, 2)])

    def lerp(a, b, w):
        return a * (f32(1.0) - w) + b * w

    out = np.zeros_like(inp)
    for y in range(H):
        for x in range(W):
            val = clamp01(f32(inp[y, x]))
            zv = val * inv_r
            zi = int(zv)
            zf = zv - f32(zi)
            xf, yf = f32(x % s) * f32(0.125), f32(y % s) * f32(0.125)
            xi, yi = x // s, y // s
            r = []
            for c in range(2):
                a = lerp(lerp(blury(xi, yi, zi, c), blury(xi + 1, yi, zi, c), xf),
                         lerp(blury(xi, yi + 1, zi, c), blury(xi + 1, yi + 1, zi, c), xf), yf)
                b = lerp(lerp(blury(xi, yi, zi + 1, c), blury(xi + 1, yi, zi + 1, c), xf),
                         lerp(blury(xi, yi + 1, zi + 1, c), blury(xi + 1, yi + 1, zi + 1, c), xf), yf)
                r.append(lerp(a, b, zf))
            out[y, x] = r[0] / r[1]
    return out


def _img(w, h, seed, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((h, w), dtype=np.float32)  # RunGen convention: floats uniform [0,1)
    yy, xx = np.mgrid[0:h, 0:w]
    s = (np.sin(xx / 23.0 + seed) + np.cos(yy / 17.0)) / 4 + 0.5 + (xx > w // 2) * 0.2 + rng.normal(0, 0.02, (h, w))
    return s.astype(np.float32)  # deliberately leaves [0,1] in places: exercises the clamp


@pytest.mark.parametrize("w,h,r_sigma", [(11, 9, 0.1), (20, 13, 0.25), (1, 1, 0.1)])
def test_oracle_matches_naive_restatement(oracle, canon0, w, h, r_sigma):
    inp = _img(w, h, seed=w + h, kind="smooth")
    got = oracle.bilateral_grid(inp, r_sigma)
    want = naive_bilateral_grid(inp, r_sigma)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_oracle_preserves_constant_and_smooth_images(oracle):
    c = np.full((40, 56), 0.3, np.float32)
    assert np.max(np.abs(oracle.bilateral_grid(c, 0.1) - c)) < 1e-6
    s = _img(160, 120, 3, "smooth").clip(0, 1)
    assert np.max(np.abs(oracle.bilateral_grid(s, 0.1) - s)) < 0.12


@pytest.fixture(params=["two_launches", "one_launch"])
def bg_path(request, monkeypatch):
    """Round 6 built a one-launch form for grids of at most 16 planes (bg_blur_slice<.., HIST = true> builds the tile's blurz cells from
    the input itself; HLMI_BG_ONE_LAUNCH=1).  It is slower and therefore opt-in, but both paths face the oracle."""
    if request.param == "one_launch":
        monkeypatch.setenv("HLMI_BG_ONE_LAUNCH", "1")
    return request.param


def _run(hl, inp, r_sigma, out=None, in_min=None, out_min=None):
    a = hl.Buffer(inp)
    o = hl.Buffer(np.zeros_like(inp) if out is None else out)
    if in_min:
        a.set_min(*in_min)
    if out_min:
        o.set_min(*out_min)
    hl.bilateral_grid(a, r_sigma, o)
    return o.numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,r_sigma,kind", [
    (1920, 1080, 0.1, "uniform"), (1920, 1080, 0.1, "smooth"), (1, 1, 0.1, "uniform"), (7, 9, 0.1, "uniform"),
    (8, 8, 0.1, "smooth"), (333, 129, 0.05, "smooth"), (250, 100, 0.3, "uniform"), (250, 100, 1.0, "smooth"),
    (64, 64, 0.004, "uniform"), (1536, 2560, 0.1, "uniform"),
])
def test_hip_matches_oracle(hl, oracle, bg_path, w, h, r_sigma, kind):
    inp = _img(w, h, seed=w + h, kind=kind)
    got = _run(hl, inp, r_sigma)
    want = oracle.bilateral_grid(inp, r_sigma)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
        f"{np.count_nonzero(got.view(np.uint32) != want.view(np.uint32))} of {got.size} differ"


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,r_sigma", [(1920, 1080, 0.1), (333, 129, 0.07)])
def test_hip_64_bit_addressing_path_matches_oracle(hl, oracle, bg_path, monkeypatch, w, h, r_sigma):
    """Images / grids beyond the 32-bit bounds the host checks take bg_blur_slice<.., A32 = false>; HLMI_BG_NO_A32=1 selects it at
    any size (12- and 16-plane grids)."""
    monkeypatch.setenv("HLMI_BG_NO_A32", "1")
    inp = _img(w, h, seed=w + h, kind="uniform")
    got = _run(hl, inp, r_sigma)
    want = oracle.bilateral_grid(inp, r_sigma)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got.view(np.uint32) != want.view(np.uint32))} differ"


@pytest.mark.gpu
@pytest.mark.parametrize("mx,my", [(8, 16), (3, -5), (-17, 29)])
def test_hip_nonzero_min(hl, oracle, bg_path, mx, my):
    inp = _img(150, 70, seed=9, kind="smooth")
    got = _run(hl, inp, 0.1, in_min=(mx, my), out_min=(mx, my))
    want = oracle.bilateral_grid(inp, 0.1, origin=(mx, my))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_hip_output_window_inside_larger_input(hl, oracle, bg_path):
    inp = _img(200, 120, seed=4, kind="smooth")
    full = oracle.bilateral_grid(inp, 0.1)
    out = np.zeros((50, 90), np.float32)
    got = _run(hl, inp, 0.1, out=out, out_min=(37, 22))
    assert np.array_equal(got.view(np.uint32), full[22:72, 37:127].view(np.uint32))


@pytest.mark.gpu
def test_hip_special_values(hl, oracle, bg_path):
    inp = _img(64, 48, seed=1)
    inp[3, 5] = -4.0
    inp[10, 11] = 7.5
    inp[20, 20] = 1.0
    inp[21, 21] = 0.0
    got, want = _run(hl, inp, 0.1), oracle.bilateral_grid(inp, 0.1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.isfinite(got).all()
