"""nl_means (BASELINE configs[3]): 7x7 search / 7x7 patch, f32 1920x1080x3, sigma=0.12.
Reference algorithm: /root/reference/apps/nl_means/nl_means_generator.cpp:24-63.  GPU must equal the oracle
bit-for-bit (stated tolerance <= 1 ulp; the kernels keep the oracle's summation orders)."""
import numpy as np
import pytest

f32 = np.float32


def naive_nl_means(inp, patch, search, sigma):
    """Independent restatement from the generator text; pure Python, tiny images only."""
    import functools
    import oracle_lib
    _, H, W = inp.shape
    inv = f32(-1.0) / (((f32(sigma) * f32(sigma)) * f32(patch)) * f32(patch))

    def cl(x, y, c):
        return f32(inp[min(max(c, 0), 2), min(max(y, 0), H - 1), min(max(x, 0), W - 1)])

    @functools.lru_cache(maxsize=None)
    def d(x, y, dx, dy):
        acc = f32(0)
        for c in range(3):
            t = cl(x, y, c) - cl(x + dx, y + dy, c)
            acc = acc + t * t
        return acc

    @functools.lru_cache(maxsize=None)
    def bdy(x, y, dx, dy):
        acc = f32(0)
        for p in range(-(patch // 2), -(patch // 2) + patch):
            acc = acc + d(x, y + p, dx, dy)
        return acc

    def bd(x, y, dx, dy):
        acc = f32(0)
        for p in range(-(patch // 2), -(patch // 2) + patch):
            acc = acc + bdy(x + p, y, dx, dy)
        return acc

    out = np.zeros_like(inp)
    s0 = -(search // 2)
    for y in range(H):
        for x in range(W):
            s = [f32(0)] * 4
            for sy in range(s0, s0 + search):
                for sx in range(s0, s0 + search):
                    w = f32(oracle_lib.fast_exp(float(bd(x, y, sx, sy) * inv)))
                    for c in range(3):
                        s[c] = s[c] + w * cl(x + sx, y + sy, c)
                    s[3] = s[3] + w * f32(1.0)
            for c in range(3):
                out[c, y, x] = max(min(s[c] / s[3], f32(1.0)), f32(0.0))
    return out


def _img(w, h, seed, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((3, h, w), dtype=np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    s = (np.sin(xx / 13.0 + seed) + np.cos(yy / 9.0)) / 4 + 0.5
    img = np.stack([s, np.roll(s, 3, 1) * 0.8, s[::-1] * 0.6]) + rng.normal(0, 0.04, (3, h, w))
    return img.clip(0, 1).astype(np.float32)


@pytest.mark.parametrize("w,h,patch,search", [(6, 5, 7, 7), (7, 4, 3, 5), (5, 5, 4, 2)])
def test_oracle_matches_naive_restatement(oracle, canon0, w, h, patch, search):
    inp = _img(w, h, seed=w * h, kind="smooth")
    got = oracle.nl_means(inp, patch, search, 0.12)
    want = naive_nl_means(inp, patch, search, 0.12)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_oracle_denoises(oracle):
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:60, 0:80]
    s = np.stack([(np.sin(xx / 20) + np.cos(yy / 15)) / 4 + 0.5] * 3).astype(np.float32)
    n = (s + rng.normal(0, 0.05, s.shape)).clip(0, 1).astype(np.float32)
    r = oracle.nl_means(n, 7, 7, 0.12)
    assert np.abs(r - s).mean() < 0.4 * np.abs(n - s).mean()
    const = np.full((3, 20, 30), 0.4, np.float32)
    assert np.max(np.abs(oracle.nl_means(const, 7, 7, 0.12) - const)) < 1e-6


def _run(hl, inp, patch, search, sigma, out=None, in_min=None, out_min=None):
    a = hl.Buffer(inp)
    o = hl.Buffer(np.zeros_like(inp) if out is None else out)
    if in_min:
        a.set_min(*in_min)
    if out_min:
        o.set_min(*out_min)
    hl.nl_means(a, patch, search, sigma, o)
    return o.numpy()


def _eq(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,kind", [(1920, 1080, "uniform"), (1920, 1080, "smooth"), (1, 1, "uniform"), (58, 64, "smooth"),
                                      (59, 65, "smooth"), (130, 77, "uniform"), (5, 200, "smooth")])
def test_hip_7x7_matches_oracle(hl, oracle, w, h, kind):
    inp = _img(w, h, seed=w + h, kind=kind)
    got, want = _run(hl, inp, 7, 7, 0.12), oracle.nl_means(inp, 7, 7, 0.12)
    assert _eq(got, want), f"{np.count_nonzero(got.view(np.uint32) != want.view(np.uint32))} of {got.size} differ"


@pytest.mark.gpu
@pytest.mark.parametrize("patch,search,sigma", [(3, 5, 0.12), (5, 3, 0.2), (1, 1, 0.12), (8, 6, 0.1), (7, 9, 0.05)])
def test_hip_generic_sizes_match_oracle(hl, oracle, patch, search, sigma):
    inp = _img(97, 61, seed=patch * 10 + search, kind="smooth")
    assert _eq(_run(hl, inp, patch, search, sigma), oracle.nl_means(inp, patch, search, sigma))


@pytest.mark.gpu
def test_hip_min_and_window(hl, oracle):
    inp = _img(140, 90, seed=2, kind="smooth")
    full = oracle.nl_means(inp, 7, 7, 0.12)
    # non-zero min on x/y: the algorithm has no absolute-coordinate dependence
    assert _eq(_run(hl, inp, 7, 7, 0.12, in_min=(-7, 19, 0), out_min=(-7, 19, 0)), full)
    # output window inside a larger input: taps clamp at the INPUT's edges
    out = np.zeros((3, 40, 66), np.float32)
    got = _run(hl, inp, 7, 7, 0.12, out=out, out_min=(30, 25, 0))
    assert _eq(got, full[:, 25:65, 30:96])


@pytest.mark.gpu
def test_hip_rejects_wrong_channel_extent(hl):
    inp = _img(16, 16, 1)
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros((2, 16, 16), np.float32))
    with pytest.raises(hl.HalideError) as e:
        hl.nl_means(a, 7, 7, 0.12, o)
    assert e.value.code == -8  # non_local_means.dim(2).set_bounds(0, 3), generator :68
