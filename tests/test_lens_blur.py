"""lens_blur: oracle vs an independent second reading of the generator, GPU vs oracle.

Reference: apps/lens_blur/lens_blur_generator.cpp:24-152 (algorithm), :279-294 (downsample / upsample), src/Random.cpp:20-104
(random_float), src/InlineReductions.cpp:290-314 (argmin).  The oracle's header states what is assumed about the random
stream: call ids 0 and 1, definition tag = oracle_lib.lens_blur_default_tag() unless set.
"""
import functools

import numpy as np
import pytest

f32 = np.float32


def _pair(w, h, seed, shift=3):
    """A stereo-like pair: textured scene, the right view displaced by a depth-dependent disparity."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    tex = (np.sin(xx / 2.3 + seed) * np.cos(yy / 3.1) * 60 + 128 + rng.normal(0, 25, (h, w)))
    left = np.clip(np.stack([tex, np.roll(tex, 2, 0) * 0.9, tex[::-1] * 0.8]), 0, 255).astype(np.uint8)
    disp = np.where(xx > w // 2, shift + 4, shift)
    right = np.zeros_like(left)
    for c in range(3):
        for y in range(h):
            right[c, y] = left[c, y, np.clip(xx[y] - disp[y], 0, w - 1)]
    return left, right


# ---------------------------------------------------------------------------------------------------- second reading
def naive_lens_blur(left, right, slices, focus_depth, scale, samples, tag):
    """Every Func of the generator as a memoised pure function on Z^n, float32 arithmetic operator by operator."""
    _, H, W = left.shape
    _, RH, RW = right.shape
    R = int(f32(max(slices - focus_depth, focus_depth)) * f32(scale))
    fs = f32(slices)

    def L(x, y, c):
        return int(left[c, min(max(y, 0), H - 1), min(max(x, 0), W - 1)])

    def Rt(x, y, c):
        return int(right[c, min(max(y, 0), RH - 1), min(max(x, 0), RW - 1)])

    @functools.lru_cache(None)
    def cost(x, y, z):
        t = None
        for c in range(3):
            d = f32(min(abs(L(x, y, c) - Rt(x + 2 * z, y, c)), abs(L(x, y, c) - Rt(x + 2 * z + 1, y, c))))
            t = d * d if t is None else f32(t + d * d)
        return t

    @functools.lru_cache(None)
    def conf(x, y):
        a, b = f32(0), f32(0)
        for r in range(slices):
            a = f32(a + f32(cost(x, y, r) * cost(x, y, r)))
            b = f32(b + f32(cost(x, y, r) / fs))
        return f32(f32(a / fs) - f32(b * b))

    ws, hs = [W], [H]
    for i in range(1, 8):
        ws.append(ws[-1] // 2), hs.append(hs[-1] // 2)

    @functools.lru_cache(None)
    def push(i, x, y, z, c):
        if i == 0:
            return f32(cost(x, y, z) * conf(x, y)) if c == 0 else conf(x, y)
        # repeat_edge(.., {{0, w}, {0, h}}): clamp(x, 0, w - 1) = max(min(x, w - 1), 0)
        x, y = max(min(x, ws[i] - 1), 0), max(min(y, hs[i] - 1), 0)

        def downx(xx, yy):
            return f32(f32(f32(push(i - 1, 2 * xx - 1, yy, z, c) + f32(f32(3) * f32(push(i - 1, 2 * xx, yy, z, c) + push(i - 1, 2 * xx + 1, yy, z, c))))
                           + push(i - 1, 2 * xx + 2, yy, z, c)) / f32(8))
        return f32(f32(f32(downx(x, 2 * y - 1) + f32(f32(3) * f32(downx(x, 2 * y) + downx(x, 2 * y + 1)))) + downx(x, 2 * y + 2)) / f32(8))

    @functools.lru_cache(None)
    def pull(i, x, y, z, c):
        if i == 7:
            return push(7, x, y, z, c)

        def upx(xx, yy):
            return f32(f32(f32(0.25) * pull(i + 1, xx // 2 - 1 + 2 * (xx % 2), yy, z, c)) + f32(f32(0.75) * pull(i + 1, xx // 2, yy, z, c)))
        up = f32(f32(f32(0.25) * upx(x, y // 2 - 1 + 2 * (y % 2))) + f32(f32(0.75) * upx(x, y // 2)))
        return f32(f32(up * f32(0.5)) + f32(push(i, x, y, z, c) * f32(0.5)))   # lerp(up, push, 0.5)

    @functools.lru_cache(None)
    def depth(x, y):
        best, bi = f32(np.finfo(np.float32).max), 0
        for r in range(slices):
            with np.errstate(divide="ignore", invalid="ignore"):
                v = f32(pull(0, x, y, r, 0) / pull(0, x, y, r, 1))
            if v < best:
                best, bi = v, r
        return bi

    def br(x, y):
        return f32(f32(abs(depth(x, y) - focus_depth)) * f32(scale))

    def rnd(i, s, y, x):
        def rng32(v):
            return ((1040796640 * v + 1121052041) * v + 576942909) & 0xFFFFFFFF
        r = rng32(i)
        for e in (tag, s, y, x):
            r = rng32((r + e) & 0xFFFFFFFF)
        r ^= r >> 16
        return f32(np.array([(127 << 23) | (r >> 9)], np.uint32).view(np.float32)[0] - f32(1))

    out = np.zeros((3, H, W), np.float32)
    for y in range(H):
        for x in range(W):
            worst = max(max(br(x + rx, y + ry) for ry in range(-R, R + 1)) for rx in range(-R, R + 1))
            acc = [f32(L(x, y, 0)), f32(L(x, y, 1)), f32(L(x, y, 2)), f32(255)]
            for s in range(samples):
                u = int(f32(f32(f32(rnd(0, s, y, x) - f32(0.5)) * f32(2)) * worst))
                v = int(f32(f32(f32(rnd(1, s, y, x) - f32(0.5)) * f32(2)) * worst))
                u, v = min(max(u, -R), R), min(max(v, -R), R)
                sx, sy = x + u, y + v
                r2 = f32(u * u + v * v)
                take = (r2 < f32(br(x, y) * br(x, y)) or depth(sx, sy) < depth(x, y)) and r2 < f32(br(sx, sy) * br(sx, sy))
                wgt = f32(1 if take else 0)
                for c in range(3):
                    acc[c] = f32(acc[c] + f32(wgt * f32(L(sx, sy, c))))
                acc[3] = f32(acc[3] + f32(wgt * f32(255)))
            for c in range(3):
                out[c, y, x] = f32(acc[c] / acc[3])
    dmap = np.array([[depth(x, y) for x in range(-R, W + R)] for y in range(-R, H + R)], np.int32)
    return out, dmap


# ---------------------------------------------------------------------------------------------------- CPU
def test_random_float_is_the_hash_of_src_random_cpp(oracle):
    """src/Random.cpp:20-104 in Python integers, against the oracle's C version; values lie in [0, 1) with 23 random bits."""
    def rng32(v):
        return ((1040796640 * v + 1121052041) * v + 576942909) & 0xFFFFFFFF
    for args in [(0, 71, 0, 0, 0), (1, 71, 5, 17, 33), (0, 3, 31, 2559, 1535), (1, 0, 63, 1, 0)]:
        r = rng32(args[0])
        for e in args[1:]:
            r = rng32((r + e) & 0xFFFFFFFF)
        r ^= r >> 16
        want = np.array([(127 << 23) | (r >> 9)], np.uint32).view(np.float32)[0] - np.float32(1)
        got = oracle.lens_blur_random(*args)
        assert 0.0 <= got < 1.0 and np.float32(got) == want


@pytest.mark.parametrize("w,h,slices,focus,scale,samples", [(12, 9, 4, 2, 0.5, 5), (7, 10, 6, 1, 0.9, 3), (16, 8, 3, 3, 0.4, 4), (5, 4, 2, 1, 1.0, 2)])
def test_oracle_matches_naive_pure_function_evaluator(oracle, w, h, slices, focus, scale, samples):
    left, right = _pair(w, h, seed=w + h, shift=1)
    want, want_depth = naive_lens_blur(left, right, slices, focus, scale, samples, oracle.lens_blur_default_tag())
    got, got_depth = oracle.lens_blur(left, right, slices, focus, scale, samples, return_depth=True)
    assert np.array_equal(got_depth, want_depth)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_oracle_depth_finds_the_disparity_of_a_shifted_pair(oracle):
    """Sanity of the stereo stage: a right view displaced by 2 z + 1 pixels puts the depth of the interior at slice z."""
    rng = np.random.default_rng(1)
    left = rng.integers(0, 256, (3, 48, 96), dtype=np.uint8)
    right = np.zeros_like(left)
    right[:, :, 11:] = left[:, :, :-11]          # right(x + 11) = left(x): 2 z = 10 or 2 z + 1 = 11 -> z = 5
    _, depth = oracle.lens_blur(left, right, 16, 5, 0.5, 4, return_depth=True)
    R = int(np.float32(11) * np.float32(0.5))
    inner = depth[R + 8:-R - 8, R + 8:-R - 24]
    assert np.count_nonzero(inner == 5) > 0.95 * inner.size


def test_two_fma_division_of_the_cost_stack_is_the_correctly_rounded_quotient(tmp_path):
    """halide_amd/csrc/lens_blur.hip divides a cost by `slices` with a multiply and two FMAs; tests/cpp/lb_div_check.c
    checks every (cost, slices) pair the generator admits against the correctly rounded division."""
    import subprocess
    from pathlib import Path
    src = Path(__file__).parent / "cpp" / "lb_div_check.c"
    exe = tmp_path / "lb_div_check"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "0", r.stdout + r.stderr


# ---------------------------------------------------------------------------------------------------- GPU
def _run(hl, left, right, slices, focus, scale, samples, out_shape=None):
    bl, br_ = hl.Buffer(left), hl.Buffer(right)
    bo = hl.Buffer(np.zeros(out_shape or (3,) + left.shape[1:], np.float32))
    hl.lens_blur(bl, br_, slices, focus, scale, samples, bo)
    return bo.numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,slices,focus,scale,samples", [(192, 320, 32, 13, 0.5, 32), (200, 130, 32, 13, 0.5, 32), (64, 48, 8, 3, 1.0, 64),
                                                           (37, 23, 5, 2, 0.7, 7), (1, 1, 1, 1, 0.0, 1), (130, 17, 64, 32, 0.25, 16)])
def test_hip_matches_oracle(hl, oracle, w, h, slices, focus, scale, samples):
    left, right = _pair(w, h, seed=w * 3 + h, shift=3)
    got = _run(hl, left, right, slices, focus, scale, samples)
    want = oracle.lens_blur(left, right, slices, focus, scale, samples)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got != want)} of {got.size} differ"


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["HLMI_LB_NO_A32", "HLMI_LB_WCY_LAUNCH"])
def test_hip_64_bit_pyramid_kernels_and_the_separate_maximum_launch_match_the_oracle(hl, oracle, monkeypatch, switch):
    """HLMI_LB_NO_A32=1: lb_pull_multi / lb_down with size_t indices (what planes of 2^29 elements and more take);
    HLMI_LB_WCY_LAUNCH=1: the vertical maximum of the bokeh radius as its own launch (lb_wcy + lb_final<false>)."""
    monkeypatch.setenv(switch, "1")
    left, right = _pair(200, 130, seed=77, shift=3)
    got = _run(hl, left, right, 32, 13, 0.5, 32)
    want = oracle.lens_blur(left, right, 32, 13, 0.5, 32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got != want)} of {got.size} differ"


@pytest.mark.gpu
@pytest.mark.parametrize("unfused", [False, True])
@pytest.mark.parametrize("w,h,slices,samples", [(530, 70, 32, 8), (300, 41, 33, 5), (258, 36, 64, 4), (515, 19, 31, 3)])
def test_hip_front_ends_agree_with_the_oracle(hl, oracle, monkeypatch, unfused, w, h, slices, samples):
    """The fused front end (lb_cost_down + lb_depth_rc: level 0 of the push pyramid is never stored) and the staged one
    (HLMI_LB_UNFUSED=1), on widths that need several strips of uneven width and on slice counts that are / are not the
    compile-time bound of the cost stack (32, 64)."""
    if unfused:
        monkeypatch.setenv("HLMI_LB_UNFUSED", "1")
    elif (w + h) % 2:
        monkeypatch.setenv("HLMI_LB_ROWS2", "0" if slices <= 32 else "1")   # the other front-end kernel than the default
    for ndy in ("", "3", "16"):
        if ndy:
            monkeypatch.setenv("HLMI_LB_NDY", ndy)
        left, right = _pair(w, h, seed=w + h + slices, shift=5)
        got = _run(hl, left, right, slices, min(slices, 32) // 2 + 1, 0.5, samples)
        want = oracle.lens_blur(left, right, slices, min(slices, 32) // 2 + 1, 0.5, samples)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"ndy={ndy!r}: {np.count_nonzero(got != want)} of {got.size} differ"
        if unfused:
            break


@pytest.mark.gpu
def test_hip_random_tag_is_the_oracles_parameter(hl, oracle):
    """The definition tag of the reference's random_float() calls is a parameter on both sides; another tag gives other
    sample positions (another image), and the same one on both sides gives the same image."""
    import ctypes
    left, right = _pair(96, 64, seed=4)
    hl.lib.hlmi_lens_blur_get_random_tag.restype = ctypes.c_int
    assert hl.lib.hlmi_lens_blur_get_random_tag() == oracle.lens_blur_default_tag()
    base = _run(hl, left, right, 32, 13, 0.5, 32)
    try:
        hl.lib.hlmi_lens_blur_set_random_tag(5)
        other = _run(hl, left, right, 32, 13, 0.5, 32)
        assert np.array_equal(other.view(np.uint32), oracle.lens_blur(left, right, 32, 13, 0.5, 32, tag=5).view(np.uint32))
        assert not np.array_equal(other, base)
    finally:
        hl.lib.hlmi_lens_blur_set_random_tag(oracle.lens_blur_default_tag())


@pytest.mark.gpu
def test_hip_right_image_of_another_size_and_scalar_ranges(hl, oracle):
    left, _ = _pair(80, 50, seed=2)
    _, right = _pair(96, 40, seed=2)
    got = _run(hl, left, right, 16, 4, 0.5, 8)
    assert np.array_equal(got.view(np.uint32), oracle.lens_blur(left, right, 16, 4, 0.5, 8).view(np.uint32))
    for bad, code in [((0, 13, 0.5, 32), -9), ((65, 13, 0.5, 32), -10), ((32, 33, 0.5, 32), -10), ((32, 13, 1.5, 32), -10), ((32, 13, 0.5, 0), -9)]:
        with pytest.raises(hl.HalideError) as e:
            _run(hl, left, right, *bad)
        assert e.value.code == code


@pytest.mark.gpu
def test_hip_padded_strides_of_every_buffer(hl, oracle):
    """Row / plane padding of both inputs and of the output: padding bytes are neither read into the result nor written."""
    left, right = _pair(70, 46, seed=9)
    big_l = np.pad(left, ((0, 0), (0, 3), (0, 6)), constant_values=200)
    big_r = np.pad(right, ((0, 0), (0, 1), (0, 10)), constant_values=13)
    big_o = np.full((3, 46 + 2, 70 + 5), np.float32(-3), np.float32)
    bo = hl.Buffer(big_o[:, :46, :70])
    hl.lens_blur(hl.Buffer(big_l[:, :46, :70]), hl.Buffer(big_r[:, :46, :70]), 16, 5, 0.5, 12, bo)
    bo.copy_to_host()
    want = oracle.lens_blur(left, right, 16, 5, 0.5, 12)
    assert np.array_equal(np.ascontiguousarray(big_o[:, :46, :70]).view(np.uint32), want.view(np.uint32))
    assert np.all(big_o[:, 46:, :] == -3) and np.all(big_o[:, :, 70:] == -3)
