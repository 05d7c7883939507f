"""local_laplacian: oracle validation (CPU) and HIP-vs-oracle parity (GPU).

Reference algorithm: /root/reference/apps/local_laplacian/local_laplacian_generator.cpp:18-87.
Parity bar: bit-exact u16 output against the oracle (oracle <-> real Halide object is unpinned for the
float stages, see oracle/local_laplacian_oracle.c header).
"""
import functools

import numpy as np
import pytest

f32 = np.float32


# ----------------------------------------------------------------------------------------------------
# An independent, deliberately naive evaluator: every Func is a memoised pure function on Z^2,
# evaluated lazily at whatever coordinate is demanded — i.e. exactly the reference's semantics with no
# region bookkeeping at all.  Pure Python, so only for tiny images; it validates the oracle's R_j/G_j
# region logic and its evaluation order.
def naive_local_laplacian(inp, levels, alpha, beta, J=8):
    import oracle_lib
    C_, H, W = inp.shape
    K = levels
    lut = oracle_lib.ll_remap_lut(K, alpha)
    half = (K - 1) * 256
    r = f32(1.0 / 65535.0)
    # the simplifier's folded constants (src/Simplify_Div.cpp:204, src/Simplify_Mul.cpp:70; oracle header)
    C0, C1, C2 = (f32(float(r) * float(f32(c))) for c in (0.299, 0.587, 0.114))
    beta = f32(beta)
    Km1 = f32(K - 1)
    inv = f32(1.0) / Km1

    @functools.lru_cache(maxsize=None)
    def gray(x, y):
        xc, yc = min(max(x, 0), W - 1), min(max(y, 0), H - 1)
        u0, u1, u2 = (f32(inp[c, yc, xc]) for c in range(3))
        return (u0 * C0 + u1 * C1) + u2 * C2

    def g0(x, y, k):
        gr = gray(x, y)
        level = f32(k) * inv
        idx = min(max(int((gr * Km1) * f32(256.0)), 0), half)
        return (beta * (gr - level) + level) + lut[idx - 256 * k + half]

    def down(f):
        @functools.lru_cache(maxsize=None)
        def dy(x, y):
            return ((f(x, 2 * y - 1) + f32(3.0) * (f(x, 2 * y) + f(x, 2 * y + 1))) + f(x, 2 * y + 2)) * f32(0.125)

        @functools.lru_cache(maxsize=None)
        def dx(x, y):
            return ((dy(2 * x - 1, y) + f32(3.0) * (dy(2 * x, y) + dy(2 * x + 1, y))) + dy(2 * x + 2, y)) * f32(0.125)
        return dx

    def lerp(zero, one, w):
        return zero * (f32(1.0) - w) + one * w

    def up(f):
        @functools.lru_cache(maxsize=None)
        def ux(x, y):
            return lerp(f((x + 1) // 2, y), f((x - 1) // 2, y), f32((x % 2) * 2 + 1) * f32(0.25))

        @functools.lru_cache(maxsize=None)
        def uy(x, y):
            return lerp(ux(x, (y + 1) // 2), ux(x, (y - 1) // 2), f32((y % 2) * 2 + 1) * f32(0.25))
        return uy

    g = [[None] * J for _ in range(K)]
    for k in range(K):
        g[k][0] = functools.lru_cache(maxsize=None)(functools.partial(lambda x, y, k: g0(x, y, k), k=k))
        for j in range(1, J):
            g[k][j] = down(g[k][j - 1])
    inG = [gray]
    for j in range(1, J):
        inG.append(down(inG[j - 1]))
    upg = [[up(g[k][j + 1]) if j + 1 < J else None for j in range(J)] for k in range(K)]

    outG = [None] * J

    def make_out(j):
        upo = up(outG[j + 1]) if j + 1 < J else None

        @functools.lru_cache(maxsize=None)
        def o(x, y):
            level = inG[j](x, y) * Km1
            li = min(max(int(level), 0), K - 2)
            lf = level - f32(li)
            l0, l1 = g[li][j](x, y), g[li + 1][j](x, y)
            if j + 1 < J:
                l0 = l0 - upg[li][j](x, y)
                l1 = l1 - upg[li + 1][j](x, y)
            outL = (f32(1.0) - lf) * l0 + lf * l1
            return outL if j == J - 1 else upo(x, y) + outL
        return o

    for j in range(J - 1, -1, -1):
        outG[j] = make_out(j)
    out = np.zeros_like(inp)
    eps = f32(0.01)
    with np.errstate(all="ignore"):
        for y in range(H):
            for x in range(W):
                og, gr = outG[0](x, y) + eps, gray(x, y) + eps
                for c in range(3):
                    v = (f32(inp[c, y, x]) * og) / gr
                    out[c, y, x] = np.uint16(min(max(v, f32(0.0)), f32(65535.0)))
    return out


def _rand_image(w, h, seed, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.integers(0, 65536, (3, h, w), dtype=np.uint16)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (np.sin(xx / 37.0 + seed) + np.cos(yy / 23.0) + 2.2) / 4.4
    img = np.stack([base * 65535, np.roll(base, 5, 1) * 50000, base[::-1] * 42000])
    img += rng.normal(0, 600, img.shape)
    return np.clip(img, 0, 65535).astype(np.uint16)


# ---------------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("w,h,levels", [(1, 1, 8), (5, 3, 8), (9, 7, 4), (12, 10, 2)])
def test_oracle_matches_naive_pure_function_evaluator(oracle, canon0, w, h, levels):
    inp = _rand_image(w, h, seed=w * 100 + h)
    want = naive_local_laplacian(inp, levels, 1.0 / (levels - 1), 1.0)
    got = oracle.local_laplacian(inp, levels, 1.0 / (levels - 1), 1.0)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("w,h,levels,J,origin,beta", [(64, 48, 8, 8, (0, 0), 1.0), (333, 201, 8, 8, (-5, 7), 1.0), (1, 1, 2, 8, (0, 0), 1.0),
                                                       (130, 67, 9, 5, (3, -2), 0.75), (257, 129, 3, 2, (0, 0), 1.5), (77, 300, 20, 8, (11, 13), 1.0)])
def test_tuned_cpu_evaluation_equals_the_oracle(oracle, w, h, levels, J, origin, beta):
    """oracle/local_laplacian_fast_oracle.c (what bench.py's cpu_baseline times) computes the oracle's operations in the
    oracle's order under a CPU-friendly schedule: bit for bit the same image, also when called again with another size
    (its arena is kept between calls)."""
    rng = np.random.default_rng(w * 7 + h + levels)
    inp = rng.integers(0, 65536, (3, h, w), dtype=np.uint16)
    alpha = np.float32(1.0 / (levels - 1))
    want = oracle.local_laplacian(inp, levels, alpha, np.float32(beta), J=J, origin=origin)
    for _ in range(2):
        got = oracle.local_laplacian_fast(inp, levels, alpha, np.float32(beta), J=J, origin=origin)
        assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} of {got.size} differ"


def test_oracle_alpha0_beta1_is_identity_within_one_lsb(oracle):
    # remap == 0 and beta == 1 make every processed pyramid equal to the input pyramid, so the collapse
    # reconstructs gray and the recolouring returns the input (up to float rounding -> <= 1 LSB)
    inp = _rand_image(157, 93, seed=3, kind="smooth")
    out = oracle.local_laplacian(inp, 8, 0.0, 1.0)
    assert np.max(np.abs(out.astype(np.int64) - inp.astype(np.int64))) <= 1


def test_oracle_regression_digest(oracle, each_canon):
    """Regression pin of the oracle in both canonical forms (NOT a reference-derived vector: none exists, see module doc)."""
    import hashlib
    import json
    import os
    inp = _rand_image(64, 48, seed=11)
    out = oracle.local_laplacian(inp, 8, 1.0 / 7.0, 1.0)
    digest = hashlib.sha256(out.tobytes()).hexdigest()
    path = os.path.join(os.path.dirname(__file__), "golden", "oracle_digests.json")
    with open(path) as f:
        d = json.load(f)
    assert (d["_canon1"] if each_canon else d)["local_laplacian_64x48_seed11_K8"] == digest


# ---------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_dpp_wave_shift_probe(hl):
    """The strip kernels exchange edge columns between neighbouring lanes with DPP wave_shr:1 / wave_shl:1."""
    import ctypes
    fn = hl.lib.hlmi_debug_dpp_probe
    fn.restype = ctypes.c_int
    assert fn() == 1


@pytest.mark.gpu
def test_shared_reciprocal_division_is_correctly_rounded(hl):
    """ll_up0f divides the three colour numerators by one denominator through a shared reciprocal
    (div3_by); the quotients must equal IEEE `/` bit for bit over the operand range of the pipeline:
    d = gray + 0.01 in [0.01, 1.01], n = u16 * (outG + 0.01)."""
    import ctypes
    fn = hl.lib.hlmi_debug_div3_check
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    rng = np.random.default_rng(5)
    count = 1 << 24
    for trial in range(4):
        d = (rng.random(count, dtype=np.float32) * np.float32(1.0) + np.float32(0.01)).astype(np.float32)
        if trial == 0:
            n = (rng.integers(0, 65536, count).astype(np.float32) * (rng.random(count, dtype=np.float32) * 1.3 - 0.1)
                 ).astype(np.float32)
        elif trial == 1:   # quotients next to integers (the u16 truncation boundary)
            n = (rng.integers(0, 65536, count).astype(np.float32) * d).astype(np.float32)
        elif trial == 2:   # tiny and zero numerators
            n = (rng.random(count, dtype=np.float32) * np.float32(1e-6)).astype(np.float32)
            n[::7] = 0.0
        else:              # wide dynamic range
            n = np.exp(rng.uniform(-40, 40, count)).astype(np.float32)
            d = np.exp(rng.uniform(np.log(0.005), np.log(4.0), count)).astype(np.float32)
        bad = fn(n.ctypes.data, d.ctypes.data, count)
        assert bad == 0, f"trial {trial}: {bad} quotients differ from IEEE division"


def _run_hip(hl, inp, levels, alpha, beta, out_arr=None):
    a = hl.Buffer(inp)
    o = hl.Buffer(np.zeros_like(inp) if out_arr is None else out_arr)
    hl.local_laplacian(a, levels, alpha, beta, o)
    assert o.device_dirty and not a.host_dirty
    res = o.numpy()
    a.device_free()
    o.device_free()
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,levels,kind", [
    (1, 1, 8, "uniform"), (2, 3, 8, "uniform"), (37, 23, 8, "uniform"), (128, 16, 8, "uniform"),
    (129, 17, 8, "smooth"), (640, 480, 8, "smooth"), (333, 777, 8, "uniform"), (200, 120, 2, "uniform"),
    (200, 120, 4, "smooth"), (200, 120, 15, "uniform"), (200, 120, 16, "uniform"), (96, 64, 20, "smooth"),
    # beyond 32: the reference declares no upper bound for `levels` (generator :13)
    (96, 64, 33, "uniform"), (64, 48, 40, "smooth"),
])
def test_hip_matches_oracle(hl, oracle, on_stream, w, h, levels, kind):
    inp = _rand_image(w, h, seed=w + 7 * h + levels, kind=kind)
    alpha, beta = 1.0 / (levels - 1), 1.0
    got = _run_hip(hl, inp, levels, alpha, beta)
    want = oracle.local_laplacian(inp, levels, alpha, beta)
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} of {got.size} differ"


@pytest.mark.gpu
@pytest.mark.parametrize("fuse_from,upchain_from", [(8, 0), (6, 0), (5, 0), (4, 0), (4, 3), (4, 2), (4, 1), (3, 0)])
@pytest.mark.parametrize("w,h,origin", [(640, 480, (0, 0)), (1000, 300, (0, 0)), (301, 203, (17, 33)), (150, 90, (-6, 2))])
def test_hip_pyramid_levels_match_oracle(hl, oracle, monkeypatch, w, h, origin, fuse_from, upchain_from):
    """Every outGPyramid level (coarse to fine) must be bit-identical to the oracle's: localises a mismatch
    to the down chain (level 7 wrong), one up step, or the final recolouring.  fuse_from = S: levels >= S are
    handled by the two multi-level kernels (ll_down_multi / ll_up_multi), which materialise outGPyramid[S] but
    not the coarser ones; 8 = one launch per level."""
    monkeypatch.setenv("HLMI_LL_FUSE_FROM", str(fuse_from))
    # upchain_from = SU: outGPyramid[J-1] .. outGPyramid[SU] are collapsed by ONE ll_up_multi launch (0: SU = S)
    monkeypatch.setenv("HLMI_LL_UPCHAIN_FROM", str(upchain_from))
    if upchain_from:
        fuse_from = upchain_from
    inp = _rand_image(w, h, seed=w + h, kind="smooth")
    a = hl.Buffer(inp).set_min(origin[0], origin[1], 0)
    o = hl.Buffer(np.zeros_like(inp)).set_min(origin[0], origin[1], 0)
    hl.local_laplacian(a, 8, 1.0 / 7, 1.0, o)
    bad = []
    for level in range(min(7, fuse_from), 0, -1):
        got = hl.debug_local_laplacian_outg(level)
        want = oracle.local_laplacian_outg(inp, 8, 1.0 / 7, 1.0, level, origin=origin)
        assert got.shape == want.shape
        if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
            ys, xs = np.nonzero(got.view(np.uint32) != want.view(np.uint32))
            bad.append((level, len(ys), int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())))
    assert not bad, f"(level, #bad, xmin, xmax, ymin, ymax): {bad}"
    assert np.array_equal(o.numpy(), oracle.local_laplacian(inp, 8, 1.0 / 7, 1.0, origin=origin))


@pytest.mark.gpu
@pytest.mark.parametrize("exch,units", [(1, 0), (0, 0), (1, 48), (1, 4096), (0, 300)])
@pytest.mark.parametrize("w,h,origin", [(256, 200, (0, 0)), (256, 200, (1, 0)), (512, 131, (2, 5)), (260, 97, (3, -7)),
                                        (1024, 64, (-2, 1)), (128, 333, (-1, -1)), (8, 40, (0, 3)), (2048, 36, (4, 2))])
def test_hip_fused_levels_1_and_2_match_oracle(hl, oracle, monkeypatch, w, h, origin, exch, units):
    """ll_down01f (levels 1 AND 2 from the input in one walk) on vectorisable inputs (width % 4 == 0): every parity
    of the level-0 / level-1 storage origins (origin x = 0..3 mod 4), both seam treatments (HLMI_LL_D01_EXCH: rows
    exchanged through LDS inside a workgroup / every unit walks its own two extra rows) and several unit heights
    (HLMI_LL_UNITS0 = resident-wave target; small = tall units, large = 2-row units and idle waves).  All outGPyramid
    levels and the result must equal the oracle's bit for bit."""
    monkeypatch.setenv("HLMI_LL_D01_EXCH", str(exch))
    if units:
        monkeypatch.setenv("HLMI_LL_UNITS0", str(units))
    inp = _rand_image(w, h, seed=3 * w + h + origin[0], kind="smooth" if (w + h) & 1 else "uniform")
    a = hl.Buffer(inp).set_min(origin[0], origin[1], 0)
    o = hl.Buffer(np.zeros_like(inp)).set_min(origin[0], origin[1], 0)
    hl.local_laplacian(a, 8, 1.0 / 7, 1.0, o)
    bad = []
    for level in range(3, 0, -1):   # outGPyramid[3] is the coarsest level the default launch chain materialises
        got = hl.debug_local_laplacian_outg(level)
        want = oracle.local_laplacian_outg(inp, 8, 1.0 / 7, 1.0, level, origin=origin)
        assert got.shape == want.shape
        if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
            ys, xs = np.nonzero(got.view(np.uint32) != want.view(np.uint32))
            bad.append((level, len(ys), int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())))
    assert not bad, f"(level, #bad, xmin, xmax, ymin, ymax): {bad}"
    assert np.array_equal(o.numpy(), oracle.local_laplacian(inp, 8, 1.0 / 7, 1.0, origin=origin))


@pytest.mark.gpu
@pytest.mark.parametrize("alpha,beta", [(0.0, 1.0), (1.0, 1.0), (0.4, 0.0), (-0.3, 1.7), (2.0, 0.5)])
def test_hip_matches_oracle_parameter_sweep(hl, oracle, alpha, beta):
    inp = _rand_image(301, 203, seed=5, kind="smooth")
    got = _run_hip(hl, inp, 8, alpha, beta)
    want = oracle.local_laplacian(inp, 8, alpha, beta)
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_hip_extreme_inputs(hl, oracle):
    for fill in (0, 65535, 1):
        inp = np.full((3, 50, 70), fill, np.uint16)
        assert np.array_equal(_run_hip(hl, inp, 8, 1.0 / 7, 1.0), oracle.local_laplacian(inp, 8, 1.0 / 7, 1.0))
    # checkerboard of extremes
    inp = np.zeros((3, 64, 64), np.uint16)
    inp[:, ::2, ::2] = 65535
    inp[:, 1::2, 1::2] = 65535
    assert np.array_equal(_run_hip(hl, inp, 8, 1.0 / 7, 1.0), oracle.local_laplacian(inp, 8, 1.0 / 7, 1.0))


@pytest.mark.gpu
@pytest.mark.parametrize("mx,my", [(128, 256), (17, 33), (-5, -130)])
def test_hip_padded_strides_and_nonzero_min(hl, oracle, mx, my):
    """Row padding (stride[1] > extent[0]) must not matter and padding bytes must stay untouched; a
    non-zero min shifts the ABSOLUTE coordinates the 2x-1 pyramid taps are taken at, so results are
    translation-invariant only for shifts that are multiples of 2^(J-1) = 128 — for other mins compare
    against the oracle evaluated at the same origin."""
    w, h = 150, 90
    inp = _rand_image(w, h, seed=21)
    big_in = np.zeros((3, h + 4, w + 10), np.uint16)
    big_in[:, :h, :w] = inp
    big_out = np.full((3, h + 2, w + 6), 0xABCD, np.uint16)
    a = hl.Buffer(big_in[:, :h, :w]).set_min(mx, my, 0)
    o = hl.Buffer(big_out[:, :h, :w]).set_min(mx, my, 0)
    hl.local_laplacian(a, 8, 1.0 / 7, 1.0, o)
    o.copy_to_host()
    want = oracle.local_laplacian(inp, 8, 1.0 / 7, 1.0, origin=(mx, my))
    if (mx, my) == (128, 256):
        assert np.array_equal(want, oracle.local_laplacian(inp, 8, 1.0 / 7, 1.0))
    assert np.array_equal(big_out[:, :h, :w], want)
    assert np.all(big_out[:, h:, :] == 0xABCD) and np.all(big_out[:, :, w:] == 0xABCD)


@pytest.mark.gpu
def test_hip_output_crop_of_larger_input(hl, oracle):
    """Output region strictly inside the input: taps clamp at the INPUT's edges (repeat_edge acts on the
    buffer passed in, BoundaryConditions.h:160-168), so the crop of the full result is expected."""
    w, h = 160, 100
    inp = _rand_image(w, h, seed=8, kind="smooth")
    full = oracle.local_laplacian(inp, 8, 1.0 / 7, 1.0)
    x0, y0, cw, chh = 22, 10, 77, 51
    a = hl.Buffer(inp)
    out = np.zeros((3, chh, cw), np.uint16)
    o = hl.Buffer(out).set_min(x0, y0, 0)
    hl.local_laplacian(a, 8, 1.0 / 7, 1.0, o)
    assert np.array_equal(o.numpy(), full[:, y0:y0 + chh, x0:x0 + cw])


@pytest.mark.gpu
def test_hip_full_4k_matches_oracle_and_is_deterministic(hl, oracle, on_stream):
    """BASELINE config: 3840x2160, 8 levels — the oracle finishes in seconds, so compare directly.  On the device-wide stream and
    on a CU partition (the defaults bench.py times: HLMI_LL_NT / HLMI_LL_FUSE_UP2 on, taller units)."""
    inp = _rand_image(3840, 2160, seed=0)
    a = hl.Buffer(inp)
    o = hl.Buffer(np.zeros_like(inp))
    hl.local_laplacian(a, 8, 1.0 / 7, 1.0, o)
    first = o.numpy().copy()
    hl.local_laplacian(a, 8, 1.0 / 7, 1.0, o)  # inputs now device-resident, no re-upload
    assert np.array_equal(o.numpy(), first)
    want = oracle.local_laplacian(inp, 8, 1.0 / 7, 1.0)
    assert np.array_equal(first, want), f"{np.count_nonzero(first != want)} differ"
    # size-independent property: alpha=0, beta=1 reproduces the input within 1 LSB
    hl.local_laplacian(a, 8, 0.0, 1.0, o)
    assert np.max(np.abs(o.numpy().astype(np.int64) - inp.astype(np.int64))) <= 1
    a.device_free()
    o.device_free()


@pytest.mark.gpu
def test_hip_8k_smooth_frame_matches_oracle(hl, oracle):
    """SURVEY.md §8(d) also names 7680x4320: 33 Mpx, 400 MB of pyramid — the level table, the 32-bit lane offsets of
    ll_up0f and the multi-level kernels at twice every extent.  A smooth frame (the data-dependent plane choice is
    locally constant, unlike uniform noise)."""
    inp = _rand_image(7680, 4320, seed=2, kind="smooth")
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
    hl.local_laplacian(a, 8, 1.0 / 7, 1.0, o)
    got = o.numpy()
    want = oracle.local_laplacian(inp, 8, 1.0 / 7, 1.0)
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} differ"
    a.device_free()
    o.device_free()


# ---- HIP-graph replay of the launch chain (second call with the same buffers / shape / parameters captures, later ones replay)
@pytest.mark.gpu
@pytest.mark.parametrize("graph", ["1", "0"])
def test_hip_repeated_calls_replay_a_graph_and_follow_new_contents(hl, oracle, monkeypatch, graph):
    """Five calls into the SAME pair of buffers: the first is eager, the second captures, the rest replay.  Between calls the
    input's contents, then a parameter, then a switch change: every result must equal the oracle's for what that call was
    given (a replayed graph may only ever repeat launches the eager path would have issued with identical arguments)."""
    monkeypatch.setenv("HLMI_LL_GRAPH", graph)
    rng = np.random.default_rng(77)
    w, h = 512, 208
    imgs = [rng.integers(0, 65536, (3, h, w), dtype=np.uint16) for _ in range(3)]
    a, o = hl.Buffer(imgs[0].copy()), hl.Buffer(np.zeros((3, h, w), np.uint16))
    alpha = np.float32(1.0 / 7.0)

    def call(img, levels=8, beta=1.0):
        a.array[...] = img          # same host and device allocation, new contents
        a.set_host_dirty()
        hl.local_laplacian(a, levels, alpha, beta, o)
        return o.numpy().copy()
    want = [oracle.local_laplacian(im, 8, alpha, 1.0) for im in imgs]
    assert np.array_equal(call(imgs[0]), want[0])          # eager
    assert np.array_equal(call(imgs[0]), want[0])          # captured + launched
    assert np.array_equal(call(imgs[1]), want[1])          # replayed on new contents
    assert np.array_equal(call(imgs[2]), want[2])
    # another parameter value: a different key (eager again), then back to the replayed one
    assert np.array_equal(call(imgs[2], beta=0.5), oracle.local_laplacian(imgs[2], 8, alpha, 0.5))
    assert np.array_equal(call(imgs[1]), want[1])
    # a switch that changes the launch chain must not hit the graph captured without it
    monkeypatch.setenv("HLMI_LL_FUSE_FROM", "8")
    assert np.array_equal(call(imgs[0]), want[0])
    assert np.array_equal(call(imgs[0]), want[0])
    monkeypatch.delenv("HLMI_LL_FUSE_FROM")
    assert np.array_equal(call(imgs[2]), want[2])
    # the pyramid debug hook sees the same bookkeeping after a replay as after an eager call
    got4 = hl.debug_local_laplacian_outg(4)
    assert got4.size > 0 and np.isfinite(got4).all()
    a.device_free()
    o.device_free()


@pytest.mark.gpu
def test_hip_graph_replay_on_caller_streams(hl, oracle, monkeypatch):
    """Two caller streams, one frame pair each, interleaved calls: keys differ by stream (and workspace), each stream
    replays its own graph."""
    monkeypatch.setenv("HLMI_LL_GRAPH", "1")      # opt-in: measured to buy nothing on MI355X (local_laplacian.hip)
    hip = hl.hip_runtime()
    import ctypes as C
    streams = []
    for _ in range(2):
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
        streams.append(s)
    rng = np.random.default_rng(78)
    imgs = [rng.integers(0, 65536, (3, 120, 256), dtype=np.uint16) for _ in range(2)]
    bufs = [(hl.Buffer(im), hl.Buffer(np.zeros_like(im))) for im in imgs]
    alpha = np.float32(1.0 / 7.0)
    for _ in range(4):
        for s, (a, o) in zip(streams, bufs):
            hl.set_stream(s.value)
            hl.local_laplacian(a, 8, alpha, 1.0, o)
    hl.set_stream(None)
    for (a, o), im in zip(bufs, imgs):
        assert np.array_equal(o.numpy(), oracle.local_laplacian(im, 8, alpha, 1.0))
        a.device_free()
        o.device_free()
    for s in streams:
        hip.hipStreamSynchronize(s)
        hip.hipStreamDestroy(s)


# ---- round 4: the re-cut dataflow (ll_down01e emits outLPyramid[0] + three level-1 planes, ll_up0h collapses) against the
# materialised-pyramid pair (ll_down01f / ll_up0f, HLMI_LL_EMIT=0)
@pytest.mark.gpu
@pytest.mark.parametrize("emit", ["1", "0"])
@pytest.mark.parametrize("w,h,origin,kind,beta", [(1500, 334, (0, 0), "smooth", 1.0), (776, 250, (-4, 5), "uniform", 1.0), (1024, 131, (2, -3), "uniform", 0.7),
                                                   (260, 97, (6, 1), "smooth", 1.0), (3840, 2160, (0, 0), "uniform", 1.0)])
def test_hip_emit_and_materialised_dataflows_match_oracle(hl, oracle, monkeypatch, emit, w, h, origin, kind, beta):
    """Both dataflows are the same pure functions: bit-identical results, and (small cases) every outGPyramid level.  The frame
    is preceded by a DIFFERENT frame through the same workspace: outLPyramid[0] rows or level-1 planes that a unit fails to
    emit would hold the other frame's values."""
    monkeypatch.setenv("HLMI_LL_EMIT", emit)
    monkeypatch.setenv("HLMI_LL_FUSE_UP2", "1")   # outGPyramid[2] only in LDS tiles of ll_up0h (what frame-queue streams run)
    other = _rand_image(w, h, seed=w + 3 * h + 2, kind="uniform" if kind == "smooth" else "smooth")
    inp = _rand_image(w, h, seed=w + h + 31, kind=kind)
    for img in (other, inp):
        a = hl.Buffer(img).set_min(origin[0], origin[1], 0)
        o = hl.Buffer(np.zeros_like(img)).set_min(origin[0], origin[1], 0)
        hl.local_laplacian(a, 8, 1.0 / 7, beta, o)
    if w * h < 1 << 20:
        for level in range(3, 0, -1):   # outGPyramid[3] is the coarsest level the default launch chain materialises
            got = hl.debug_local_laplacian_outg(level)
            want = oracle.local_laplacian_outg(inp, 8, 1.0 / 7, beta, level, origin=origin)
            assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"level {level}"
    assert np.array_equal(o.numpy(), oracle.local_laplacian(inp, 8, 1.0 / 7, beta, origin=origin))


@pytest.mark.gpu
@pytest.mark.parametrize("units,ru,exch,nt", [(0, 0, "1", "0"), (1024, 8, "1", "1"), (300, 32, "1", "0"), (4096, 3, "1", "1"), (600, 16, "0", "1")])
def test_hip_emit_launch_geometries_match_oracle(hl, oracle, monkeypatch, units, ru, exch, nt):
    """Unit heights (ll_down01e), tile heights (ll_up0h), the seam treatment and the non-temporal variants (what a CU-partitioned
    stream runs) only change who computes what and how it travels."""
    monkeypatch.setenv("HLMI_LL_D01_EXCH", exch)
    monkeypatch.setenv("HLMI_LL_NT", nt)
    monkeypatch.setenv("HLMI_LL_FUSE_UP2", nt)   # the level-2 collapse inside ll_up0h: the other thing a partitioned stream switches on
    if units:
        monkeypatch.setenv("HLMI_LL_UNITS0", str(units))
    if ru:
        monkeypatch.setenv("HLMI_LL_RU", str(ru))
    for (w, h, origin) in [(2048, 700, (0, 0)), (520, 333, (-2, 7)), (260, 4, (0, 0)), (132, 2, (2, 1))]:
        inp = _rand_image(w, h, seed=w + h + units, kind="uniform")
        a = hl.Buffer(inp).set_min(origin[0], origin[1], 0)
        o = hl.Buffer(np.zeros_like(inp)).set_min(origin[0], origin[1], 0)
        hl.local_laplacian(a, 8, 1.0 / 7, 1.0, o)
        assert np.array_equal(o.numpy(), oracle.local_laplacian(inp, 8, 1.0 / 7, 1.0, origin=origin))


@pytest.mark.gpu
def test_hip_ll_mid_one_launch_for_levels_5_to_7_and_the_collapse_matches_oracle(hl, oracle, monkeypatch, on_stream):
    """HLMI_LL_FUSE_MID=1 (opt-in, round 6): ll_down_multi and ll_up_multi as ONE launch, levels 5-7 handed from its producer blocks to
    its consumer blocks through agent-coherent stores / loads and a count.  DIFFERENT frames go through the same workspace back to
    back (a consumer that read a stale copy of the previous frame's levels would show here), on the device's stream and on a frame
    queue, at sizes where the default chain is the five-launch one (levels 5-7 exist) and at sizes where a level is a single row."""
    monkeypatch.setenv("HLMI_LL_FUSE_MID", "1")
    for (w, h) in [(1920, 1080), (520, 332), (2048, 64), (260, 40)]:
        for seed in range(3):
            inp = _rand_image(w, h, seed=100 * seed + w, kind="uniform" if seed != 1 else "smooth")
            a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
            hl.local_laplacian(a, 8, 1.0 / 7, 1.0, o)
            assert np.array_equal(o.numpy(), oracle.local_laplacian(inp, 8, 1.0 / 7, 1.0)), (w, h, seed)


# ---- round 5
@pytest.mark.gpu
def test_hip_eight_4k_frames_in_flight_on_partitioned_and_plain_streams(hl, oracle):
    """Eight 4K frames in flight on four frame-queue streams and four plain streams at once (every stream has its own
    workspace; the launches' workgroups compete for the same compute units): every frame equals the oracle's, or — for the five
    frames the oracle is not run on — the same call alone on the device's own stream."""
    hip = hl.hip_runtime()
    import ctypes as C
    plain = []
    for _ in range(4):
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
        plain.append(s)
    streams = [hl.partition_stream(p, 4) for p in range(4)] + [s.value for s in plain]
    imgs = [_rand_image(3840, 2160, seed=40 + i, kind="uniform" if i & 1 else "smooth") for i in range(8)]
    want = [oracle.local_laplacian(im, 8, 1.0 / 7, 1.0) for im in imgs[:3]]
    bufs = [(hl.Buffer(im), hl.Buffer(np.zeros_like(im))) for im in imgs]
    for _ in range(6):
        for s, (a, o) in zip(streams, bufs):
            hl.set_stream(s)
            hl.local_laplacian(a, 8, 1.0 / 7, 1.0, o)
    hl.set_stream(None)
    first = [o.numpy().copy() for (_, o) in bufs]
    for k in range(3):
        assert np.array_equal(first[k], want[k]), f"frame {k}"
    # the frames without an oracle result: the same call alone on the device's own stream
    for k in range(3, 8):
        a, o = bufs[k]
        hl.local_laplacian(a, 8, 1.0 / 7, 1.0, o)
        assert np.array_equal(o.numpy(), first[k]), f"frame {k} differs between the concurrent and the lone call"
    for a, o in bufs:
        a.device_free()
        o.device_free()
    for s in plain:
        hip.hipStreamSynchronize(s)
        hip.hipStreamDestroy(s)
