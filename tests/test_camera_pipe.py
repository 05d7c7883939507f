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



def _numpy_camera_pipe(raw, matrix, curve, strength, out_w, out_h):
    """Independent restatement of the integer stages (generator :14-33, :47-152, :240-299, :346-404, :406-413) with numpy
    rolls on the whole raw frame; wrapped borders never reach the output region (>= 5 quads of margin, <= 4 of reach).
    matrix, curve and strength are the set-up values (checked separately in test_oracle_setup_values)."""
    u16, u32, i32 = np.uint16, np.uint32, np.int32
    R = raw.astype(u16)
    sh = lambda a, dx, dy: np.roll(a, (-dy, -dx), (0, 1))                 # sh(a, dx, dy)[y, x] = a[y + dy, x + dx]
    avg = lambda a, b: ((a.astype(u32) + b + 1) // 2).astype(a.dtype)
    absd = lambda a, b: np.where(a > b, a - b, b - a).astype(u16)
    a = np.maximum(np.maximum(sh(R, -2, 0), sh(R, 2, 0)), np.maximum(sh(R, 0, -2), sh(R, 0, 2)))
    den = np.minimum(R, a)                                                # clamp(input, 0, a), :246
    g_gr, r_r, b_b, g_gb = den[0::2, 0::2], den[0::2, 1::2], den[1::2, 0::2], den[1::2, 1::2]
    g_r = np.where(absd(sh(g_gr, 1, 0), g_gr) < absd(sh(g_gb, 0, -1), g_gb), avg(sh(g_gr, 1, 0), g_gr), avg(sh(g_gb, 0, -1), g_gb))
    g_b = np.where(absd(sh(g_gb, -1, 0), g_gb) < absd(sh(g_gr, 0, 1), g_gr), avg(sh(g_gb, -1, 0), g_gb), avg(sh(g_gr, 0, 1), g_gr))
    r_gr = (g_gr - avg(g_r, sh(g_r, -1, 0))) + avg(sh(r_r, -1, 0), r_r)   # u16 arithmetic wraps
    b_gr = (g_gr - avg(g_b, sh(g_b, 0, -1))) + avg(b_b, sh(b_b, 0, -1))
    r_gb = (g_gb - avg(g_r, sh(g_r, 0, 1))) + avg(r_r, sh(r_r, 0, 1))
    b_gb = (g_gb - avg(g_b, sh(g_b, 1, 0))) + avg(b_b, sh(b_b, 1, 0))
    rp = (g_b - avg(g_r, sh(g_r, -1, 1))) + avg(r_r, sh(r_r, -1, 1))
    rn = (g_b - avg(sh(g_r, -1, 0), sh(g_r, 0, 1))) + avg(sh(r_r, -1, 0), sh(r_r, 0, 1))
    r_b = np.where(absd(r_r, sh(r_r, -1, 1)) < absd(sh(r_r, -1, 0), sh(r_r, 0, 1)), rp, rn)
    bp = (g_r - avg(g_b, sh(g_b, 1, -1))) + avg(b_b, sh(b_b, 1, -1))
    bn = (g_r - avg(sh(g_b, 1, 0), sh(g_b, 0, -1))) + avg(sh(b_b, 1, 0), sh(b_b, 0, -1))
    b_r = np.where(absd(b_b, sh(b_b, 1, -1)) < absd(sh(b_b, 1, 0), sh(b_b, 0, -1)), bp, bn)

    def interleave(gr, r, b, gb):                                         # :24-33, :133-138
        out = np.empty(R.shape, u16)
        out[0::2, 0::2], out[0::2, 1::2], out[1::2, 0::2], out[1::2, 1::2] = gr, r, b, gb
        return out.view(np.int16).astype(i32)                             # reinterpret as signed (:143)
    ir, ig, ib = interleave(r_gr, r_r, r_b, r_gb), interleave(g_gr, g_r, g_b, g_gb), interleave(b_gr, b_r, b_b, b_gb)
    m = matrix.astype(i32)
    curved = []
    for c in range(3):
        v = (((m[c, 3] + m[c, 0] * ir) + m[c, 1] * ig) + m[c, 2] * ib) >> 8   # floor division by 256 (:287-289)
        v = v.astype(np.int16)
        curved.append(curve[np.clip(v, 0, 1023)].astype(np.uint8))           # :346
    out = np.zeros((3, out_h, out_w), np.uint8)
    wrap16 = lambda v: ((v + 32768) % 65536 - 32768)
    for c in range(3):
        p = curved[c]
        uy = avg(avg(sh(p, 0, -1), sh(p, 0, 1)), p)                      # blur121 (:20-22, :386-390)
        un = avg(avg(sh(uy, -1, 0), sh(uy, 1, 0)), uy)
        mask = p.astype(i32) - un.astype(i32)
        q = wrap16(mask * int(strength)) >> 5                             # int16 x uint8 -> int16 wraps; floor / 32
        v = wrap16(p.astype(i32) + q)
        out[c] = np.clip(v, 0, 255).astype(np.uint8)[12:12 + out_h, 16:16 + out_w]   # shifted(x, y) = input(x + 16, y + 12)
    return out


@pytest.mark.parametrize("kind,seed", [("scene", 1), ("uniform", 2), ("full", 3)])
def test_oracle_matches_numpy_restatement(oracle, kind, seed):
    """The C oracle against a second, array-at-a-time reading of the generator (integer stages, bit for bit).  "full"
    feeds 16-bit raw values: the correction terms of the demosaic wrap, the matrix product exceeds int16."""
    raw = _raw(200, 152, seed, kind)
    m, curve, s = oracle.camera_pipe_setup(M3200, M7000, 3700.0, 2.0, 50.0, 1.0, 25, 1023)
    want = _numpy_camera_pipe(raw, m, curve, s, 160, 120)
    got = oracle.camera_pipe(raw, M3200, M7000, 3700.0, 2.0, 50.0, 1.0, 25, 1023, 160, 120)
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} of {got.size} differ"


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
def test_hip_setup_block_follows_the_matrices_and_scalars(hl, oracle):
    """The set-up block (colour matrix, curve, strength) is cached per (matrix allocations + versions, scalars): resident
    matrices and unchanged scalars skip cp_setup; a matrix the caller rewrites and marks host_dirty, other scalars, or a
    re-allocated matrix at the same address must produce the new result."""
    raw = _raw(200, 150, seed=11)
    bi = hl.Buffer(raw)
    m3, m7 = M3200.copy(), M7000.copy()
    b3, b7 = hl.Buffer(m3), hl.Buffer(m7)

    def run(p):
        bo = hl.Buffer(np.zeros((3, 120, 160), np.uint8))
        hl.camera_pipe(bi, b3, b7, p["color_temp"], p["gamma"], p["contrast"], p["sharpen"], p["black"], p["white"], bo)
        return bo.numpy().copy()

    def want(p, a3, a7):
        return oracle.camera_pipe(raw, a3, a7, p["color_temp"], p["gamma"], p["contrast"], p["sharpen"], p["black"], p["white"], 160, 120)
    first, again = run(PARAMS), run(PARAMS)            # second call: cached block
    assert np.array_equal(first, again) and np.array_equal(first, want(PARAMS, m3, m7))
    p2 = dict(PARAMS, gamma=1.4, sharpen=3.0)
    assert np.array_equal(run(p2), want(p2, m3, m7))   # other scalars
    assert np.array_equal(run(PARAMS), first)          # and back: the first block is still cached
    m3[...] = m3 * np.float32(0.75)                     # rewrite the SAME host array, tell the library
    b3.set_host_dirty()
    changed = run(PARAMS)
    assert np.array_equal(changed, want(PARAMS, m3, m7)) and not np.array_equal(changed, first)
    b7.device_free()                                    # a new allocation, most likely at the same address
    m7b = (m7 * np.float32(1.25)).astype(np.float32)
    b7 = hl.Buffer(m7b)
    assert np.array_equal(run(PARAMS), want(PARAMS, m3, m7b))


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
