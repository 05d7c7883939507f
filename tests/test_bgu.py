"""bgu: oracle vs an independent second reading of the generator, GPU vs oracle.

Reference: apps/bgu/bgu_generator.cpp:268-488 (algorithm), :131-238 (solve_symmetric), src/Lerp.cpp:127-128 (float lerp),
src/runtime/ptx_dev.ll:61-66 / x86.ll:100-106 (fast_inverse on the CUDA path / on x86).  The oracle's header states the
canonical form: generator order after the simplifier, fast_inverse = correctly rounded 1/x as on the reference's GPU path,
histogram sums in the CPU schedule's serial order.
"""
import functools

import numpy as np
import pytest

f32 = np.float32


def _scene(W, H, seed, factor=8, channels=3):
    """A full-res image, its box-downsampled copy and a tone-mapped version of the copy (what apps/bgu/filter.cpp builds)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    hi = np.stack([(np.sin(xx / (9.0 + c) + seed) + np.cos(yy / (13.0 - c))) * 0.22 + 0.5 + rng.random((H, W)) * 0.08
                   for c in range(3)]).astype(f32)
    lw, lh = max(1, W // factor), max(1, H // factor)
    ys, xs = (np.arange(lh) * H // lh), (np.arange(lw) * W // lw)
    lo = hi[:, ys][:, :, xs][:channels].copy()
    val = (lo * lo * (3 - 2 * lo)).astype(f32)
    return hi, lo, val


# ---------------------------------------------------------------------------------------------------- second reading
def naive_bgu(r_sigma, s_sigma, splat, values, slice_loc, region):
    """Every Func of the generator as a memoised pure function on Z^n, float32 operator by operator, in SOURCE form
    (gray = 0.25 a + 0.5 b + 0.25 c; taps multiplied by t3 = 1)."""
    r_sigma = f32(r_sigma)
    lc, lh, lw = splat.shape
    vc, vh, vw = values.shape
    _, H, W = slice_loc.shape
    x0, y0, ow, oh = region
    uf = max(int(np.ceil(f32(W) / f32(lw))), int(np.ceil(f32(H) / f32(lh))))
    big = s_sigma * uf
    inv_r = f32(1) / r_sigma
    nb = int(f32(1) / r_sigma)
    t = [f32(1) / f32(64), f32(1) / f32(27), f32(1) / f32(8), f32(1)]
    w7 = [t[0], t[1], t[2], t[3], t[2], t[1], t[0]]

    def S(x, y, c):
        return splat[min(max(c, 0), lc - 1), min(max(y, 0), lh - 1), min(max(x, 0), lw - 1)]

    def V(x, y, c):
        return values[min(max(c, 0), vc - 1), min(max(y, 0), vh - 1), min(max(x, 0), vw - 1)]

    @functools.lru_cache(None)
    def histogram(x, y):
        """-> dict z -> 22 sums, accumulated r.y outer / r.x inner"""
        h = {}
        for ry in range(s_sigma):
            for rx in range(s_sigma):
                sx, sy = x * s_sigma + rx - s_sigma // 2, y * s_sigma + ry - s_sigma // 2
                sr, sg, sb = S(sx, sy, 0), S(sx, sy, 1), S(sx, sy, 2)
                vr, vg, vb = V(sx, sy, 0), V(sx, sy, 1), V(sx, sy, 2)
                pos = f32(f32(f32(f32(0.25) * sr) + f32(f32(0.5) * sg)) + f32(f32(0.25) * sb))
                pos = max(min(pos, f32(1)), f32(0))
                zi = int(np.rint(f32(pos * inv_r)))
                terms = [sr * sr, sr * sg, sr * sb, sr, sg * sg, sg * sb, sg, sb * sb, sb, f32(1),
                         vr * sr, vr * sg, vr * sb, vr, vg * sr, vg * sg, vg * sb, vg, vb * sr, vb * sg, vb * sb, vb]
                acc = h.setdefault(zi, [f32(0)] * 22)
                for c in range(22):
                    acc[c] = f32(acc[c] + f32(terms[c]))
        return h

    def hist(x, y, z, c):
        return histogram(x, y).get(z, [f32(0)] * 22)[c]

    def blur(fn):
        def g(*taps):
            acc = f32(taps[0] * w7[0])
            for i in range(1, 7):
                acc = f32(acc + f32(taps[i] * w7[i]))
            return acc
        return g

    @functools.lru_cache(None)
    def blurz(x, y, z, c):
        return blur(None)(*[hist(x, y, z + d, c) for d in range(-3, 4)])

    @functools.lru_cache(None)
    def blury(x, y, z, c):
        return blur(None)(*[blurz(x, y + d, z, c) for d in range(-3, 4)])

    @functools.lru_cache(None)
    def blurx(x, y, z, c):
        return blur(None)(*[blury(x + d, y, z, c) for d in range(-3, 4)])

    @functools.lru_cache(None)
    def line(x, y, z):
        b = [blurx(x, y, z, c) for c in range(22)]
        lam = f32(1e-1)
        A = [[b[0], b[1], b[2], b[3]], [b[1], b[4], b[5], b[6]], [b[2], b[5], b[7], b[8]], [b[3], b[6], b[8], b[9]]]
        for i in range(4):
            A[i][i] = f32(A[i][i] + lam)
        rhs = [[b[10 + 4 * k + j] for k in range(3)] for j in range(4)]   # rhs[j][k]
        for i in range(3):
            rhs[i][i] = f32(rhs[i][i] + lam)
        # sqrt-free Cholesky, exactly the generator's statements (:162-229)
        M = 4
        for j in range(M):
            A[j][j] = f32(f32(1) / A[j][j])
            for i in range(j + 1, M):
                A[i][j] = f32(A[i][j] * A[j][j])
            for i in range(j + 1, M):
                for k in range(j + 1, M):
                    if k < i:
                        A[i][k] = A[k][i]
                    else:
                        A[i][k] = f32(A[i][k] - f32(A[k][j] * A[j][i]))
        for k in range(3):
            for j in range(M):
                for i in range(j):
                    rhs[j][k] = f32(rhs[j][k] - f32(A[j][i] * rhs[i][k]))
            for j in range(M):
                rhs[j][k] = f32(rhs[j][k] * A[j][j])
            for j in range(M - 1, -1, -1):
                for i in range(j + 1, M):
                    rhs[j][k] = f32(rhs[j][k] - f32(A[i][j] * rhs[i][k]))
        return [rhs[j][k] for k in range(3) for j in range(4)]   # c = 4 k + j

    def lerp(a, b, w):
        return f32(f32(a * f32(f32(1) - w)) + f32(b * w))

    out = np.zeros((3, oh, ow), f32)
    for yo in range(oh):
        y = y0 + yo
        yf = f32(f32(y) / f32(big))
        yi = int(np.floor(yf))
        yf = f32(yf - f32(yi))
        for xo in range(ow):
            x = x0 + xo
            xf = f32(f32(x) / f32(big))
            xi = int(np.floor(xf))
            xf = f32(xf - f32(xi))
            s = [slice_loc[c, y, x] for c in range(3)]
            val = f32(f32(f32(f32(0.25) * s[0]) + f32(f32(0.5) * s[1])) + f32(f32(0.25) * s[2]))
            val = max(min(val, f32(1)), f32(0))
            zv = f32(val * f32(nb))
            zi = int(zv)
            zf = f32(zv - f32(zi))
            m = []
            for c in range(12):
                mz = []
                for dz in range(2):
                    a = lerp(line(xi, yi, zi + dz)[c], line(xi, yi + 1, zi + dz)[c], yf)
                    b = lerp(line(xi + 1, yi, zi + dz)[c], line(xi + 1, yi + 1, zi + dz)[c], yf)
                    mz.append(lerp(a, b, xf))
                m.append(lerp(mz[0], mz[1], zf))
            for c in range(3):
                v = f32(f32(f32(f32(m[4 * c] * s[0]) + f32(m[4 * c + 1] * s[1])) + f32(m[4 * c + 2] * s[2])) + m[4 * c + 3])
                out[c, yo, xo] = max(min(v, f32(1)), f32(0))
    return out


@pytest.mark.parametrize("W,H,factor,s_sigma,r_sigma,region", [
    (24, 16, 2, 4, 0.25, None),
    (20, 14, 1, 3, 0.125, (3, 2, 11, 9)),       # no upsampling, odd cell size, a crop of the output
    (18, 12, 3, 2, 0.3, None),                   # 1 / r_sigma not an integer: nb = 3, samples land in bins 0..3
])
def test_oracle_matches_naive_pure_function_evaluator(oracle, canon0, W, H, factor, s_sigma, r_sigma, region):
    hi, lo, val = _scene(W, H, seed=W + H, factor=factor)
    reg = region or (0, 0, W, H)
    with np.errstate(all="ignore"):
        want = naive_bgu(r_sigma, s_sigma, lo, val, hi, reg)
    got = oracle.bgu(r_sigma, s_sigma, lo, val, hi, region=reg)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got != want)} of {got.size} differ"


def test_oracle_reproduces_the_operator_it_was_shown(oracle):
    """What the app is for: the low-res pair demonstrates a tone curve, the output applies it at full resolution."""
    hi, lo, val = _scene(192, 256, seed=5)
    out = oracle.bgu(1 / 8, 16, lo, lo, hi)
    assert np.abs(out - np.clip(hi, 0, 1)).max() < 1e-4                     # identity pair -> identity transform
    out = oracle.bgu(1 / 8, 16, lo, val, hi)
    assert np.abs(out - np.clip(hi * hi * (3 - 2 * hi), 0, 1)).mean() < 0.02


def test_x86_fast_inverse_variant_spread(oracle):
    """The reference's x86 path solves with the rcpss estimate (src/runtime/x86.ll:100-106); the canonical form (its CUDA
    path, 1/x correctly rounded) differs from it by about 2^-12 relative in the transforms, i.e. a few 1e-3 in the output —
    this is the distance a maintainer should expect between this library and the reference's CPU build."""
    hi, lo, val = _scene(192, 256, seed=7)
    a = oracle.bgu(1 / 8, 16, lo, val, hi)
    b = oracle.bgu(1 / 8, 16, lo, val, hi, variant=oracle.BGU_X86_RCP)
    d = np.abs(a - b).max()
    assert 0 < d < 2e-2, d


# ---------------------------------------------------------------------------------------------------- GPU
def _run(hl, r_sigma, s_sigma, lo, val, hi, region=None):
    bl, bv, bh = hl.Buffer(lo), hl.Buffer(val), hl.Buffer(hi)
    _, H, W = hi.shape
    x0, y0, ow, oh = region or (0, 0, W, H)
    bo = hl.Buffer(np.zeros((3, oh, ow), np.float32)).set_min(x0, y0, 0)
    hl.bgu(r_sigma, s_sigma, bl, bv, bh, bo)
    return bo.numpy()


def _same(got, want):
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got != want)} of {got.size} differ"


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,factor,s_sigma,r_sigma", [
    (1536, 2560, 8, 16, 1 / 8),      # the generator's estimates (:674-687) = apps/bgu/filter.cpp on a 1536x2560 image
    (768, 1280, 8, 16, 1 / 8),       # apps/images/rgb.png's size
    (300, 211, 4, 8, 1 / 8),         # ragged everything
    (200, 120, 1, 16, 1 / 4),        # no upsampling
    (260, 130, 2, 5, 1 / 16),        # 17 bins x 22 terms: two passes of the histogram's (bin, term) threads
    (96, 64, 2, 3, 0.3),             # 1 / r_sigma not an integer
])
def test_hip_matches_oracle(hl, oracle, W, H, factor, s_sigma, r_sigma):
    hi, lo, val = _scene(W, H, seed=W + 3 * H, factor=factor)
    _same(_run(hl, r_sigma, s_sigma, lo, val, hi), oracle.bgu(r_sigma, s_sigma, lo, val, hi))


@pytest.mark.gpu
def test_hip_output_crops_and_the_direct_slice_kernel(hl, oracle):
    hi, lo, val = _scene(400, 300, seed=11, factor=4)
    for reg in [(37, 21, 300, 200), (128, 64, 1, 1), (0, 299, 400, 1), (399, 0, 1, 300)]:
        _same(_run(hl, 1 / 8, 8, lo, val, hi, region=reg), oracle.bgu(1 / 8, 8, lo, val, hi, region=reg))
    # cells of 2 pixels and 66 intensity planes: the slice tables do not fit LDS, every pixel reads `line` itself
    hi, lo, val = _scene(96, 80, seed=12, factor=1)
    _same(_run(hl, 1 / 64, 2, lo, val, hi), oracle.bgu(1 / 64, 2, lo, val, hi))


@pytest.mark.gpu
def test_hip_low_res_pair_is_clamped_in_every_dimension(hl, oracle):
    """repeat_edge clamps channels too (:270-271): a one-channel splat_loc serves all three, and the two low-res images
    need not have the same extents."""
    hi, lo, val = _scene(256, 192, seed=13, factor=8)
    _same(_run(hl, 1 / 8, 16, lo[:1].copy(), val, hi), oracle.bgu(1 / 8, 16, lo[:1].copy(), val, hi))
    _same(_run(hl, 1 / 8, 16, lo, val[:2, :-3, :-5].copy(), hi), oracle.bgu(1 / 8, 16, lo, val[:2, :-3, :-5].copy(), hi))


def test_rejects_scalars_the_generator_gives_no_meaning_to(hl):
    """s_sigma < 1 is an empty reduction domain and a division by zero (:292, :441), r_sigma <= 0 a negative bin count; the
    library's own limit is 4096 intensity bins.  Reported before a device is looked for."""
    hi, lo, val = _scene(64, 48, seed=14, factor=8)
    for r_sigma, s_sigma in [(1 / 8, 0), (0.0, 16), (-1.0, 16), (1e-6, 16)]:
        with pytest.raises(hl.HalideError) as e:
            _run(hl, r_sigma, s_sigma, lo, val, hi)
        assert e.value.code == -27


@pytest.mark.gpu
def test_hip_padded_strides_and_mins_of_every_buffer(hl, oracle):
    """Row / plane padding of all four buffers, a slice_loc larger than (and offset from) the output, and a low-res pair whose
    box does not start at 0: coordinates are absolute (the low-res pair is clamped to ITS box, :270-271), padding bytes
    stay untouched.  The oracle takes boxes at 0, so the shifted low-res pair is given to it edge-extended to the origin —
    the same function of absolute coordinates — with extents chosen so that the upsampling factor (:275-279) is the same."""
    W, H, lw, lh, m = 256, 192, 29, 25, 2            # ceil(256 / 29) == ceil(256 / 31) == 9, ceil(192 / 25) == ceil(192 / 27) == 8
    hi, _, _ = _scene(W, H, seed=21)
    rng = np.random.default_rng(22)
    lo = rng.random((3, lh, lw), dtype=np.float32)
    val = np.clip(lo * 0.8 + 0.1 * rng.random((3, lh, lw), dtype=np.float32), 0, 1).astype(np.float32)
    pad = lambda a, py, px: np.pad(a, ((0, 0), (0, py), (0, px)), constant_values=np.float32(7))   # noqa: E731
    big_lo, big_val, big_hi = pad(lo, 3, 5), pad(val, 1, 9), pad(hi, 2, 6)
    x0, y0, ow, oh = 37, 21, 180, 150
    big_out = np.full((3, oh + 3, ow + 11), np.float32(-3), np.float32)
    bl = hl.Buffer(big_lo[:, :lh, :lw]).set_min(m, m, 0)
    bv = hl.Buffer(big_val[:, :lh, :lw]).set_min(m, m, 0)
    bh = hl.Buffer(big_hi[:, :H, :W])
    bo = hl.Buffer(big_out[:, :oh, :ow]).set_min(x0, y0, 0)
    hl.bgu(1 / 8, 4, bl, bv, bh, bo)
    bo.copy_to_host()
    ext = lambda a: np.pad(a, ((0, 0), (m, 0), (m, 0)), mode="edge")   # noqa: E731
    want = oracle.bgu(1 / 8, 4, ext(lo), ext(val), hi, region=(x0, y0, ow, oh))
    _same(np.ascontiguousarray(big_out[:, :oh, :ow]), want)
    assert np.all(big_out[:, oh:, :] == -3) and np.all(big_out[:, :, ow:] == -3)
