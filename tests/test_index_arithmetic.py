"""The divisions-by-multiplication of round 5's 32-bit index arithmetic, checked exhaustively over the ranges the kernels use them on
(CPU; the kernels carry the same bounds as static_asserts or host-side guards):
  e / d == (e * ceil(2^k / d)) >> k   as long as   e * (ceil(2^k / d) * d - 2^k) < 2^k."""
import numpy as np
import pytest


def magic(d, k):
    return ((1 << k) + d - 1) // d


def exact_upto(d, k, n):
    e = np.arange(n, dtype=np.uint64)
    return bool(np.array_equal((e * np.uint64(magic(d, k))) >> np.uint64(k), e // np.uint64(d)))


@pytest.mark.parametrize("d,k,n,where", [
    (130, 22, 130 * 302, "local_laplacian.hip ll_up0h: level-1 tile value e -> row (U0_TW = 130, at most RU + 2 = 302 rows fit LDS)"),
    (12, 16, 1680 + 1, "bilateral_grid.hip divc<12, NBZ>: blurz element -> cell"),
    (14, 16, 140 + 1, "bilateral_grid.hip divc<14, NBZ / 12>: cell -> grid row"),
    (12, 16, 1200 + 256 + 1, "bilateral_grid.hip divc<12, NBX + 256>"),
    (10, 16, (1200 + 256) // 12 + 2, "bilateral_grid.hip divc<10, ..>"),
    (35, 16, 11 * 35 + 1, "lens_blur.hip lb_divc<PM2W, PM2H * PM2W>"),
    (21, 16, 9 * 21 + 1, "lens_blur.hip lb_divc<PM3W, PM3H * PM3W>"),
    (3, 16, 128, "depthwise_separable_conv.hip: filter element -> (ry, rx), q < 128"),
    (112, 21, 35 * 256, "bilateral_grid.hip bg_blur_slice<HIST>: staged pixel number -> row of the tile's 112 x 80 input region"),
])
def test_division_by_a_constant_as_one_multiplication_is_exact_on_its_range(d, k, n, where):
    assert exact_upto(d, k, n), where
    # and the product stays inside 32 bits (v_mul_u32_u24 returns the low 32 bits; both factors below 2^24)
    assert (n - 1) * magic(d, k) < 1 << 32 and magic(d, k) < 1 << 24 and n <= 1 << 24, where


def test_ll_up0h_level2_tile_rows_for_every_run_time_width():
    """ll_up0h, FUSE2 phase: e / n2x with m2 = ceil(2^22 / n2x) for the run-time tile width n2x <= 68 (U0H_T2) and e < n2x * n2y,
    n2y <= RU / 2 + 3 <= 153 (the largest tile LDS can hold)."""
    for n2x in range(1, 69):
        n = n2x * 153
        assert exact_upto(n2x, 22, n), n2x
        assert (n - 1) * magic(n2x, 22) < 1 << 32 and magic(n2x, 22) < 1 << 24


def test_the_unsharp_edge_columns_index():
    """unsharp_tile2: ei / 6 as (ei * 10923) >> 16 for ei < 228 (six edge columns x 38 rows)."""
    e = np.arange(228)
    assert np.array_equal((e * 10923) >> 16, e // 6)


def test_upsample_taps_are_adjacent_pairs():
    """fdiv2(v + 1) == fdiv2(v - 1) + 1 for every integer v: the four taps of an upsampled value are two adjacent pairs (ll_up0h's
    8-byte tap loads), and lens_blur's xa = (x >> 1) - 1 + 2 (x & 1) is xb - 1 or xb + 1 by the parity of x."""
    v = np.arange(-1000, 1000)
    assert np.array_equal((v + 1) // 2, (v - 1) // 2 + 1)
    x = np.arange(0, 1000)
    xa, xb = (x >> 1) - 1 + 2 * (x & 1), x >> 1
    assert np.array_equal(xa - xb, np.where(x & 1, 1, -1))
