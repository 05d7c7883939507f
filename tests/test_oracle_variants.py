"""The unpinned part of parity, quantified (VERDICT r1 "Missing #3").

The oracle's canonical form of local_laplacian = the generator's expressions after the reference's simplifier
(src/Simplify_Div.cpp:204 x/c -> x*fold(1/c); src/Simplify_Mul.cpp:70 (x*c0)*c1 -> x*fold(c0*c1); constants fold in
double and round to float32, src/IRMatch.h:1014-1016), no contraction.  What LLVM's fast-math contraction
(src/CodeGen_LLVM.cpp:483-500) adds on top is not reproducible here; oracle/local_laplacian_oracle.c carries the
plausible alternatives as variants and these tests keep the spread between them small and known.  The 4K table lives
in profiles/r02_oracle_variants.md (scripts/oracle_variants.py)."""
import numpy as np
import pytest

f32 = np.float32


def test_canonical_gray_constants_are_the_simplifier_folds(oracle):
    r = f32(1.0 / 65535.0)                                     # fold(1 / 65535.0f): double division, rounded once
    assert r == f32(1.0) / f32(65535.0)                        # ... which the float division agrees with here
    want = [f32(float(r) * float(f32(c))) for c in (0.299, 0.587, 0.114)]
    got = oracle.ll_gray_constants()
    assert [float(g).hex() for g in got] == [float(w).hex() for w in want]
    # and folding is NOT a no-op: for most u16 inputs u * C differs from (u * r) * coef in the last place
    u = np.arange(65536, dtype=np.float32)
    assert np.count_nonzero(u * want[0] != (u * r) * f32(0.299)) > 10000


def _frames():
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:270, 0:480].astype(np.float32)
    base = (np.sin(xx / 41.0) + np.cos(yy / 23.0) + 2.2) / 4.4
    smooth = np.clip(np.stack([base * 65535, base * 52000, base[::-1] * 46000]) + rng.normal(0, 900, (3, 270, 480)), 0, 65535)
    return {"smooth": smooth.astype(np.uint16), "noise": rng.integers(0, 65536, (3, 270, 480), dtype=np.uint16)}


@pytest.mark.parametrize("kind", ["smooth", "noise"])
def test_spread_between_plausible_canonicalisations_is_small(oracle, kind):
    """Every variant stays within a fraction of a percent of differing u16 outputs, nearly all by one LSB; the rare
    large differences come from a truncation (`int(gray*(K-1)*256)`, `int(level)`) landing on the other side of an
    integer, which moves the result to a neighbouring LUT entry / pyramid plane."""
    frame = _frames()[kind]
    base = oracle.local_laplacian(frame, 8, 1.0 / 7.0, 1.0)
    seen_any_difference = False
    for v in (oracle.LL_VAR_SOURCE, oracle.LL_VAR_FMA, oracle.LL_VAR_SOURCE | oracle.LL_VAR_FMA, oracle.LL_VAR_DIV,
              oracle.LL_VAR_DIV | oracle.LL_VAR_FMA):
        other = oracle.local_laplacian(frame, 8, 1.0 / 7.0, 1.0, variant=v)
        d = np.abs(base.astype(np.int32) - other.astype(np.int32))
        n = np.count_nonzero(d)
        seen_any_difference |= n > 0
        assert n / d.size < 0.01, (v, n)
        if n:
            assert np.percentile(d[d > 0], 90) <= 4, (v, np.percentile(d[d > 0], 90))
        assert d.max() < 1024, (v, int(d.max()))
    assert seen_any_difference            # the variants are real alternatives, not aliases of the canonical form
    # the variant switch is per call: the canonical result is reproduced afterwards
    assert np.array_equal(base, oracle.local_laplacian(frame, 8, 1.0 / 7.0, 1.0))
