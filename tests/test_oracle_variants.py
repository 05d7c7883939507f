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
    with oracle.canon(0):
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
    with oracle.canon(0):
        assert np.array_equal(base, oracle.local_laplacian(frame, 8, 1.0 / 7.0, 1.0))
    # and the fma variant IS canon 1
    with oracle.canon(1):
        assert np.array_equal(oracle.local_laplacian(frame, 8, 1.0 / 7.0, 1.0),
                              oracle.local_laplacian(frame, 8, 1.0 / 7.0, 1.0, variant=oracle.LL_VAR_FMA))


def test_distance_between_the_two_canonical_forms_per_pipeline(oracle, capsys):
    """Round 6: every float pipeline has a contracted canonical form (canon 1) beside the one-rounding-per-operator form
    (canon 0).  Whichever the reference's object is, the gap is bounded by their distance, printed here per pipeline (the
    larger table: profiles/r06_oracle_canon_distance.md, scripts/oracle_variants.py --canon) together with the re-association
    study for the two pipelines with re-associable sums."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("oracle_variants", os.path.join(root, "scripts", "oracle_variants.py"))
    ov = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ov)
    rows = ov.canon_table(scale=1)
    with capsys.disabled():
        print()
        for name, what, n, tot, mx, p99, unit, mabs in rows:
            print(f"  {name:32s} {what:32s} {n:7d} of {tot:7d} differ ({n / tot:7.3%}), max {mx} {unit} (p99 {p99:.0f}), max |diff| {mabs:.3g}")
    by = {(r[0], r[1]): r for r in rows}
    moved = 0
    for (name, what), r in by.items():
        n, tot, mx, unit, mabs = r[2], r[3], r[4], r[6], r[7]
        moved += n > 0
        if unit == "LSB":
            assert n / tot < 0.02, (name, what, n)          # integer outputs: a fraction of a percent moves
        else:
            assert mabs < 2e-3, (name, what, mabs)          # float outputs: far below anything visible (values are O(1))
    assert moved >= 8                                       # the two forms are real alternatives for most pipelines
    assert oracle.get_canon() in (0, 1)
