"""Canonicalisation study of the local_laplacian oracle (SURVEY.md §7 "hard parts", §8c).

The reference's float results are the generator's expressions after Halide's simplifier (deterministic: restated as the
oracle's CANONICAL form) and after LLVM's fast-math contraction (not deterministic across targets / versions).  This
script measures how far the u16 OUTPUT moves between the canonical form and the other plausible evaluations, on the
bench frame (smooth) and on uniform full-range noise — i.e. it bounds the unpinned oracle <-> Halide gap by the spread of
everything Halide could plausibly have emitted.  CPU only (the oracle); writes a markdown table.

    python scripts/oracle_variants.py [--size 3840x2160] [--out profiles/r02_oracle_variants.md]

Round 6: the oracle has TWO canonical forms (oracle/oracle_common.h: canon 0 = one rounding per operator, canon 1 = mul+add
pairs contracted as LLVM contracts them); `--canon` prints, for EVERY float pipeline, how far the output moves between the
two, and for local_laplacian and nl_means how far re-association (balanced trees instead of left-to-right sums, LLVM's
`reassoc` flag) moves it from each:

    python scripts/oracle_variants.py --canon [--out profiles/r06_oracle_canon_distance.md]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as o  # noqa: E402  (the checker; this script is test infrastructure)

VARIANTS = [
    ("source order: (u*r)*coef, no fma  [round 1's canonical form]", o.LL_VAR_SOURCE),
    ("canonical + fma contraction  [simplifier + LLVM contract: most likely x86-64 AVX2/FMA object]", o.LL_VAR_FMA),
    ("source order + fma", o.LL_VAR_SOURCE | o.LL_VAR_FMA),
    ("true division u/65535.0f, no fma", o.LL_VAR_DIV),
    ("true division + fma", o.LL_VAR_DIV | o.LL_VAR_FMA),
]


def smooth_frame(w, h, seed=1):
    """bench.py's frame (low-frequency gradients + mild noise), at any size."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = (np.sin(xx / 311.0 + seed) + np.cos(yy / 173.0) + np.sin((xx + yy) / 97.0) + 3.3) / 6.6
    img = np.stack([base * 65535.0, np.roll(base, 64, 1) * 52000.0, base[::-1] * 46000.0])
    img += rng.normal(0.0, 900.0, img.shape).astype(np.float32)
    return np.clip(img, 0, 65535).astype(np.uint16)


def noise_frame(w, h, seed=0):
    return np.random.default_rng(seed).integers(0, 65536, (3, h, w), dtype=np.uint16)


def compare(base, other):
    d = np.abs(base.astype(np.int32) - other.astype(np.int32))
    n = int(np.count_nonzero(d))
    return n, n / d.size, int(d.max()), float(np.percentile(d[d > 0], 99)) if n else 0.0


def _ulp(a, b):
    """distance in units in the last place of two float32 arrays (sign-magnitude order of the bit patterns)"""
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia, ib = np.where(ia < 0, -(ia & 0x7fffffff), ia), np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def distance(a, b):
    """(number that differ, total, max distance, p99 of the nonzero distances, unit) between two outputs of one pipeline"""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.dtype.kind == "f":
        d, unit = _ulp(a.astype(np.float32), b.astype(np.float32)), "ulp"
    else:
        d, unit = np.abs(a.astype(np.int64) - b.astype(np.int64)), "LSB"
    n = int(np.count_nonzero(d))
    mabs = float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if d.size else 0.0
    return n, d.size, int(d.max()) if d.size else 0, float(np.percentile(d[d > 0], 99)) if n else 0.0, unit, mabs


def canon_cases(scale=1):
    """one seeded case per float pipeline, (name, callable(oracle) -> output); scale 1 = the sizes the CPU test uses"""
    r = lambda seed: np.random.default_rng(seed)
    f32 = lambda seed, shape: r(seed).random(shape, dtype=np.float32)
    k = scale
    rgba = f32(9, (4, 60 * k, 80 * k))
    rgba[3][r(10).random((60 * k, 80 * k)) < 0.4] = 0
    m3 = np.array([[1.6697, -0.2693, -0.4004, -42.4346], [-0.3576, 1.0615, 1.5949, -37.1158], [-0.2175, -1.8751, 6.9640, -26.6970]], np.float32)
    m7 = np.array([[2.2997, -0.4478, 0.1706, -39.0923], [-0.3826, 1.5906, -0.2080, -25.4311], [-0.0888, -0.7344, 2.2832, -20.0826]], np.float32)
    left = r(21).integers(0, 256, (3, 48 * k, 64 * k), dtype=np.uint8)
    right = np.roll(left, 6, axis=2)
    sl = f32(31, (3, 64 * k, 96 * k))
    lo = np.ascontiguousarray(sl[:, ::8, ::8])
    return [
        ("local_laplacian (u16, smooth)", lambda o: o.local_laplacian(smooth_frame(240 * k, 135 * k), 8, 1.0 / 7.0, 1.0)),
        ("local_laplacian (u16, noise)", lambda o: o.local_laplacian(noise_frame(240 * k, 135 * k), 8, 1.0 / 7.0, 1.0)),
        ("bilateral_grid (f32)", lambda o: o.bilateral_grid(f32(3, (108 * k, 192 * k)), 0.1)),
        # a smooth scene + mild noise: on white noise every weight but the centre's underflows and the output IS the input
        ("nl_means (f32)", lambda o: o.nl_means(np.clip(smooth_frame(64 * k, 40 * k).astype(np.float32) / 65535.0
                                                        + r(4).normal(0, 0.02, (3, 40 * k, 64 * k)).astype(np.float32), 0, 1), 7, 7, 0.12)),
        ("camera_pipe (u8)", lambda o: o.camera_pipe(r(8).integers(0, 1024, (152, 200), dtype=np.uint16), m3, m7, 3700.0, 2.0, 50.0, 1.0,
                                                      25, 1023, 160, 120)),
        ("unsharp (f32)", lambda o: o.unsharp(f32(13, (3, 80 * k, 100 * k)) * 0.9 + 0.05)),
        ("harris (f32)", lambda o: o.harris(f32(15, (3, 80 * k, 100 * k)))),
        ("hist (u8)", lambda o: o.hist(r(14).integers(0, 256, (3, 60 * k, 90 * k), dtype=np.uint8))),
        ("interpolate (f32)", lambda o: o.interpolate(rgba)),
        ("iir_blur (f32)", lambda o: o.iir_blur(f32(16, (3, 64 * k, 96 * k)), 0.3)),
        ("lens_blur (f32)", lambda o: o.lens_blur(left, right)),
        ("bgu (f32)", lambda o: o.bgu(0.125, 8, lo, np.ascontiguousarray(lo[:, ::-1] * 0.8 + 0.1), sl)),
    ]


def canon_table(scale=1):
    """rows of (pipeline, what, n, total, max, p99, unit): canon 1 against canon 0 for every float pipeline; re-association against
    each canonical form for the two pipelines whose sums LLVM could re-associate into trees"""
    rows = []
    for name, fn in canon_cases(scale):
        with o.canon(0):
            c0 = np.asarray(fn(o))
            r0 = None
            if name.startswith(("local_laplacian", "nl_means")):
                with o.reassoc():
                    r0 = np.asarray(fn(o))
        with o.canon(1):
            c1 = np.asarray(fn(o))
            r1 = None
            if r0 is not None:
                with o.reassoc():
                    r1 = np.asarray(fn(o))
        rows.append((name, "canon 1 (fma) vs canon 0") + distance(c0, c1))
        if r0 is not None:
            rows.append((name, "re-associated vs canon 0") + distance(c0, r0))
            rows.append((name, "re-associated + fma vs canon 1") + distance(c1, r1))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--out", default="")
    ap.add_argument("--canon", action="store_true", help="distance between the two canonical forms, every float pipeline")
    ap.add_argument("--scale", type=int, default=4)
    a = ap.parse_args()
    if a.canon:
        lines = ["Distance between the oracle's two canonical forms (oracle/oracle_common.h) and the re-association study, per pipeline: outputs "
                 "that differ, in LSB for integer outputs and in float32 ulp for float outputs.", "",
                 "| pipeline | comparison | outputs that differ | fraction | max | p99 of the nonzero | unit | max abs difference |", "|---|---|---|---|---|---|---|---|"]
        for name, what, n, tot, mx, p99, unit, mabs in canon_table(a.scale):
            lines.append(f"| {name} | {what} | {n} of {tot} | {n / tot:.3%} | {mx} | {p99:.0f} | {unit} | {mabs:.3g} |")
        text = "\n".join(lines) + "\n"
        print(text)
        if a.out:
            with open(a.out, "w") as f:
                f.write(text)
        return
    w, h = (int(v) for v in a.size.split("x"))
    lines = [f"local_laplacian oracle, {w}x{h}x3 u16, levels=8 alpha=1/7 beta=1: u16 outputs that differ from the CANONICAL "
             "form (simplifier-folded constants, no contraction)", "",
             "| input | variant | outputs that differ | fraction | max abs diff (LSB of 65535) | p99 of the nonzero diffs |",
             "|---|---|---|---|---|---|"]
    for iname, frame in (("smooth (bench frame)", smooth_frame(w, h)), ("uniform noise", noise_frame(w, h))):
        t0 = time.time()
        with o.canon(0):
            base = o.local_laplacian(frame, 8, 1.0 / 7.0, 1.0)
        for vname, v in VARIANTS:
            n, frac, mx, p99 = compare(base, o.local_laplacian(frame, 8, 1.0 / 7.0, 1.0, variant=v))
            lines.append(f"| {iname} | {vname} | {n} | {frac:.4%} | {mx} | {p99:.0f} |")
        print(f"{iname}: {time.time() - t0:.1f} s", file=sys.stderr)
    text = "\n".join(lines) + "\n"
    print(text)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
