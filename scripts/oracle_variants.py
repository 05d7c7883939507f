"""Canonicalisation study of the local_laplacian oracle (SURVEY.md §7 "hard parts", §8c).

The reference's float results are the generator's expressions after Halide's simplifier (deterministic: restated as the
oracle's CANONICAL form) and after LLVM's fast-math contraction (not deterministic across targets / versions).  This
script measures how far the u16 OUTPUT moves between the canonical form and the other plausible evaluations, on the
bench frame (smooth) and on uniform full-range noise — i.e. it bounds the unpinned oracle <-> Halide gap by the spread of
everything Halide could plausibly have emitted.  CPU only (the oracle); writes a markdown table.

    python scripts/oracle_variants.py [--size 3840x2160] [--out profiles/r02_oracle_variants.md]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    w, h = (int(v) for v in a.size.split("x"))
    lines = [f"local_laplacian oracle, {w}x{h}x3 u16, levels=8 alpha=1/7 beta=1: u16 outputs that differ from the CANONICAL "
             "form (simplifier-folded constants, no contraction)", "",
             "| input | variant | outputs that differ | fraction | max abs diff (LSB of 65535) | p99 of the nonzero diffs |",
             "|---|---|---|---|---|---|"]
    for iname, frame in (("smooth (bench frame)", smooth_frame(w, h)), ("uniform noise", noise_frame(w, h))):
        t0 = time.time()
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
