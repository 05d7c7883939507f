#!/usr/bin/env python3
"""Find lens_blur's random_float() tag from an output of the reference (scripts/lens_blur_tag.md).

    python scripts/lens_blur_find_tag.py --stmt-constant K [--call-id 0]
    python scripts/lens_blur_find_tag.py --images left.png right.png reference_output.png [--slices 32 --focus 13 --scale 0.5 --samples 32]

--stmt-constant: K = the 32-bit literal the lowered IR adds to uint32(z) in the hash chain of sample_u (call id 0):
K = P(P(call_id) + tag) with P the permutation polynomial of src/Random.cpp:14-16, :61; the tag is found by search.
--images: evaluates the CPU oracle (oracle/lens_blur_oracle.c through tests/oracle_lib.py) for tags 0..255 on the given stereo pair and
reports the tag whose result equals the reference's saved output (compared as the 8-bit PNG process.cpp writes).  Test /
integration tooling: uses the oracle, not the product."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stmt-constant", type=lambda v: int(v, 0) & 0xFFFFFFFF, default=None)
    ap.add_argument("--call-id", type=int, default=0)
    ap.add_argument("--images", nargs=3, metavar=("LEFT", "RIGHT", "REFERENCE_OUTPUT"))
    ap.add_argument("--slices", type=int, default=32)
    ap.add_argument("--focus", type=int, default=13)
    ap.add_argument("--scale", type=float, default=0.5)
    ap.add_argument("--samples", type=int, default=32)
    ap.add_argument("--max-tag", type=int, default=255)
    a = ap.parse_args()
    if a.stmt_constant is not None:
        M = 0xFFFFFFFF

        def P(v):   # rng32 of src/Random.cpp:20-62 on uint32
            return (((1040796640 * v) & M) + 1121052041 & M) * v + 576942909 & M
        base = P(a.call_id & M)
        hits = [t for t in range(1 << 16) if P((base + t) & M) == a.stmt_constant]
        print(f"tag {hits[0]}" if hits else "no tag in 0..65535 gives that constant (wrong literal, or another call id?)")
        return 0 if hits else 1
    if not a.images:
        ap.error("give --stmt-constant K or --images LEFT RIGHT REFERENCE_OUTPUT")
    a.left, a.right, a.reference_output = a.images
    import oracle_lib
    from test_dropin_drivers import read_png
    left, right, ref = read_png(a.left)[:3], read_png(a.right)[:3], read_png(a.reference_output)[:3]
    if left.dtype != np.uint8 or right.dtype != np.uint8:
        raise SystemExit("the stereo pair must be 8-bit PNG files (apps/images/rgb.png style)")
    best = None
    for tag in range(a.max_tag + 1):
        out = oracle_lib.lens_blur(left, right, a.slices, a.focus, a.scale, a.samples, tag=tag)
        # process.cpp saves the float result through convert_and_save_image: [0, 1] -> u8 with rounding
        q = np.clip(np.floor(out * 255.0 + 0.5), 0, 255).astype(np.uint8) if ref.dtype == np.uint8 else \
            np.clip(np.floor(out * 65535.0 + 0.5), 0, 65535).astype(np.uint16)
        diff = int(np.count_nonzero(q != ref))
        if best is None or diff < best[1]:
            best = (tag, diff)
        if diff == 0:
            print(f"tag {tag}: output identical to the reference's")
            return 0
    print(f"no tag in 0..{a.max_tag} reproduces the reference output; closest: tag {best[0]} with {best[1]} differing samples")
    return 1


if __name__ == "__main__":
    sys.exit(main())
