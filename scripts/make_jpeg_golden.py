#!/usr/bin/env python3
"""tests/golden/jpeg/: small baseline JPEG files and the samples libjpeg-turbo (through Pillow) decodes them to, and source images
with the files libjpeg-turbo writes for them — the pin of halide_amd/tools/hlmi_jpeg.h (decoder and encoder).  Needs Pillow (present in the build container, not on the GPU box): the files and the expected
arrays are committed; this script is how they were made.

    python scripts/make_jpeg_golden.py"""
import os

import numpy as np
from PIL import Image

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "jpeg")


def scene(h, w, c, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    planes = [128 + 90 * np.sin(xx / (7.0 + 3 * i)) * np.cos(yy / (5.0 + 2 * i)) + 30 * ((xx + yy) % 17 > 8) for i in range(c)]
    return (np.stack(planes, -1) + rng.normal(0, 12, (h, w, c))).clip(0, 255).astype(np.uint8)


CASES = [  # name, (h, w), mode, subsampling, quality, extra save options
    ("rgb444_q95", (24, 40), "RGB", 0, 95, {}),
    ("rgb422_q75", (33, 50), "RGB", 1, 75, {}),
    ("rgb420_q75", (37, 61), "RGB", 2, 75, {}),
    ("rgb420_q30_opt", (48, 64), "RGB", 2, 30, {"optimize": True}),
    ("rgb420_q90_rst", (40, 56), "RGB", 2, 90, {"restart_marker_blocks": 2}),
    ("rgb420_narrow", (9, 4), "RGB", 2, 95, {}),     # chroma two samples wide: libjpeg replicates instead of filtering
    ("gray_q85", (29, 43), "L", 0, 85, {}),
    ("gray_q100_rst", (16, 16), "L", 0, 100, {"restart_marker_blocks": 1}),
]

ENC_CASES = [  # name, (h, w), mode, quality: what libjpeg writes with its defaults (4:2:0 for RGB, the Annex K tables)
    ("enc_rgb_q99", (37, 61), "RGB", 99),          # the reference's save_jpg setting; 61 columns: an MCU padded with a dummy block
    ("enc_rgb_q75_odd", (17, 23), "RGB", 75),
    ("enc_gray_q99", (29, 43), "L", 99),
    ("enc_rgb_q30_tiny", (3, 5), "RGB", 30),
]

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    expected = {}
    for seed, (name, (h, w), mode, sub, q, extra) in enumerate(CASES):
        img = scene(h, w, 3 if mode == "RGB" else 1, seed)
        path = os.path.join(OUT, name + ".jpg")
        Image.fromarray(img if mode == "RGB" else img[..., 0], mode).save(path, "JPEG", quality=q, subsampling=sub, **extra)
        dec = np.asarray(Image.open(path))
        expected[name] = dec if dec.ndim == 3 else dec[..., None]
        print(name, dec.shape, os.path.getsize(path), "bytes")
    for seed, (name, (h, w), mode, q) in enumerate(ENC_CASES):
        img = scene(h, w, 3 if mode == "RGB" else 1, 100 + seed)
        Image.fromarray(img if mode == "RGB" else img[..., 0], mode).save(os.path.join(OUT, name + ".jpg"), "JPEG", quality=q)
        expected["src_" + name] = img
        print(name, img.shape, os.path.getsize(os.path.join(OUT, name + ".jpg")), "bytes")
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **expected)
