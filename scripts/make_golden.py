#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_digests.json: sha256 of the CPU oracle's output on fixed seeded inputs, one entry per
pipeline.  These are REGRESSION pins of the restatement (the reference ships no golden outputs for its apps and cannot be
built here), so that an accidental change of an oracle — the thing every GPU parity test is measured against — fails a
CPU test.  tests/test_oracle_digests.py recomputes and compares them."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cases():
    import oracle_lib as o
    r = lambda seed: np.random.default_rng(seed)
    f32 = lambda seed, shape: r(seed).random(shape, dtype=np.float32)
    u16 = lambda seed, shape: r(seed).integers(0, 65536, shape, dtype=np.uint16)
    m3 = np.array([[1.6697, -0.2693, -0.4004, -42.4346], [-0.3576, 1.0615, 1.5949, -37.1158], [-0.2175, -1.8751, 6.9640, -26.6970]], np.float32)
    m7 = np.array([[2.2997, -0.4478, 0.1706, -39.0923], [-0.3826, 1.5906, -0.2080, -25.4311], [-0.0888, -0.7344, 2.2832, -20.0826]], np.float32)
    rgba = f32(9, (4, 21, 34))
    rgba[3][r(10).random((21, 34)) < 0.4] = 0
    return {
        "local_laplacian_64x48_seed11_K8": lambda: o.local_laplacian(u16(11, (3, 48, 64)), 8, 1.0 / 7.0, 1.0),
        "blur_70x50_seed1": lambda: o.blur(u16(1, (52, 72))),
        "stencil_chain_80x60_seed2": lambda: o.stencil_chain(u16(2, (60, 80))),
        "bilateral_grid_96x72_seed3_r0.1": lambda: o.bilateral_grid(f32(3, (72, 96)), 0.1),
        "nl_means_40x30_seed4": lambda: o.nl_means(f32(4, (3, 30, 40)), 7, 7, 0.12),
        "conv_layer_2x7x9_32to128_seed5": lambda: o.conv_layer(r(5).uniform(-1, 1, (2, 9, 11, 32)).astype(np.float32),
                                                               r(6).uniform(-1, 1, (32, 3, 3, 128)).astype(np.float32),
                                                               r(7).uniform(-1, 1, 128).astype(np.float32)),
        "conv_layer_bf16_1x5x6_64to128_seed5": lambda: o.conv_layer_bf16(r(5).uniform(-1, 1, (1, 7, 8, 64)).astype(np.float32),
                                                                        r(6).uniform(-1, 1, (64, 3, 3, 128)).astype(np.float32),
                                                                        r(7).uniform(-1, 1, 128).astype(np.float32))[0],
        "camera_pipe_160x120_seed8": lambda: o.camera_pipe(r(8).integers(0, 1024, (152, 200), dtype=np.uint16), m3, m7, 3700.0, 2.0,
                                                           50.0, 1.0, 25, 1023, 160, 120),
        "depthwise_separable_conv_2x6x7_8to5_seed9": lambda: o.depthwise_separable_conv(
            r(9).uniform(-1, 1, (2, 6, 7, 8)).astype(np.float32), r(10).uniform(-1, 1, (3, 3, 8, 1)).astype(np.float32),
            r(11).uniform(-1, 1, (8, 5)).astype(np.float32), r(12).uniform(-1, 1, 5).astype(np.float32)),
        "unsharp_50x40_seed13": lambda: o.unsharp(f32(13, (3, 40, 50)) * 0.9 + 0.05),
        "max_filter_70x64_seed17": lambda: o.max_filter(f32(17, (3, 64, 70))),
        "hist_90x60_seed14": lambda: o.hist(r(14).integers(0, 256, (3, 60, 90), dtype=np.uint8)),
        "harris_50x40_seed15": lambda: o.harris(f32(15, (3, 40, 50))),
        "interpolate_34x21_seed9": lambda: o.interpolate(rgba),
        "iir_blur_30x20_seed16_a0.3": lambda: o.iir_blur(f32(16, (3, 20, 30)), 0.3),
    }


def digests():
    return {k: hashlib.sha256(np.ascontiguousarray(f()).tobytes()).hexdigest() for k, f in cases().items()}


def decode_png_rgb8(path):
    """8-bit RGB, non-interlaced PNG -> (3, H, W) uint8 (zlib + the five PNG row filters; nothing else is needed here)."""
    import struct
    import zlib
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            assert (depth, ctype, interlace) == (8, 2, 0)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    out = np.zeros((h, 3 * w), np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        up = out[y - 1] if y else np.zeros(3 * w, np.int32)
        cur = out[y]
        for x in range(3 * w):
            a = cur[x - 3] if x >= 3 else 0
            b, c = up[x], (up[x - 3] if x >= 3 else 0)
            if ft == 0:
                pr = 0
            elif ft == 1:
                pr = a
            elif ft == 2:
                pr = b
            elif ft == 3:
                pr = (a + b) >> 1
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            cur[x] = (line[x] + pr) & 255
    return np.ascontiguousarray(out.astype(np.uint8).reshape(h, w, 3).transpose(2, 0, 1))


def natural_image():
    """tests/golden/rgb_small_u8.npz: the pixels of the reference checkout's one natural RGB test image
    (/root/reference/apps/images/rgb_small.png, 192 x 320, 8 bit) as a planar array — INPUT DATA for bench.py's `natural` variant
    (SURVEY.md §8d (iii)), not source code; the GPU box has no /root/reference."""
    src = "/root/reference/apps/images/rgb_small.png"
    if not os.path.exists(src):
        return False
    rgb = decode_png_rgb8(src)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rgb_small_u8.npz"), rgb=rgb)
    return True


if __name__ == "__main__":
    if "--natural" in sys.argv:
        print("natural image fixture:", natural_image())
        sys.exit(0)
    d = {"_comment": "Regression digests of the canonical CPU oracle (sha256 of the output bytes on fixed seeded inputs, "
                     "scripts/make_golden.py). NOT reference-derived: the reference ships no golden outputs for these pipelines."}
    import oracle_lib
    with oracle_lib.canon(0):
        d.update(digests())
    with oracle_lib.canon(1):   # the contracted canonical form (oracle/oracle_common.h)
        d["_canon1"] = digests()
    with open(os.path.join(ROOT, "tests", "golden", "oracle_digests.json"), "w") as f:
        json.dump(d, f, indent=1)
    print(json.dumps(d, indent=1))
