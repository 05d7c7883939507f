#!/usr/bin/env python3
"""Parity at sizes where element offsets leave 31 bits: GPU vs oracle, bit for bit (the suite's largest case is 7680x4320).

    python scripts/big_parity.py [--only a,b]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import halide_amd as hl  # noqa: E402
import oracle_lib as oracle  # noqa: E402

oracle.set_canon(hl.canon_fma())   # the oracle evaluates the canonical form the loaded library was built for

f32 = np.float32
rng = np.random.default_rng(3)


def u16(shape):
    return rng.integers(0, 65536, shape, dtype=np.uint16)


def smooth(c, h, w):
    yy = np.arange(h, dtype=f32)[:, None]
    xx = np.arange(w, dtype=f32)[None, :]
    return np.stack([(np.sin(xx / (97.0 + 7 * i)) + np.cos(yy / (131.0 - 5 * i))) * 0.24 + 0.5 for i in range(c)]).astype(f32)


def run(name, fn):
    t = time.time()
    try:
        ok, note = fn()
        print(f"{name}: {'OK' if ok else 'MISMATCH'} {note} ({time.time() - t:.1f} s)", flush=True)
        return ok
    except Exception as e:  # noqa: BLE001
        print(f"{name}: EXCEPTION {type(e).__name__}: {str(e)[:300]} ({time.time() - t:.1f} s)", flush=True)
        return False


def c_local_laplacian():
    w = h = 16384                                   # 268 Mpx: the 9 level-1 planes alone are 2.4 G floats
    inp = (smooth(3, h, w) * 65535).astype(np.uint16)
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
    hl.local_laplacian(a, 8, f32(1 / 7), 1.0, o)
    got = o.numpy()
    want = oracle.local_laplacian(inp, 8, f32(1 / 7), 1.0)
    return np.array_equal(got, want), f"{w}x{h}x3 u16"


def c_blur():
    w, h = 40000, 30000                             # 1.2 G elements
    inp = u16((h + 2, w + 2))
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros((h, w), np.uint16))
    hl.halide_blur(a, o)
    return np.array_equal(o.numpy(), oracle.blur(inp)), f"{w}x{h} u16"


def c_stencil_chain():
    w, h = 24000, 20000
    inp = u16((h, w))
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
    hl.stencil_chain(a, o)
    return np.array_equal(o.numpy(), oracle.stencil_chain(inp)), f"{w}x{h} u16"


def c_bilateral_grid():
    w, h = 24000, 20000
    inp = smooth(1, h, w)[0]
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
    hl.bilateral_grid(a, f32(0.1), o)
    return np.array_equal(o.numpy().view(np.uint32), oracle.bilateral_grid(inp, f32(0.1)).view(np.uint32)), f"{w}x{h} f32"


def c_unsharp():
    w, h = 16000, 15000                             # 3 x 240 M floats: plane offsets beyond 2^31 bytes
    inp = smooth(3, h + 6, w + 6)
    a, o = hl.Buffer(inp).set_min(-3, -3, 0), hl.Buffer(np.zeros((3, h, w), f32))
    hl.unsharp(a, o)
    return np.array_equal(o.numpy().view(np.uint32), oracle.unsharp(inp, (0, 0), (w, h), (-3, -3)).view(np.uint32)), f"{w}x{h}x3 f32"


def c_iir_blur():
    w, h = 16384, 12288
    inp = smooth(3, h, w)
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
    hl.iir_blur(a, f32(0.1), o)
    return np.array_equal(o.numpy().view(np.uint32), oracle.iir_blur(inp, f32(0.1)).view(np.uint32)), f"{w}x{h}x3 f32"


def c_hist():
    w, h = 27000, 26000                             # 702 Mpx x 3 u8: 2.1 G elements, just under the 2^31 - 1 the entry checks allow
    inp = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
    hl.hist(a, o)
    return np.array_equal(o.numpy(), oracle.hist(inp)), f"{w}x{h}x3 u8"


def c_interpolate():
    w, h = 14000, 12000                             # 4 x 168 M floats
    inp = smooth(4, h, w)
    inp[3] *= (rng.random((h, w), dtype=f32) > 0.6)
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros((3, h, w), f32))
    hl.interpolate(a, o)
    return np.array_equal(o.numpy().view(np.uint32), oracle.interpolate(inp).view(np.uint32)), f"{w}x{h}x4 f32"


def c_bgu():
    W, H = 16384, 12288
    hi = smooth(3, H, W)
    lo = hi[:, ::8, ::8].copy()
    val = (lo * lo * (3 - 2 * lo)).astype(f32)
    o = hl.Buffer(np.zeros((3, H, W), f32))
    hl.bgu(1 / 8, 16, hl.Buffer(lo), hl.Buffer(val), hl.Buffer(hi), o)
    return np.array_equal(o.numpy().view(np.uint32), oracle.bgu(1 / 8, 16, lo, val, hi).view(np.uint32)), f"{W}x{H}x3 f32"


def c_camera_pipe():
    ow, oh = 27000, 26000
    raw = rng.integers(0, 1024, (oh + 32, ow + 40), dtype=np.uint16)
    m3 = np.array([[1.6697, -0.2693, -0.4004, -42.4346], [-0.3576, 1.0615, 1.5949, -37.1158], [-0.2175, -1.8751, 6.9640, -26.6970]], f32)
    m7 = np.array([[2.2997, -0.4478, 0.1706, -39.0923], [-0.3826, 1.5906, -0.2080, -25.4311], [-0.0888, -0.7344, 2.2832, -20.0826]], f32)
    o = hl.Buffer(np.zeros((3, oh, ow), np.uint8))
    hl.camera_pipe(hl.Buffer(raw), hl.Buffer(m3), hl.Buffer(m7), 3700.0, 2.0, 50.0, 1.0, 25, 1023, o)
    return np.array_equal(o.numpy(), oracle.camera_pipe(raw, m3, m7, 3700.0, 2.0, 50.0, 1.0, 25, 1023, ow, oh)), f"{ow}x{oh}x3 u8"


CASES = {k[2:]: v for k, v in list(globals().items()) if k.startswith("c_")}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    only = [s for s in args.only.split(",") if s]
    bad = [n for n, fn in CASES.items() if (not only or n in only) and not run(n, fn)]
    print("BIG " + ("CLEAN" if not bad else "FOUND " + ",".join(bad)))
    sys.exit(1 if bad else 0)
