"""ctypes bindings to the CPU oracle (oracle/lib/liboracle.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
(halide_amd/, libhlmi.so) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

# The GPU box exposes 256 hardware threads; an OpenMP team that wide only adds fork/join + spin overhead for
# the small parity cases, so cap the team (bench.py's cpu_baseline reports the count it actually used).
os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 64)))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "lib", "liboracle.so")


def build_oracle() -> None:
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def load():
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    return C.CDLL(ORACLE_SO)


_lib = load()
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")

for _name in ("oracle_halide_exp", "oracle_halide_log", "oracle_fast_exp"):
    getattr(_lib, _name).argtypes = [C.c_float]
    getattr(_lib, _name).restype = C.c_float
_lib.oracle_halide_pow.argtypes = [C.c_float, C.c_float]
_lib.oracle_halide_pow.restype = C.c_float
_lib.oracle_ll_remap_lut.argtypes = [C.c_int, C.c_float, _f32p]
_lib.oracle_local_laplacian.argtypes = [_u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_float, C.c_float, _u16p, C.c_int, C.c_int, C.c_int, C.c_void_p]
_lib.oracle_local_laplacian.restype = C.c_int
_lib.oracle_local_laplacian_fast.argtypes = [_u16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_float, C.c_float, _u16p, C.c_int, C.c_int]
_lib.oracle_local_laplacian_fast.restype = C.c_int
_lib.oracle_set_threads.argtypes = [C.c_int]
_lib.oracle_set_threads.restype = None


_lib.oracle_set_canon.argtypes = [C.c_int]
_lib.oracle_get_canon.restype = C.c_int


def set_canon(fma: int) -> None:
    """Selects the oracle's canonical form (oracle/oracle_common.h): 0 = one rounding per operator, 1 = mul+add pairs
    contracted into fma as LLVM contracts the reference's float operations.  tests/conftest.py sets it to the form the
    loaded libhlmi.so was built for (hlmi_canon_fma())."""
    _lib.oracle_set_canon(int(fma))


def get_canon() -> int:
    return int(_lib.oracle_get_canon())


_lib.oracle_set_reassoc.argtypes = [C.c_int]


class reassoc:
    """with oracle_lib.reassoc(): ...   — the re-association study's evaluation (balanced trees for local_laplacian's taps and
    nl_means' patch sums); never a canonical form."""

    def __enter__(self):
        _lib.oracle_set_reassoc(1)
        return self

    def __exit__(self, *exc):
        _lib.oracle_set_reassoc(0)
        return False


class canon:
    """with oracle_lib.canon(0): ...   — evaluates the oracle in the given form, then restores the one in force."""

    def __init__(self, fma: int):
        self.fma = int(fma)

    def __enter__(self):
        self.prev = get_canon()
        set_canon(self.fma)
        return self

    def __exit__(self, *exc):
        set_canon(self.prev)
        return False


def halide_exp(x: float) -> float:
    return _lib.oracle_halide_exp(x)


def halide_log(x: float) -> float:
    return _lib.oracle_halide_log(x)


def halide_pow(x: float, y: float) -> float:
    return _lib.oracle_halide_pow(x, y)


def fast_exp(x: float) -> float:
    return _lib.oracle_fast_exp(x)


def ll_remap_lut(levels: int, alpha: float) -> np.ndarray:
    lut = np.zeros(2 * (levels - 1) * 256 + 1, np.float32)
    _lib.oracle_ll_remap_lut(levels, alpha, lut)
    return lut


def omp_threads() -> int:
    return int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))


# canonicalisation variants of the local_laplacian oracle (oracle/local_laplacian_oracle.c header); 0 = canonical
LL_VAR_SOURCE, LL_VAR_FMA, LL_VAR_DIV = 1, 2, 4
_lib.oracle_ll_set_variant.argtypes = [C.c_int]
_lib.oracle_ll_gray_constants.argtypes = [_f32p]


def ll_gray_constants() -> np.ndarray:
    c = np.zeros(3, np.float32)
    _lib.oracle_ll_gray_constants(c)
    return c


def local_laplacian(inp: np.ndarray, levels: int, alpha: float, beta: float, J: int = 8, origin=(0, 0),
                    variant: int = 0) -> np.ndarray:
    """inp: u16 array of shape (3, H, W) (planar). Returns the same shape.  `origin` = (min_x, min_y) of the
    buffer in the pipeline's absolute coordinate system (the 2x-1 pyramid taps depend on it).  `variant` selects a
    non-canonical evaluation (LL_VAR_*) for the canonicalisation study only."""
    inp = np.ascontiguousarray(inp, np.uint16)
    c, h, w = inp.shape
    assert c == 3
    out = np.zeros_like(inp)
    prev = get_canon()
    if variant & LL_VAR_FMA:   # the fma variant IS canon 1 (halide_exp's polynomial is contracted with the rest)
        set_canon(1)
    _lib.oracle_ll_set_variant(int(variant))
    try:
        r = _lib.oracle_local_laplacian(inp, w, h, w, w * h, int(origin[0]), int(origin[1]), J, levels, alpha, beta, out,
                                        w, w * h, -1, None)
    finally:
        _lib.oracle_ll_set_variant(0)
        set_canon(prev)
    assert r == 0
    return out



def set_threads(n: int) -> None:
    """OpenMP threads of the oracle library from now on (n <= 0: the default, every processor)."""
    _lib.oracle_set_threads(int(n))


def local_laplacian_fast(inp: np.ndarray, levels: int, alpha: float, beta: float, J: int = 8, origin=(0, 0)) -> np.ndarray:
    """The tuned CPU evaluation (oracle/local_laplacian_fast_oracle.c): same operations, CPU-friendly schedule; the
    cpu_baseline leg of bench.py times this one, tests/test_local_laplacian.py pins it to the oracle bit for bit."""
    inp = np.ascontiguousarray(inp, np.uint16)
    c, h, w = inp.shape
    assert c == 3
    out = np.zeros_like(inp)
    r = _lib.oracle_local_laplacian_fast(inp, w, h, w, w * h, int(origin[0]), int(origin[1]), J, levels, alpha, beta, out, w, w * h)
    assert r == 0
    return out


def local_laplacian_outg(inp: np.ndarray, levels: int, alpha: float, beta: float, level: int, J: int = 8,
                         origin=(0, 0)) -> np.ndarray:
    """outGPyramid[level] on its required region R_level, as an (rh, rw) float32 array (debug aid)."""
    inp = np.ascontiguousarray(inp, np.uint16)
    c, h, w = inp.shape
    x0, x1, y0, y1 = origin[0], origin[0] + w - 1, origin[1], origin[1] + h - 1
    for _ in range(level):
        x0, x1, y0, y1 = (x0 - 1) // 2, (x1 + 1) // 2, (y0 - 1) // 2, (y1 + 1) // 2
    dbg = np.zeros((y1 - y0 + 1, x1 - x0 + 1), np.float32)
    out = np.zeros_like(inp)
    r = _lib.oracle_local_laplacian(inp, w, h, w, w * h, int(origin[0]), int(origin[1]), J, levels, alpha, beta, out,
                                    w, w * h, level, dbg.ctypes.data_as(C.c_void_p))
    assert r == 0
    return dbg

_lib.oracle_blur.argtypes = [_u16p, C.c_int, C.c_int, C.c_int, _u16p, C.c_int]
_lib.oracle_blur.restype = C.c_int


def blur(inp: np.ndarray) -> np.ndarray:
    """inp: u16 (H+2, W+2) -> (H, W)."""
    inp = np.ascontiguousarray(inp, np.uint16)
    h, w = inp.shape[0] - 2, inp.shape[1] - 2
    out = np.zeros((h, w), np.uint16)
    assert _lib.oracle_blur(inp, inp.shape[1], w, h, out, w) == 0
    return out

_lib.oracle_stencil_chain.argtypes = [_u16p, C.c_int, C.c_int, C.c_int, C.c_int, _u16p, C.c_int]
_lib.oracle_stencil_chain.restype = C.c_int


def stencil_chain(inp: np.ndarray, stencils: int = 32) -> np.ndarray:
    inp = np.ascontiguousarray(inp, np.uint16)
    h, w = inp.shape
    out = np.zeros_like(inp)
    assert _lib.oracle_stencil_chain(inp, w, w, h, stencils, out, w) == 0
    return out

_lib.oracle_bilateral_grid.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _f32p, C.c_int]
_lib.oracle_bilateral_grid.restype = C.c_int


def bilateral_grid(inp: np.ndarray, r_sigma: float, origin=(0, 0)) -> np.ndarray:
    inp = np.ascontiguousarray(inp, np.float32)
    h, w = inp.shape
    out = np.zeros_like(inp)
    assert _lib.oracle_bilateral_grid(inp, w, h, w, int(origin[0]), int(origin[1]), r_sigma, out, w) == 0
    return out

_lib.oracle_nl_means.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _f32p, C.c_int,
                                 C.c_int]
_lib.oracle_nl_means.restype = C.c_int


def nl_means(inp: np.ndarray, patch: int, search: int, sigma: float) -> np.ndarray:
    """inp: f32 (3, H, W) planar."""
    inp = np.ascontiguousarray(inp, np.float32)
    c, h, w = inp.shape
    assert c == 3
    out = np.zeros_like(inp)
    assert _lib.oracle_nl_means(inp, w, h, w, w * h, patch, search, sigma, out, w, w * h) == 0
    return out

_lib.oracle_conv_layer.argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
_lib.oracle_conv_layer.restype = C.c_int


def conv_layer(inp: np.ndarray, filt: np.ndarray, bias: np.ndarray) -> np.ndarray:
    """inp: (N, H+2, W+2, CI); filt: (CI, 3, 3, CO) [= halide (CO, kx, ky, CI) reversed]; bias: (CO,) -> (N, H, W, CO)."""
    inp, filt, bias = (np.ascontiguousarray(a, np.float32) for a in (inp, filt, bias))
    n, hp, wp, ci = inp.shape
    co = bias.shape[0]
    assert filt.shape == (ci, 3, 3, co)
    out = np.zeros((n, hp - 2, wp - 2, co), np.float32)
    assert _lib.oracle_conv_layer(inp, filt, bias, out, ci, co, wp - 2, hp - 2, n) == 0
    return out

_lib.oracle_conv_layer_bf16.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
_lib.oracle_conv_layer_bf16.restype = C.c_int


def conv_layer_bf16(inp: np.ndarray, filt: np.ndarray, bias: np.ndarray):
    """Operands rounded to bfloat16 (nearest even), double accumulation.  Returns (relu, mag) where
    mag = |bias| + sum |products| per output, the scale of the accumulation error of an f32 MFMA chain."""
    inp, filt, bias = (np.ascontiguousarray(a, np.float32) for a in (inp, filt, bias))
    n, hp, wp, ci = inp.shape
    co = bias.shape[0]
    assert filt.shape == (ci, 3, 3, co)
    out = np.zeros((n, hp - 2, wp - 2, co), np.float32)
    mag = np.zeros_like(out)
    assert _lib.oracle_conv_layer_bf16(inp, filt, bias, out, mag, ci, co, wp - 2, hp - 2, n) == 0
    return out, mag

_lib.oracle_depthwise_separable_conv.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p] + [C.c_int] * 9
_lib.oracle_depthwise_separable_conv.restype = C.c_int


def depthwise_separable_conv(inp: np.ndarray, dw: np.ndarray, pw: np.ndarray, bias: np.ndarray) -> np.ndarray:
    """inp (N, H, W, CI); dw (FH, FW, IC, CM); pw (IC, CO); bias (CO,) [numpy axes = halide dims reversed] -> (N, H, W, CO)."""
    inp, dw, pw, bias = (np.ascontiguousarray(a, np.float32) for a in (inp, dw, pw, bias))
    n, h, w, ci = inp.shape
    fh, fw, ic, cm = dw.shape
    ic2, co = pw.shape
    assert ic == ic2 and bias.shape == (co,)
    out = np.zeros((n, h, w, co), np.float32)
    assert _lib.oracle_depthwise_separable_conv(inp, dw, pw, bias, out, ci, w, h, n, cm, fw, fh, ic, co) == 0
    return out

_lib.oracle_unsharp.argtypes = [_f32p, C.c_int, C.c_int, C.c_long, C.c_long, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_long, C.c_long]
_lib.oracle_unsharp.restype = C.c_int
_lib.oracle_unsharp_kernel.argtypes = [_f32p]
_lib.oracle_unsharp_kernel.restype = None


def unsharp_kernel() -> np.ndarray:
    k = np.zeros(4, np.float32)
    _lib.oracle_unsharp_kernel(k)
    return k


def unsharp(inp: np.ndarray, out_origin=(0, 0), out_size=None, in_origin=(0, 0)) -> np.ndarray:
    """inp: f32 (3, H, W) planar whose first element sits at absolute in_origin; output region out_origin + out_size."""
    inp = np.ascontiguousarray(inp, np.float32)
    c, h, w = inp.shape
    assert c == 3
    ow, oh = out_size if out_size else (w, h)
    out = np.zeros((3, oh, ow), np.float32)
    assert _lib.oracle_unsharp(inp, w, h, w, w * h, in_origin[0], in_origin[1], out, out_origin[0], out_origin[1], ow, oh,
                               ow, ow * oh) == 0
    return out

_lib.oracle_max_filter.argtypes = [_f32p, C.c_int, C.c_int, C.c_long, C.c_long, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_long, C.c_long]
_lib.oracle_max_filter.restype = C.c_int


_lib.oracle_max_filter_tables.argtypes = [np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")] * 2
_lib.oracle_max_filter_tables.restype = None


def max_filter_tables():
    """(slice_for_radius[0..27], filter_height[dx = -26..26]) as the oracle computes them."""
    sfr, fh = np.zeros(28, np.int32), np.zeros(53, np.int32)
    _lib.oracle_max_filter_tables(sfr, fh)
    return sfr, fh


def max_filter(inp: np.ndarray, out_origin=(0, 0), out_size=None, in_origin=(0, 0)) -> np.ndarray:
    """inp: f32 (C, H, W) planar whose first element sits at absolute in_origin; output region out_origin + out_size (any
    region: every tap is edge-clamped)."""
    inp = np.ascontiguousarray(inp, np.float32)
    c, h, w = inp.shape
    ow, oh = out_size if out_size else (w, h)
    out = np.zeros((c, oh, ow), np.float32)
    assert _lib.oracle_max_filter(inp, w, h, w, w * h, in_origin[0], in_origin[1], out, out_origin[0], out_origin[1], ow, oh, c,
                                  ow, ow * oh) == 0
    return out

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_lib.oracle_hist.argtypes = [_u8p, C.c_int, C.c_int, C.c_long, C.c_long, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long,
                             _i32p]
_lib.oracle_hist.restype = C.c_int


def hist(inp: np.ndarray, out_origin=(0, 0), out_size=None, return_cdf=False):
    """inp: u8 (3, H, W) planar with origin (0, 0); output region out_origin + out_size (default: the whole image)."""
    inp = np.ascontiguousarray(inp, np.uint8)
    c, h, w = inp.shape
    assert c == 3
    ow, oh = out_size if out_size else (w, h)
    out = np.zeros((3, oh, ow), np.uint8)
    cdf = np.zeros(256, np.int32)
    assert _lib.oracle_hist(inp, w, h, w, w * h, out, out_origin[0], out_origin[1], ow, oh, ow, ow * oh, cdf) == 0
    return (out, cdf) if return_cdf else out

_lib.oracle_harris.argtypes = [_f32p, C.c_long, C.c_long, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long]
_lib.oracle_harris.restype = C.c_int


def harris(inp: np.ndarray, out_origin=(3, 3), out_size=None, in_origin=(0, 0)) -> np.ndarray:
    """inp: f32 (3, H, W) planar at absolute in_origin; output region out_origin + out_size (default: the driver's W-6 x H-6 at (3, 3))."""
    inp = np.ascontiguousarray(inp, np.float32)
    c, h, w = inp.shape
    assert c == 3
    ow, oh = out_size if out_size else (w - 6, h - 6)
    out = np.zeros((oh, ow), np.float32)
    assert _lib.oracle_harris(inp, w, w * h, in_origin[0], in_origin[1], out, out_origin[0], out_origin[1], ow, oh, ow) == 0
    return out

_lib.oracle_interpolate.argtypes = [_f32p, C.c_int, C.c_int, C.c_long, C.c_long, _f32p, C.c_long, C.c_long]
_lib.oracle_interpolate.restype = C.c_int
_lib.oracle_interpolate_boxes.argtypes = [C.c_int, C.c_int, _i32p, _i32p]
_lib.oracle_interpolate_boxes.restype = None


def interpolate(inp: np.ndarray) -> np.ndarray:
    """inp: f32 (4, H, W) planar r, g, b, alpha -> (3, H, W)."""
    inp = np.ascontiguousarray(inp, np.float32)
    c, h, w = inp.shape
    assert c == 4
    out = np.zeros((3, h, w), np.float32)
    assert _lib.oracle_interpolate(inp, w, h, w, w * h, out, w, w * h) == 0
    return out


def interpolate_boxes(w: int, h: int):
    bi, bd = np.zeros(40, np.int32), np.zeros(40, np.int32)
    _lib.oracle_interpolate_boxes(w, h, bi, bd)
    return bi.reshape(10, 4), bd.reshape(10, 4)

_lib.oracle_iir_blur.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_float, _f32p, C.c_long, C.c_long]
_lib.oracle_iir_blur.restype = C.c_int


def iir_blur(inp: np.ndarray, alpha: float) -> np.ndarray:
    """inp: f32 (C, H, W) planar."""
    inp = np.ascontiguousarray(inp, np.float32)
    c, h, w = inp.shape
    out = np.zeros_like(inp)
    assert _lib.oracle_iir_blur(inp, w, h, c, w, w * h, alpha, out, w, w * h) == 0
    return out

_i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_lib.oracle_camera_pipe.argtypes = [_u16p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.c_int]
_lib.oracle_camera_pipe.restype = C.c_int
_lib.oracle_camera_pipe_setup.argtypes = [_f32p, _f32p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _i16p,
                                          _u8p, _u8p]


def camera_pipe(raw: np.ndarray, m3200: np.ndarray, m7000: np.ndarray, color_temp, gamma, contrast, sharpen, black, white,
                out_w: int, out_h: int) -> np.ndarray:
    """raw: u16 (H_in, W_in); matrices f32 (3, 4); returns u8 (3, out_h, out_w)."""
    raw = np.ascontiguousarray(raw, np.uint16)
    m3200, m7000 = np.ascontiguousarray(m3200, np.float32), np.ascontiguousarray(m7000, np.float32)
    out = np.zeros((3, out_h, out_w), np.uint8)
    r = _lib.oracle_camera_pipe(raw, raw.shape[1], raw.shape[0], raw.shape[1], m3200, m7000, color_temp, gamma, contrast,
                                sharpen, black, white, out, out_w, out_h, out_w, out_w * out_h)
    assert r == 0, r
    return out


def camera_pipe_setup(m3200, m7000, color_temp, gamma, contrast, sharpen, black, white):
    matrix, curve, s = np.zeros(12, np.int16), np.zeros(1024, np.uint8), np.zeros(1, np.uint8)
    _lib.oracle_camera_pipe_setup(np.ascontiguousarray(m3200, np.float32), np.ascontiguousarray(m7000, np.float32), color_temp,
                                  gamma, contrast, sharpen, black, white, matrix, curve, s)
    return matrix.reshape(3, 4), curve, int(s[0])


# ---- lens_blur
_lib.oracle_lens_blur.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _f32p,
                                  C.c_void_p]
_lib.oracle_lens_blur.restype = C.c_int
_lib.oracle_lens_blur_random.argtypes = [C.c_int] * 5
_lib.oracle_lens_blur_random.restype = C.c_float
_lib.oracle_lens_blur_default_tag.restype = C.c_int


def lens_blur_default_tag() -> int:
    return _lib.oracle_lens_blur_default_tag()


def lens_blur_random(call_id, tag, s, y, x) -> float:
    return float(_lib.oracle_lens_blur_random(call_id, tag, s, y, x))


def lens_blur(left: np.ndarray, right: np.ndarray, slices=32, focus_depth=13, blur_radius_scale=0.5, aperture_samples=32, tag=None,
              return_depth=False):
    """left / right: u8 (3, H, W) planar -> f32 (3, H, W); optionally also depth on the image grown by R on every side."""
    left, right = np.ascontiguousarray(left, np.uint8), np.ascontiguousarray(right, np.uint8)
    _, h, w = left.shape
    _, rh, rw = right.shape
    out = np.zeros((3, h, w), np.float32)
    md = max(slices - focus_depth, focus_depth)
    R = int(np.float32(md) * np.float32(blur_radius_scale))
    depth = np.zeros((h + 2 * R, w + 2 * R), np.int32)
    t = lens_blur_default_tag() if tag is None else tag
    assert _lib.oracle_lens_blur(left, w, h, right, rw, rh, slices, focus_depth, blur_radius_scale, aperture_samples, t, out,
                                 depth.ctypes.data) == 0
    return (out, depth) if return_depth else out


# ---- bgu
BGU_CANONICAL, BGU_X86_RCP = 0, 1
_lib.oracle_bgu.argtypes = [C.c_float, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_int,
                            C.c_int, C.c_int, C.c_int, C.c_int, _f32p, C.c_int, C.c_void_p, C.c_void_p]
_lib.oracle_bgu.restype = C.c_int


def bgu(r_sigma, s_sigma, splat_loc: np.ndarray, values: np.ndarray, slice_loc: np.ndarray, region=None, variant=BGU_CANONICAL,
        return_line=False):
    """splat_loc / values: f32 (C, h, w) planar low-res pair; slice_loc: f32 (3, H, W) -> f32 (3, oh, ow) on
    region = (x0, y0, ow, oh) (default: all of slice_loc); optionally also the fitted transforms (ncy, ncx, nz, 12) and
    (cx0, cy0, ncx, ncy, nz, big_sigma)."""
    splat_loc, values = np.ascontiguousarray(splat_loc, np.float32), np.ascontiguousarray(values, np.float32)
    slice_loc = np.ascontiguousarray(slice_loc, np.float32)
    _, H, W = slice_loc.shape
    x0, y0, ow, oh = region if region is not None else (0, 0, W, H)
    out = np.zeros((3, oh, ow), np.float32)
    dims = (C.c_int * 6)()
    lc, lh, lw = splat_loc.shape
    vc, vh, vw = values.shape
    args = [np.float32(r_sigma), s_sigma, splat_loc, lw, lh, lc, values, vw, vh, vc, slice_loc, W, H, x0, y0, ow, oh, out, variant]
    if not return_line:
        assert _lib.oracle_bgu(*args, None, None) == 0
        return out
    assert _lib.oracle_bgu(*args, None, C.cast(dims, C.c_void_p)) == 0
    cx0, cy0, ncx, ncy, nz, big = list(dims)
    line = np.zeros((ncy, ncx, nz, 12), np.float32)
    assert _lib.oracle_bgu(*args, line.ctypes.data, C.cast(dims, C.c_void_p)) == 0
    return out, line, (cx0, cy0, ncx, ncy, nz, big)
