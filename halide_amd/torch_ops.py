"""PyTorch-ROCm operators over libhlmi.so — SURVEY.md §8(f4).

The reference puts AOT pipelines behind torch through generated wrappers that build a ``halide_buffer_t`` around a
tensor's storage with the dimensions reversed (``src/runtime/HalidePyTorchHelpers.h:28-120``, ``apps/HelloPyTorch``).
This module does the same against the C ABI, without a copy: a CUDA(HIP) tensor's ``data_ptr()`` is attached with
``halide_hip_wrap_device_ptr`` (counterpart of ``halide_cuda_wrap_device_ptr``, ``src/runtime/HalideRuntimeCuda.h:44-58``),
the library enqueues on torch's current stream (``halide_hip_set_stream``), the result lands in a tensor allocated by
torch, and the wrapper detaches before returning so the library never owns torch memory.

    import torch, halide_amd.torch_ops          # registers torch.ops.hlmi.*
    out = torch.ops.hlmi.local_laplacian(img_u16_cuda, 8, 1 / 7, 1.0)       # (3, H, W) uint16 -> same

Tensor axes are the Halide dimensions REVERSED (innermost last), as in the reference's Python bindings: an image
``Buffer<uint16_t, 3>(W, H, 3)`` is a tensor of shape ``(3, H, W)``.  There is no CPU fallback: CPU tensors are an error.
"""
from __future__ import annotations

import numpy as np
import torch

import halide_amd as hl

_NP = {torch.uint8: np.uint8, torch.uint16: np.uint16, torch.int16: np.int16, torch.int32: np.int32, torch.float32: np.float32}


class _Wrapped:
    """`with _Wrapped(t0, t1, ...) as (b0, b1, ...)`: zero-copy halide buffers on torch's current stream."""

    def __init__(self, *tensors):
        self.tensors, self.bufs = tensors, []

    def __enter__(self):
        for t in self.tensors:
            if not t.is_cuda:
                raise RuntimeError("hlmi ops need tensors on the GPU (there is no CPU implementation)")
            if t.dtype not in _NP:
                raise TypeError(f"unsupported dtype {t.dtype}")
            if t.dim() and t.stride(-1) != 1:
                raise RuntimeError("the innermost dimension must be dense (halide dim[0].stride == 1)")
            self.bufs.append(hl.Buffer.wrap_device(t.data_ptr(), _NP[t.dtype], list(reversed(t.shape)),
                                                   list(reversed(t.stride()))))
        # torch's default stream is the NULL stream (handle 0), which halide_hip_set_stream reads as "the library's own
        # stream": name it by HIP's explicit handle hipStreamLegacy (= 1) instead
        s = torch.cuda.current_stream(self.tensors[0].device).cuda_stream
        hl.set_stream(s if s else 1)
        hl.set_gpu_device(self.tensors[0].device.index or 0)
        return self.bufs

    def __exit__(self, *exc):
        hl.set_stream(None)
        for b in self.bufs:
            b.device_detach()   # the storage belongs to torch
        return False


@torch.library.custom_op("hlmi::local_laplacian", mutates_args=())
def local_laplacian(input: torch.Tensor, levels: int, alpha: float, beta: float) -> torch.Tensor:
    """apps/local_laplacian: (3, H, W) uint16 -> (3, H, W) uint16; `alpha` as the drivers pass it (alpha / (levels - 1))."""
    out = torch.empty_like(input)
    with _Wrapped(input, out) as (a, o):
        hl.local_laplacian(a, levels, alpha, beta, o)
    return out


@torch.library.custom_op("hlmi::bilateral_grid", mutates_args=())
def bilateral_grid(input: torch.Tensor, r_sigma: float) -> torch.Tensor:
    """apps/bilateral_grid: (H, W) float32 -> (H, W) float32, s_sigma = 8."""
    out = torch.empty_like(input)
    with _Wrapped(input, out) as (a, o):
        hl.bilateral_grid(a, r_sigma, o)
    return out


@torch.library.custom_op("hlmi::nl_means", mutates_args=())
def nl_means(input: torch.Tensor, patch_size: int, search_area: int, sigma: float) -> torch.Tensor:
    """apps/nl_means: (3, H, W) float32 -> (3, H, W) float32."""
    out = torch.empty_like(input)
    with _Wrapped(input, out) as (a, o):
        hl.nl_means(a, patch_size, search_area, sigma, o)
    return out


@torch.library.custom_op("hlmi::stencil_chain", mutates_args=())
def stencil_chain(input: torch.Tensor) -> torch.Tensor:
    """apps/stencil_chain: (H, W) uint16 -> (H, W) uint16, 32 stages."""
    out = torch.empty_like(input)
    with _Wrapped(input, out) as (a, o):
        hl.stencil_chain(a, o)
    return out


@torch.library.custom_op("hlmi::blur", mutates_args=())
def blur(input: torch.Tensor) -> torch.Tensor:
    """apps/blur: (H + 2, W + 2) uint16 -> (H, W) uint16."""
    out = torch.empty((input.shape[0] - 2, input.shape[1] - 2), dtype=input.dtype, device=input.device)
    with _Wrapped(input, out) as (a, o):
        hl.halide_blur(a, o)
    return out


def _conv(fn, input, filter, bias):
    n, hp, wp, _ = input.shape
    out = torch.empty((n, hp - 2, wp - 2, bias.shape[0]), dtype=torch.float32, device=input.device)
    with _Wrapped(input, filter, bias, out) as (a, f, b, o):
        fn(a, f, b, o)
    return out


@torch.library.custom_op("hlmi::conv_layer", mutates_args=())
def conv_layer(input: torch.Tensor, filter: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """apps/conv_layer, exact f32: input (N, H+2, W+2, CI), filter (CI, 3, 3, CO), bias (CO,) -> relu (N, H, W, CO)."""
    return _conv(hl.conv_layer, input, filter, bias)


@torch.library.custom_op("hlmi::conv_layer_bf16", mutates_args=())
def conv_layer_bf16(input: torch.Tensor, filter: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """Same buffers as conv_layer; operands rounded to bf16, f32 accumulation on the matrix cores."""
    return _conv(hl.conv_layer_bf16, input, filter, bias)


@torch.library.custom_op("hlmi::depthwise_separable_conv", mutates_args=())
def depthwise_separable_conv(input: torch.Tensor, depthwise_filter: torch.Tensor, pointwise_filter: torch.Tensor,
                             bias: torch.Tensor) -> torch.Tensor:
    """apps/depthwise_separable_conv: input (N, H, W, CI), depthwise (FH, FW, IC, CM), pointwise (IC, CO), bias (CO,)."""
    n, h, w, _ = input.shape
    out = torch.empty((n, h, w, bias.shape[0]), dtype=torch.float32, device=input.device)
    with _Wrapped(input, depthwise_filter, pointwise_filter, bias, out) as (a, d, p, b, o):
        hl.depthwise_separable_conv(a, d, p, b, o)
    return out


@torch.library.custom_op("hlmi::unsharp", mutates_args=())
def unsharp(input: torch.Tensor) -> torch.Tensor:
    """apps/unsharp: (3, H, W) float32 -> (3, H, W) float32, sigma = 1.5."""
    out = torch.empty_like(input)
    with _Wrapped(input, out) as (a, o):
        hl.unsharp(a, o)
    return out


@torch.library.custom_op("hlmi::max_filter", mutates_args=())
def max_filter(input: torch.Tensor) -> torch.Tensor:
    """apps/max_filter: (3, H, W) float32 -> (3, H, W) float32, max over the radius-26 footprint of the edge-clamped input."""
    out = torch.empty_like(input)
    with _Wrapped(input, out) as (a, o):
        hl.max_filter(a, o)
    return out


@torch.library.custom_op("hlmi::hist", mutates_args=())
def hist(input: torch.Tensor) -> torch.Tensor:
    """apps/hist: (3, H, W) uint8 -> (3, H, W) uint8, histogram equalisation of the luma."""
    out = torch.empty_like(input)
    with _Wrapped(input, out) as (a, o):
        hl.hist(a, o)
    return out


@torch.library.custom_op("hlmi::camera_pipe", mutates_args=())
def camera_pipe(input: torch.Tensor, matrix_3200: torch.Tensor, matrix_7000: torch.Tensor, color_temp: float, gamma: float,
                contrast: float, sharpen_strength: float, black_level: int, white_level: int, out_width: int,
                out_height: int) -> torch.Tensor:
    """apps/camera_pipe: raw (IH, IW) uint16 Bayer -> (3, out_height, out_width) uint8."""
    out = torch.empty((3, out_height, out_width), dtype=torch.uint8, device=input.device)
    with _Wrapped(input, matrix_3200, matrix_7000, out) as (a, m3, m7, o):
        hl.camera_pipe(a, m3, m7, color_temp, gamma, contrast, sharpen_strength, black_level, white_level, o)
    return out


@torch.library.custom_op("hlmi::harris", mutates_args=())
def harris(input: torch.Tensor) -> torch.Tensor:
    """apps/harris: (3, H, W) float32 -> (H - 6, W - 6) float32 corner response of the interior (the driver's region:
    output origin (3, 3) in the input's frame, apps/harris/filter.cpp:22-28)."""
    c, h, w = input.shape
    out = torch.empty((h - 6, w - 6), dtype=torch.float32, device=input.device)
    with _Wrapped(input, out) as (a, o):
        o.set_min(3, 3)
        hl.harris(a, o)
    return out


@torch.library.custom_op("hlmi::interpolate", mutates_args=())
def interpolate(input: torch.Tensor) -> torch.Tensor:
    """apps/interpolate: (4, H, W) float32 RGBA -> (3, H, W) float32, alpha-weighted pull-push."""
    out = torch.empty((3,) + tuple(input.shape[1:]), dtype=torch.float32, device=input.device)
    with _Wrapped(input, out) as (a, o):
        hl.interpolate(a, o)
    return out


@torch.library.custom_op("hlmi::iir_blur", mutates_args=())
def iir_blur(input: torch.Tensor, alpha: float) -> torch.Tensor:
    """apps/iir_blur: (C, H, W) float32 -> same, first-order IIR low pass down / up the columns, then along the rows."""
    out = torch.empty_like(input)
    with _Wrapped(input, out) as (a, o):
        hl.iir_blur(a, alpha, o)
    return out


@torch.library.custom_op("hlmi::lens_blur", mutates_args=())
def lens_blur(left_im: torch.Tensor, right_im: torch.Tensor, slices: int, focus_depth: int, blur_radius_scale: float,
              aperture_samples: int) -> torch.Tensor:
    """apps/lens_blur: two (3, H, W) uint8 views -> (3, H, W) float32 (depth from stereo, depth-dependent bokeh)."""
    out = torch.empty(tuple(left_im.shape), dtype=torch.float32, device=left_im.device)
    with _Wrapped(left_im, right_im, out) as (a, b, o):
        hl.lens_blur(a, b, slices, focus_depth, blur_radius_scale, aperture_samples, o)
    return out


@torch.library.custom_op("hlmi::bgu", mutates_args=())
def bgu(r_sigma: float, s_sigma: int, splat_loc: torch.Tensor, values: torch.Tensor, slice_loc: torch.Tensor) -> torch.Tensor:
    """apps/bgu: low-res (3, h, w) float32 pair splat_loc -> values, applied to the full-res (3, H, W) slice_loc."""
    out = torch.empty_like(slice_loc)
    with _Wrapped(splat_loc, values, slice_loc, out) as (a, b, c, o):
        hl.bgu(r_sigma, s_sigma, a, b, c, o)
    return out


# shape functions for torch.compile / meta tensors
@local_laplacian.register_fake
def _(input, levels, alpha, beta):
    return torch.empty_like(input)


@bilateral_grid.register_fake
def _(input, r_sigma):
    return torch.empty_like(input)


@nl_means.register_fake
def _(input, patch_size, search_area, sigma):
    return torch.empty_like(input)


@stencil_chain.register_fake
def _(input):
    return torch.empty_like(input)


@harris.register_fake
def _(input):
    return input.new_empty((input.shape[1] - 6, input.shape[2] - 6))


@interpolate.register_fake
def _(input):
    return input.new_empty((3,) + tuple(input.shape[1:]))


@iir_blur.register_fake
def _(input, alpha):
    return torch.empty_like(input)


@lens_blur.register_fake
def _(left_im, right_im, slices, focus_depth, blur_radius_scale, aperture_samples):
    return left_im.new_empty(tuple(left_im.shape), dtype=torch.float32)


@bgu.register_fake
def _(r_sigma, s_sigma, splat_loc, values, slice_loc):
    return torch.empty_like(slice_loc)
