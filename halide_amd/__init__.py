"""halide_amd — host-side mirror (ctypes) of the C-ABI drop-in library ``libhlmi.so``.

The product is the shared library: hand-written HIP kernels for gfx950 behind the reference's AOT
entry points (``int local_laplacian(halide_buffer_t*, int32_t, float, float, halide_buffer_t*)`` …,
see ``include/hlmi_pipelines.h``).  This package is only the thin Python caller used by the tests and
``bench.py``: a ``Buffer`` that plays the role of ``Halide::Runtime::Buffer`` (reference:
``src/runtime/HalideBuffer.h`` — planar storage, dimension 0 innermost, dirty flags, ``copy_to_host``,
``device_sync``) and one Python function per pipeline with the reference's argument order.

There is deliberately no CPU fallback: if ``libhlmi.so`` is missing the import fails, and if no
gfx950 device is usable every pipeline call raises ``HalideError`` (code -29).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HLMI_LIB: an alternative build of the same library (A/B measurements of compile-time switches, csrc/Makefile VARIANT=)
LIB_PATH = os.environ.get("HLMI_LIB") or os.path.join(_HERE, "lib", "libhlmi.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C halide_amd/csrc`). halide_amd has no pure-Python / CPU fallback.")



def _elf_dynamic_strings(path: str, tag: int) -> list:
    """Values of the string-valued entries `tag` (1 = DT_NEEDED, 14 = DT_SONAME) of a little-endian ELF64 shared object's
    dynamic section; [] when the file cannot be parsed."""
    import struct
    try:
        with open(path, "rb") as f:
            data = f.read()
        if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:
            return []
        shoff, = struct.unpack_from("<Q", data, 0x28)
        shentsize, shnum = struct.unpack_from("<HH", data, 0x3A)
        secs = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
        out = []
        for sec in secs:
            if sec[1] != 6:      # SHT_DYNAMIC
                continue
            strtab = secs[sec[6]]
            for off in range(sec[4], sec[4] + sec[5], 16):
                t, v = struct.unpack_from("<qQ", data, off)
                if t == 0:
                    break
                if t == tag:
                    a = strtab[4] + v
                    out.append(data[a:data.index(b"\0", a)].decode())
        return out
    except Exception:  # noqa: BLE001
        return []


def _preload_torch_hip_runtime() -> str | None:
    """One HIP runtime per process.  The torch-ROCm wheel bundles its own libamdhip64.so (same SONAME as the system's): a
    process that loads libhlmi.so first and torch afterwards ends up with TWO runtimes — the system's under libhlmi.so and
    the bundled one under torch — whose streams and events are not interchangeable (torch streams handed to the library,
    tests/test_torch_ops.py, then belong to the other runtime: std::bad_variant_access or a segfault inside HIP).  When
    torch is importable its copy is therefore loaded FIRST, without importing torch, and libhlmi.so binds to it exactly as
    it does when the caller imported torch first.  HLMI_SYSTEM_HIP=1 keeps the system runtime (torch-free processes)."""
    if os.environ.get("HLMI_SYSTEM_HIP"):
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except Exception:  # noqa: BLE001
        return None
    if not spec or not spec.submodule_search_locations:
        return None
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    # libhlmi.so was linked against the system ROCm; torch's copy is only a safe stand-in when it IS the same ABI: its
    # SONAME must be the libamdhip64 name libhlmi.so lists as NEEDED.  On a mismatch the system runtime is kept (a
    # process that then also imports torch has to import torch FIRST) unless HLMI_TORCH_HIP=1 insists.
    import sys
    if "torch" not in sys.modules and os.environ.get("HLMI_TORCH_HIP") != "1":
        needed = [n for n in _elf_dynamic_strings(LIB_PATH, 1) if n.startswith("libamdhip64")]
        soname = _elf_dynamic_strings(path, 14)
        if not needed or not soname or soname[0] != needed[0]:
            import warnings
            warnings.warn(f"halide_amd: torch bundles {soname[0] if soname else 'an unidentified HIP runtime'} but libhlmi.so needs "
                          f"{needed[0] if needed else 'libamdhip64'}: keeping the system HIP runtime — import torch BEFORE halide_amd "
                          "in a process that uses both (INTEGRATION.md), or set HLMI_TORCH_HIP=1", RuntimeWarning, stacklevel=2)
            return None
    try:
        C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        return None
    return path


HIP_RUNTIME_PRELOADED = _preload_torch_hip_runtime()
lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


def hip_runtime() -> C.CDLL:
    """The HIP runtime libhlmi.so is bound to (global symbol scope of the process): callers that create streams or events
    of their own for the library must use THIS runtime, not whatever dlopen("libamdhip64.so") finds."""
    return C.CDLL(None)



# ---------------------------------------------------------------------------------------------------
# ABI structs (include/hlmi_abi.h)
class halide_type_t(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("reserved", C.c_uint16)]


class halide_dimension_t(C.Structure):
    _fields_ = [("min", C.c_int32), ("extent", C.c_int32), ("stride", C.c_int32), ("flags", C.c_uint32)]


class halide_buffer_t(C.Structure):
    _fields_ = [("device", C.c_uint64), ("device_interface", C.c_void_p), ("host", C.c_void_p),
                ("flags", C.c_uint64), ("type", halide_type_t), ("dimensions", C.c_int32),
                ("dim", C.POINTER(halide_dimension_t)), ("padding", C.c_void_p)]


class halide_scalar_value_t(C.Union):
    _fields_ = [("b", C.c_uint8), ("i32", C.c_int32), ("i64", C.c_int64), ("u64", C.c_uint64),
                ("f32", C.c_float), ("f64", C.c_double), ("handle", C.c_void_p)]


class halide_filter_argument_t(C.Structure):
    _fields_ = [("name", C.c_char_p), ("kind", C.c_int32), ("dimensions", C.c_int32), ("type", halide_type_t),
                ("scalar_def", C.POINTER(halide_scalar_value_t)), ("scalar_min", C.POINTER(halide_scalar_value_t)),
                ("scalar_max", C.POINTER(halide_scalar_value_t)),
                ("scalar_estimate", C.POINTER(halide_scalar_value_t)),
                ("buffer_estimates", C.POINTER(C.POINTER(C.c_int64)))]


class halide_filter_metadata_t(C.Structure):
    _fields_ = [("version", C.c_int32), ("num_arguments", C.c_int32),
                ("arguments", C.POINTER(halide_filter_argument_t)), ("target", C.c_char_p), ("name", C.c_char_p)]


assert C.sizeof(halide_buffer_t) == 56 and C.sizeof(halide_dimension_t) == 16 and C.sizeof(halide_type_t) == 4

FLAG_HOST_DIRTY = 1
FLAG_DEVICE_DIRTY = 2

_TYPE_OF = {np.dtype(np.uint8): (1, 8), np.dtype(np.uint16): (1, 16), np.dtype(np.uint32): (1, 32),
            np.dtype(np.int8): (0, 8), np.dtype(np.int16): (0, 16), np.dtype(np.int32): (0, 32),
            np.dtype(np.float32): (2, 32), np.dtype(np.float64): (2, 64)}

ERROR_NAMES = {0: "success", -1: "generic_error", -3: "bad_type", -4: "access_out_of_bounds",
               -5: "buffer_allocation_too_large", -6: "buffer_extents_too_large", -8: "constraint_violated",
               -9: "param_too_small", -10: "param_too_large", -12: "buffer_argument_is_null",
               -14: "copy_to_host_failed", -15: "copy_to_device_failed", -16: "device_malloc_failed",
               -19: "no_device_interface", -20: "unimplemented", -23: "device_run_failed",
               -28: "buffer_extents_negative", -29: "gpu_device_error", -34: "host_is_null",
               -36: "device_interface_no_device", -37: "host_and_device_dirty", -38: "buffer_is_null",
               -42: "incompatible_device_interface", -43: "bad_dimensions",
               -44: "device_dirty_with_no_device_support"}


class HalideError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"halide error {code} ({ERROR_NAMES.get(code, '?')}): {message}")
        self.code = code
        self.message = message


# ---------------------------------------------------------------------------------------------------
# error handler: the library's default prints and abort()s like the reference
# (src/runtime/posix_error_handler.cpp:9-21); under Python we record the message and raise instead.
_ERR_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)
_tls = threading.local()


def _on_error(_uc, msg):
    _tls.last_error = msg.decode("utf-8", "replace") if msg else ""
    if _batch_active[0] and not _batch_error:   # a worker thread of hlmi_run_batch: keep the batch's FIRST message
        _batch_error.append(_tls.last_error)


_error_cb = _ERR_CB(_on_error)
lib.halide_set_error_handler.restype = C.c_void_p
lib.halide_set_error_handler.argtypes = [_ERR_CB]
lib.halide_set_error_handler(_error_cb)


def last_error() -> str:
    return getattr(_tls, "last_error", "")


def _check(code: int) -> int:
    if code != 0:
        msg = last_error()
        _tls.last_error = ""
        raise HalideError(code, msg)
    return code


# ---------------------------------------------------------------------------------------------------
_BP = C.POINTER(halide_buffer_t)
lib.halide_hip_device_interface.restype = C.c_void_p
lib.halide_device_sync.argtypes = [C.c_void_p, _BP]
lib.halide_copy_to_host.argtypes = [C.c_void_p, _BP]
lib.halide_copy_to_device.argtypes = [C.c_void_p, _BP, C.c_void_p]
lib.halide_device_free.argtypes = [C.c_void_p, _BP]
lib.halide_device_malloc.argtypes = [C.c_void_p, _BP, C.c_void_p]
lib.halide_hip_wrap_device_ptr.argtypes = [C.c_void_p, _BP, C.c_uint64]
lib.halide_hip_detach_device_ptr.argtypes = [C.c_void_p, _BP]
lib.halide_hip_get_device_ptr.argtypes = [C.c_void_p, _BP]
lib.halide_hip_get_device_ptr.restype = C.c_size_t
lib.halide_set_gpu_device.argtypes = [C.c_int]
lib.halide_hip_set_stream.argtypes = [C.c_void_p]
lib.halide_hip_get_stream.argtypes = [C.c_void_p]
lib.halide_hip_get_stream.restype = C.c_void_p
lib.halide_device_release.argtypes = [C.c_void_p, C.c_void_p]
lib.halide_reuse_device_allocations.argtypes = [C.c_void_p, C.c_bool]
lib.hlmi_kernel_timing_enable.argtypes = [C.c_int]
lib.hlmi_kernel_timing_report.argtypes = [C.c_char_p, C.c_size_t]
lib.hlmi_kernel_timing_only.argtypes = [C.c_char_p]
lib.hlmi_kernel_timing_report.restype = C.c_size_t
lib.hlmi_version.restype = C.c_char_p
lib.hlmi_canon_fma.restype = C.c_int


def hip_device_interface() -> int:
    return lib.halide_hip_device_interface()


def set_gpu_device(n: int) -> None:
    lib.halide_set_gpu_device(int(n))


def set_stream(stream_ptr: int | None) -> None:
    """Enqueue all subsequent work of this thread on `stream_ptr` (a hipStream_t, e.g.
    torch.cuda.current_stream().cuda_stream); None restores the library's own stream."""
    lib.halide_hip_set_stream(C.c_void_p(stream_ptr or 0))


def partition_stream(part: int, nparts: int, replica: int = 0) -> int | None:
    """hipStream_t of frame queue `part` of `nparts` (a library-owned stream with a hardware queue of its own whose launches are
    sized for `nparts` frames in flight; include/hlmi_runtime.h — the name dates from when its CU mask was believed to confine
    it), or None; `replica` > 0: a further queue of the same kind."""
    lib.halide_hip_partition_stream_replica.restype = C.c_void_p
    lib.halide_hip_partition_stream_replica.argtypes = [C.c_int, C.c_int, C.c_int]
    return lib.halide_hip_partition_stream_replica(int(part), int(nparts), int(replica))


def kernel_timing(enable: bool) -> None:
    lib.hlmi_kernel_timing_enable(1 if enable else 0)


def kernel_timing_only(name) -> None:
    """Measurement only: while `name` is set, every kernel launch with another timing name is skipped (hlmi_runtime.h)."""
    lib.hlmi_kernel_timing_only(name.encode() if name else None)


def kernel_timing_reset() -> None:
    lib.hlmi_kernel_timing_reset()


def kernel_timing_report() -> list:
    import json
    n = lib.hlmi_kernel_timing_report(None, 0)
    buf = C.create_string_buffer(n + 16)
    lib.hlmi_kernel_timing_report(buf, n + 16)
    return json.loads(buf.value.decode())


class Buffer:
    """Caller-side image descriptor, the role Halide::Runtime::Buffer<T> plays for the reference's drivers.

    Storage is a numpy array in C order whose axes are the Halide dimensions REVERSED (the convention of the
    reference's Python bindings, src/PythonExtensionGen.cpp): ``np.zeros((3, H, W))`` is a planar W x H x 3
    image with dimension 0 (x) innermost, exactly ``Buffer<T,3>(W, H, 3)`` (HalideBuffer.h:441-451).
    """

    def __init__(self, array: np.ndarray | None = None, *, dtype=None, shape_xyz=None, mins=None):
        if array is None:
            array = np.zeros(tuple(reversed(shape_xyz)), dtype=dtype)
        self.array = array
        nd = array.ndim
        if array.dtype not in _TYPE_OF:
            raise TypeError(f"unsupported dtype {array.dtype}")
        self._dims = (halide_dimension_t * max(nd, 1))()
        for i in range(nd):
            ax = nd - 1 - i
            st = array.strides[ax]
            assert st % array.itemsize == 0
            self._dims[i] = halide_dimension_t(0 if mins is None else int(mins[i]), array.shape[ax],
                                               st // array.itemsize, 0)
        code, bits = _TYPE_OF[array.dtype]
        self.raw = halide_buffer_t(0, None, array.ctypes.data, FLAG_HOST_DIRTY, halide_type_t(code, bits, 0), nd,
                                   self._dims, None)

    # -- construction helpers --------------------------------------------------------------------
    @classmethod
    def bounds_query(cls, dtype, ndim: int, mins=None, extents=None) -> "Buffer":
        """A buffer with host == NULL and device == 0 (HalideRuntime.h:1851-1853)."""
        b = cls.__new__(cls)
        b.array = None
        b._dims = (halide_dimension_t * max(ndim, 1))()
        stride = 1
        for i in range(ndim):
            e = 0 if extents is None else int(extents[i])
            b._dims[i] = halide_dimension_t(0 if mins is None else int(mins[i]), e, stride, 0)
            stride *= max(e, 1)
        code, bits = _TYPE_OF[np.dtype(dtype)]
        b.raw = halide_buffer_t(0, None, None, 0, halide_type_t(code, bits, 0), ndim, b._dims, None)
        return b

    @classmethod
    def wrap_device(cls, device_ptr: int, dtype, extents, strides=None, mins=None) -> "Buffer":
        """Device-only buffer around an existing allocation (mirrors halide_cuda_wrap_device_ptr)."""
        nd = len(extents)
        b = cls.bounds_query(dtype, nd, mins, extents)
        if strides is not None:
            for i in range(nd):
                b._dims[i].stride = int(strides[i])
        _check(lib.halide_hip_wrap_device_ptr(None, C.byref(b.raw), C.c_uint64(device_ptr)))
        return b

    # -- accessors -----------------------------------------------------------------------------------
    @property
    def ptr(self):
        return C.byref(self.raw)

    def dim(self, i: int) -> halide_dimension_t:
        return self._dims[i]

    @property
    def extents(self):
        return [self._dims[i].extent for i in range(self.raw.dimensions)]

    @property
    def mins(self):
        return [self._dims[i].min for i in range(self.raw.dimensions)]

    def set_min(self, *mins) -> "Buffer":
        for i, m in enumerate(mins):
            self._dims[i].min = int(m)
        return self

    def set_host_dirty(self, v: bool = True) -> None:
        self.raw.flags = (self.raw.flags | FLAG_HOST_DIRTY) if v else (self.raw.flags & ~FLAG_HOST_DIRTY)

    @property
    def host_dirty(self) -> bool:
        return bool(self.raw.flags & FLAG_HOST_DIRTY)

    @property
    def device_dirty(self) -> bool:
        return bool(self.raw.flags & FLAG_DEVICE_DIRTY)

    @property
    def has_device_allocation(self) -> bool:
        return self.raw.device != 0

    # -- device protocol (HalideBuffer.h:1810-1815, :1908) -------------------------------------
    def copy_to_host(self) -> "Buffer":
        _check(lib.halide_copy_to_host(None, self.ptr))
        return self

    def copy_to_device(self) -> "Buffer":
        _check(lib.halide_copy_to_device(None, self.ptr, hip_device_interface()))
        return self

    def device_sync(self) -> "Buffer":
        if self.raw.device_interface:
            _check(lib.halide_device_sync(None, self.ptr))
        return self

    def device_free(self) -> None:
        if self.raw.device:
            _check(lib.halide_device_free(None, self.ptr))

    def device_detach(self) -> None:
        if self.raw.device:
            _check(lib.halide_hip_detach_device_ptr(None, self.ptr))

    def numpy(self) -> np.ndarray:
        """Host array, after bringing device results back if the device copy is newer."""
        if self.device_dirty:
            self.copy_to_host()
        return self.array

    def __del__(self):
        try:
            if getattr(self, "raw", None) is not None and self.raw.device:
                lib.halide_device_free(None, C.byref(self.raw))
        except Exception:
            pass


def _as_ptr(b):
    if b is None:
        return None
    return b.ptr if isinstance(b, Buffer) else b


# ---------------------------------------------------------------------------------------------------
# pipelines — argument order and meaning exactly as in include/hlmi_pipelines.h
def _bind(name, argtypes):
    fn = getattr(lib, name, None)
    if fn is None:
        return None
    fn.argtypes = argtypes
    fn.restype = C.c_int
    return fn


_ll = _bind("local_laplacian", [_BP, C.c_int32, C.c_float, C.c_float, _BP])
_bg = _bind("bilateral_grid", [_BP, C.c_float, _BP])
_blur = _bind("halide_blur", [_BP, _BP])
_nlm = _bind("nl_means", [_BP, C.c_int32, C.c_int32, C.c_float, _BP])
_sc = _bind("stencil_chain", [_BP, _BP])
_conv = _bind("conv_layer", [_BP, _BP, _BP, _BP])
_conv_bf16 = _bind("conv_layer_bf16", [_BP, _BP, _BP, _BP])
_dsc = _bind("depthwise_separable_conv", [_BP, _BP, _BP, _BP, _BP])
_unsharp = _bind("unsharp", [_BP, _BP])
_maxf = _bind("max_filter", [_BP, _BP])
_hist = _bind("hist", [_BP, _BP])
_harris = _bind("harris", [_BP, _BP])
_interp = _bind("interpolate", [_BP, _BP])
_iir = _bind("iir_blur", [_BP, C.c_float, _BP])
_lens = _bind("lens_blur", [_BP, _BP, C.c_int32, C.c_int32, C.c_float, C.c_int32, _BP])
_bgu = _bind("bgu", [C.c_float, C.c_int32, _BP, _BP, _BP, _BP])
_cam = _bind("camera_pipe", [_BP, _BP, _BP, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, _BP])


def local_laplacian(input, levels, alpha, beta, output) -> int:
    """apps/local_laplacian: u16 [W,H,3] -> u16 [W,H,3]; drivers pass alpha/(levels-1) (process.cpp:31)."""
    return _check(_ll(_as_ptr(input), int(levels), float(alpha), float(beta), _as_ptr(output)))


def debug_local_laplacian_outg(level: int) -> np.ndarray:
    """Test hook: outGPyramid[level] (restricted to the region R_level) of this thread's last
    local_laplacian call, as an (rh, rw) float32 array."""
    fn = lib.hlmi_debug_local_laplacian_outg
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    rw, rh = C.c_int(0), C.c_int(0)
    if fn(level, None, 0, C.byref(rw), C.byref(rh)) != 0:
        raise HalideError(-1, "no local_laplacian call to inspect")
    out = np.zeros((rh.value, rw.value), np.float32)
    if fn(level, out.ctypes.data_as(C.c_void_p), out.size, None, None) != 0:
        raise HalideError(-1, "debug copy failed")
    return out


def membench_naive(nbytes: int = 1 << 30, iters: int = 10, blocks: int = 0) -> dict:
    """The round-1/2 figure: naive grid-stride float4 copy / read-only / write-only kernels over buffers of `nbytes`."""
    fn = lib.hlmi_membench
    fn.restype = C.c_int
    fn.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double)]
    out = (C.c_double * 3)()
    _check(fn(int(nbytes), int(iters), int(blocks), out))
    return {"copy_gbs": out[0], "read_gbs": out[1], "write_gbs": out[2]}


def membench_sweep(nbytes: int = 1 << 30, iters: int = 10) -> dict:
    """Every (loads in flight, workgroups per CU, temporal policy) variant of the streaming kernels + hipMemcpyDtoD
    (csrc/membench.hip); the full table, as the library measured it."""
    import json
    fn = lib.hlmi_membench_sweep
    fn.restype = C.c_long
    fn.argtypes = [C.c_size_t, C.c_int, C.c_char_p, C.c_size_t]
    n = fn(int(nbytes), int(iters), None, 0)
    if n < 0:
        _check(int(n))
    buf = C.create_string_buffer(n + 16)
    fn(int(nbytes), int(iters), buf, n + 16)
    return json.loads(buf.value.decode())


def membench_widths(nbytes: int = 1 << 30, iters: int = 4) -> dict:
    """Read-only kernels that fetch `nbytes` exactly once with 4 / 8 / 16-byte aligned, 8-byte misaligned and overlapping
    8-byte loads: the known-bytes calibration runs for rocprofv3's FETCH_SIZE (csrc/membench.hip)."""
    fn = lib.hlmi_membench_widths
    fn.restype = C.c_int
    fn.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    out = (C.c_double * 5)()
    _check(fn(int(nbytes), int(iters), out))
    return dict(zip(("ld4_gbs", "ld8_gbs", "ld16_gbs", "ld8u_gbs", "ld8o_gbs"), (round(v, 1) for v in out)))


def membench(nbytes: int = 1 << 30, iters: int = 10, blocks: int = 0) -> dict:
    """Measurement hook: the practical HBM ceiling the pipelines' roofline fractions can be read against — the BEST copy /
    read-only / write-only rate over the sweep of streaming-kernel variants, beside the naive kernel's and hipMemcpyDtoD's."""
    naive = membench_naive(nbytes, iters, blocks)
    sw = membench_sweep(nbytes, iters)
    best = sw["best"]
    return {"copy_gbs": max(best["copy"]["gbs"], naive["copy_gbs"]), "read_gbs": max(best["read"]["gbs"], naive["read_gbs"]),
            "write_gbs": max(best["write"]["gbs"], naive["write_gbs"]), "best_variant": best,
            "naive": {k: round(v, 1) for k, v in naive.items()}, "memcpy_d2d_gbs": sw["memcpy_d2d_gbs"],
            "bytes_per_buffer": sw["bytes"], "guide_copy_gbs": 6290.0}


def bilateral_grid(input, r_sigma, output) -> int:
    return _check(_bg(_as_ptr(input), float(r_sigma), _as_ptr(output)))


def halide_blur(input, blur_y) -> int:
    return _check(_blur(_as_ptr(input), _as_ptr(blur_y)))


def nl_means(input, patch_size, search_area, sigma, output) -> int:
    return _check(_nlm(_as_ptr(input), int(patch_size), int(search_area), float(sigma), _as_ptr(output)))


def stencil_chain(input, output) -> int:
    return _check(_sc(_as_ptr(input), _as_ptr(output)))


def conv_layer(input, filter, bias, relu) -> int:
    return _check(_conv(_as_ptr(input), _as_ptr(filter), _as_ptr(bias), _as_ptr(relu)))


def conv_layer_bf16(input, filter, bias, relu) -> int:
    return _check(_conv_bf16(_as_ptr(input), _as_ptr(filter), _as_ptr(bias), _as_ptr(relu)))


def depthwise_separable_conv(input, depthwise_filter, pointwise_filter, bias, output) -> int:
    return _check(_dsc(_as_ptr(input), _as_ptr(depthwise_filter), _as_ptr(pointwise_filter), _as_ptr(bias), _as_ptr(output)))


def unsharp(input, output) -> int:
    return _check(_unsharp(_as_ptr(input), _as_ptr(output)))


def max_filter(input, output) -> int:
    return _check(_maxf(_as_ptr(input), _as_ptr(output)))


def hist(input, output) -> int:
    return _check(_hist(_as_ptr(input), _as_ptr(output)))


def harris(input, output) -> int:
    return _check(_harris(_as_ptr(input), _as_ptr(output)))


def interpolate(input, output) -> int:
    return _check(_interp(_as_ptr(input), _as_ptr(output)))


def iir_blur(input, alpha, output) -> int:
    return _check(_iir(_as_ptr(input), float(alpha), _as_ptr(output)))


def lens_blur(left_im, right_im, slices, focus_depth, blur_radius_scale, aperture_samples, final) -> int:
    return _check(_lens(_as_ptr(left_im), _as_ptr(right_im), int(slices), int(focus_depth), float(blur_radius_scale),
                        int(aperture_samples), _as_ptr(final)))


def bgu(r_sigma, s_sigma, splat_loc, values, slice_loc, output) -> int:
    """apps/bgu: low-res f32 pair (splat_loc -> values) fitted per bilateral-grid cell, applied to the full-res slice_loc."""
    return _check(_bgu(float(r_sigma), int(s_sigma), _as_ptr(splat_loc), _as_ptr(values), _as_ptr(slice_loc), _as_ptr(output)))


def camera_pipe(input, matrix_3200, matrix_7000, color_temp, gamma, contrast, sharpen_strength, black_level,
                white_level, processed) -> int:
    return _check(_cam(_as_ptr(input), _as_ptr(matrix_3200), _as_ptr(matrix_7000), float(color_temp), float(gamma),
                       float(contrast), float(sharpen_strength), int(black_level), int(white_level),
                       _as_ptr(processed)))


def device_count() -> int:
    """Usable gfx950 devices visible to this process."""
    lib.hlmi_device_count.restype = C.c_int
    return lib.hlmi_device_count()


def run_batch(name: str, frames, devices=None, streams_per_device: int = 1) -> int:
    """The in-process frame sharder (`hlmi_run_batch`, include/hlmi_runtime.h): run pipeline `name` once
    per frame, frames dealt round-robin to one host thread + HIP stream per entry of `devices` (x streams_per_device).

    `frames` is a list of argument tuples in the pipeline's own order (Buffers and Python scalars), exactly what
    `<name>_argv` receives per call.  Outputs are left device-dirty on the device that produced them; `Buffer.numpy()`
    brings them back from there.  Returns 0 or raises HalideError with the first failing frame's code."""
    md = metadata(name)
    n_args = md.num_arguments
    fn = getattr(lib, name + "_argv")
    keep = []          # ctypes objects that must outlive the call
    argvs = (C.POINTER(C.c_void_p) * max(len(frames), 1))()
    for i, frame in enumerate(frames):
        if len(frame) != n_args:
            raise TypeError(f"{name} takes {n_args} arguments, frame {i} has {len(frame)}")
        argv = (C.c_void_p * n_args)()
        for j, v in enumerate(frame):
            a = md.arguments[j]
            if a.kind == 0:
                box = C.c_float(v) if a.type.code == 2 else (C.c_int32(v) if a.type.bits == 32 else C.c_int64(v))
                keep.append(box)
                argv[j] = C.cast(C.pointer(box), C.c_void_p)
            else:
                argv[j] = C.cast(C.pointer(v.raw), C.c_void_p)
        keep.append(argv)
        argvs[i] = C.cast(argv, C.POINTER(C.c_void_p))
    if devices is None:
        devices = list(range(max(device_count(), 1)))
    devs = (C.c_int * len(devices))(*devices)
    lib.hlmi_run_batch.restype = C.c_int
    lib.hlmi_run_batch.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_void_p)), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]
    with _batch_lock:      # one batch at a time records messages: a HalideError never carries another batch's text
        del _batch_error[:]
        _batch_active[0] = True
        try:
            code = lib.hlmi_run_batch(C.cast(fn, C.c_void_p), argvs, len(frames), devs, len(devices), int(streams_per_device))
        finally:
            _batch_active[0] = False
        msg = _batch_error[0] if _batch_error else ""
    if code != 0:
        raise HalideError(code, msg)
    return 0


# worker threads of hlmi_run_batch report through the same handler but on their own threads; messages are recorded only
# while a batch is running (and cleared when it starts), so unrelated calls on other Python threads leave no stale text
_batch_error: list = []
_batch_active = [False]
_batch_lock = threading.Lock()


def metadata(name: str) -> halide_filter_metadata_t:
    fn = getattr(lib, name + "_metadata")
    fn.restype = C.POINTER(halide_filter_metadata_t)
    return fn().contents


def version() -> str:
    return lib.hlmi_version().decode()


def canon_fma() -> int:
    """1: the library's float kernels contract mul+add pairs into fma (the default build), 0: one rounding per operator."""
    return int(lib.hlmi_canon_fma())
