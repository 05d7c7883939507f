"""The entry prologue of EVERY pipeline: return codes and the order of the argument checks.

Reference: test/generator/error_codes_aottest.cpp:27-120 (the cases), src/UnpackBuffers.cpp:148 (null arguments),
src/AddImageChecks.cpp:315-347, 393-470, 591-671, 716-760 (the checks and the order they are emitted in; buffers are
visited in name order), src/AddParameterChecks.cpp (scalar ranges, emitted before the image checks: src/Lower.cpp:189
vs :251), src/runtime/posix_error_handler.cpp:9-41 (the default handler prints and aborts; halide_set_error_handler
returns the previous handler).

Every case calls through `<name>_argv` (src/CodeGen_C.cpp:688-694), so the argv convention is exercised for all
entry points as well.  Argument-check failures are reported before the library looks for a GPU, which is why most of
this file also runs on a box without one (the host logic of the C ABI); the cases that need a device are marked gpu.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# output shape (halide order, dimension 0 first) + scalar arguments of a small valid call per pipeline;
# input shapes come from the pipeline's own bounds query, like tools/RunGen.h:1212-1250 obtains them
SPEC = {
    "local_laplacian": dict(out=(64, 48, 3), scalars=[8, 1.0 / 7.0, 1.0]),
    "bilateral_grid": dict(out=(64, 48), scalars=[0.1]),
    "halide_blur": dict(out=(64, 48), scalars=[]),
    "nl_means": dict(out=(32, 24, 3), scalars=[7, 7, 0.12]),
    "stencil_chain": dict(out=(64, 48), scalars=[]),
    "conv_layer": dict(out=(128, 8, 6, 2), scalars=[]),
    "conv_layer_bf16": dict(out=(128, 8, 6, 2), scalars=[]),
    "depthwise_separable_conv": dict(out=(16, 12, 10, 2), scalars=[]),
    "unsharp": dict(out=(64, 48, 3), scalars=[]),
    "max_filter": dict(out=(64, 48, 3), scalars=[]),
    "hist": dict(out=(64, 48, 3), scalars=[]),
    "harris": dict(out=(40, 30), scalars=[], out_min=(3, 3)),
    "interpolate": dict(out=(64, 48, 3), scalars=[]),
    "iir_blur": dict(out=(64, 48, 3), scalars=[0.5]),
    "camera_pipe": dict(out=(64, 32, 3), scalars=[3700.0, 2.0, 50.0, 1.0, 25, 1023]),
    "lens_blur": dict(out=(48, 40, 3), scalars=[32, 13, 0.5, 32]),
    "bgu": dict(out=(64, 48, 3), scalars=[0.125, 16]),
}
PIPELINES = sorted(SPEC)
PRIMARY = {"lens_blur": "left_im", "bgu": "slice_loc"}   # the image input the generic cases perturb ("input" everywhere else)
# pipelines whose generator pins mins / extents / strides of its buffers (conv_layer_generator.cpp:35-50,
# nl_means_generator.cpp:68, …): a malformed shape trips a constraint (-8) before the generic shape checks
PINNED = {"conv_layer", "conv_layer_bf16"}
# ... and those that tie the output's box to the input's (iir_blur_generator.cpp:160-166 set_bounds; interpolate
# :23, :83-87 bound(); depthwise_separable_conv_generator.cpp:77-99): for them a shrunken input is a constraint
# violation, not an out-of-bounds access
TIED = PINNED | {"iir_blur", "interpolate"}
# pipelines that read their input ONLY through repeat_edge (stencil_chain_generator.cpp:20, nl_means_generator.cpp:28,
# max_filter_generator.cpp:22): bounds inference clamps the required region to whatever was passed, so no input
# is ever too small
CLAMPED = {"stencil_chain", "nl_means", "max_filter", "lens_blur"}   # lens_blur_generator.cpp:27-28


def _np_type(t):
    return {(0, 8): np.int8, (0, 16): np.int16, (0, 32): np.int32, (1, 8): np.uint8, (1, 16): np.uint16,
            (1, 32): np.uint32, (2, 32): np.float32, (2, 64): np.float64}[(t.code, t.bits)]


class Call:
    """A valid argument vector for `<name>_argv`, every piece of which a test may then damage."""

    def __init__(self, hl, name):
        self.hl, self.name = hl, name
        md = hl.metadata(name)
        self.args = [md.arguments[i] for i in range(md.num_arguments)]
        self.kinds = [a.kind for a in self.args]
        self.names = [a.name.decode() for a in self.args]
        spec = SPEC[name]
        scalars = list(spec["scalars"])
        self.values = []   # hl.Buffer | ctypes scalar
        for a in self.args:
            if a.kind == 0:
                v = scalars.pop(0)
                self.values.append(C.c_float(v) if a.type.code == 2 else C.c_int32(v))
            elif a.kind == 2:
                b = hl.Buffer(np.zeros(tuple(reversed(spec["out"])), _np_type(a.type)))
                if "out_min" in spec:
                    b.set_min(*spec["out_min"])
                self.values.append(b)
            else:
                self.values.append(hl.Buffer.bounds_query(_np_type(a.type), a.dimensions))
        fn = getattr(hl.lib, name + "_argv")
        fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_void_p)]
        self.fn = fn
        assert self.run() == 0, "bounds query failed"
        for i, a in enumerate(self.args):   # allocate what the query asked for
            if a.kind == 1:
                q = self.values[i]
                ext, mins = q.extents, q.mins
                assert all(e > 0 for e in ext), (name, self.names[i], ext)
                self.values[i] = hl.Buffer(np.zeros(tuple(reversed(ext)), _np_type(a.type))).set_min(*mins)
        self.buffers = [i for i, k in enumerate(self.kinds) if k != 0]
        self.inputs = [i for i, k in enumerate(self.kinds) if k == 1]
        self.output = self.kinds.index(2)

    def run(self, null=None):
        argv = (C.c_void_p * len(self.values))()
        for i, v in enumerate(self.values):
            if i == null:
                argv[i] = None
            elif isinstance(v, self.hl.Buffer):
                argv[i] = C.cast(C.pointer(v.raw), C.c_void_p)
            else:
                argv[i] = C.cast(C.pointer(v), C.c_void_p)
        return self.fn(argv)

    def code(self, **kw):
        self.hl._tls.last_error = ""
        return self.run(**kw)

    def buf(self, i):
        return self.values[i]

    def first_by_name(self, idxs):
        return min(idxs, key=lambda i: self.names[i])


@pytest.fixture(scope="module")
def calls(hl):
    cache = {}

    def get(name):
        # a fresh, undamaged argument vector per use (bounds queries need no device)
        return Call(hl, name)
    get.cache = cache
    return get


def test_set_error_handler_returns_the_previous_one(hl):
    """posix_error_handler.cpp:40: halide_set_error_handler swaps and returns the old handler."""
    lib = hl.lib
    mine = hl._ERR_CB(lambda uc, msg: None)
    prev = lib.halide_set_error_handler(mine)
    assert prev == C.cast(hl._error_cb, C.c_void_p).value
    back = lib.halide_set_error_handler(hl._error_cb)
    assert back == C.cast(mine, C.c_void_p).value


@pytest.mark.parametrize("name", PIPELINES)
def test_null_buffer_argument(calls, name):
    c = calls(name)
    for i in c.buffers:
        assert c.code(null=i) == -12, c.names[i]          # halide_error_code_buffer_argument_is_null
        assert c.names[i] in c.hl.last_error()


@pytest.mark.parametrize("name", PIPELINES)
def test_bad_type(calls, name):
    for i in calls(name).buffers:
        c = calls(name)
        t = c.buf(i).raw.type
        t.bits = 64 if t.bits != 64 else 32
        assert c.code() == -3, c.names[i]                 # halide_error_code_bad_type
        assert c.names[i] in c.hl.last_error()


@pytest.mark.parametrize("name", PIPELINES)
def test_bad_dimensions(calls, name):
    for i in calls(name).buffers:
        c = calls(name)
        c.buf(i).raw.dimensions -= 1 if c.buf(i).raw.dimensions > 1 else -1
        if c.buf(i).raw.dimensions > len(c.buf(i)._dims):
            continue  # would need a larger dim[] than the buffer owns
        assert c.code() == -43, c.names[i]                # halide_error_code_bad_dimensions


@pytest.mark.parametrize("name", PIPELINES)
def test_stride0_must_be_one(calls, name):
    """src/Parameter.cpp:30-35 + error_codes_aottest.cpp:95-102."""
    for i in calls(name).buffers:
        c = calls(name)
        c.buf(i).dim(0).stride = 2
        assert c.code() == -8, c.names[i]                 # halide_error_code_constraint_violated
        assert "stride.0" in c.hl.last_error() and c.names[i] in c.hl.last_error()


@pytest.mark.parametrize("name", PIPELINES)
def test_input_too_small_is_out_of_bounds(calls, name):
    """error_codes_aottest.cpp:49-57.  Only for the buffer named `input`: the filters / matrices of a pipeline have
    pinned extents (a constraint, -8)."""
    c = calls(name)
    i = c.names.index(PRIMARY.get(name, "input"))
    c.buf(i).dim(0).extent -= 1
    if name in TIED:
        assert c.code() == -8 and "Constraint violated" in c.hl.last_error()
    elif name in CLAMPED:
        assert c.code() in (0, -29)                        # passes every argument check (-29: this box has no GPU)
    else:
        assert c.code() == -4                              # halide_error_code_access_out_of_bounds
        assert c.names[i] in c.hl.last_error()


@pytest.mark.parametrize("name", [n for n in PIPELINES if n not in TIED | {"nl_means", "depthwise_separable_conv"}])
def test_negative_extent(calls, name):
    """error_codes_aottest.cpp:59-71: negative extents "in a way that doesn't trigger oob checks" — here the output's,
    which only shrinks what is required of the inputs."""
    c = calls(name)
    o = c.buf(c.output)
    o.dim(1).extent = -o.dim(1).extent
    assert c.code() == -28                                 # halide_error_code_buffer_extents_negative


@pytest.mark.parametrize("name", [n for n in PIPELINES if n not in TIED | {"depthwise_separable_conv"}])
def test_too_large(calls, name):
    """error_codes_aottest.cpp:73-92: a product of extents beyond 2^31-1 (-6) and |extent * stride| beyond it (-5)."""
    c = calls(name)
    i = c.names.index(PRIMARY.get(name, "input"))
    b = c.buf(i)
    keep = [(b.dim(d).min, b.dim(d).extent, b.dim(d).stride) for d in range(2)]
    b.dim(0).min, b.dim(1).min = min(keep[0][0], 0), min(keep[1][0], 0)
    b.dim(0).extent, b.dim(1).extent, b.dim(1).stride = 10000000, 10000000, 64
    assert c.code() == -6                                  # halide_error_code_buffer_extents_too_large
    for d in range(2):
        b.dim(d).min, b.dim(d).extent, b.dim(d).stride = keep[d]
    b.dim(1).stride = 0x7fffffff
    assert c.code() == -5                                  # halide_error_code_buffer_allocation_too_large


def test_scalar_parameter_ranges(calls):
    """error_codes_aottest.cpp:104-113 (param_too_small / param_too_large).  The reference apps declare no ranges;
    this implementation's own limits are reported with the same codes, before any image check (src/Lower.cpp:189)."""
    c = calls("local_laplacian")
    c.values[1] = C.c_int32(1)
    assert c.code() == -9
    # `levels` has no upper bound in the reference (generator :13); the library's only limit is where its 32-bit table
    # index arithmetic ends (2^20 levels) — 33 and beyond are ordinary values (tests/test_local_laplacian.py runs 33 and 40)
    c.values[1] = C.c_int32((1 << 20) + 1)
    assert c.code() == -10
    c.buf(0).dim(0).stride = 2          # a bad image argument does not pre-empt the parameter check
    assert c.code() == -10


def test_check_order(calls):
    """The order of src/AddImageChecks.cpp:716-760: null, [query], type/dimensions, constraints, required region and
    negative extents, overflow — buffers in name order within each phase."""
    # null beats everything
    c = calls("local_laplacian")
    c.buf(0).raw.type.bits = 8
    assert c.code(null=c.output) == -12
    # type/dimensions (phase 3) before the stride constraint (phase 4), whatever the buffers
    c = calls("local_laplacian")
    c.buf(0).dim(0).stride = 2
    c.buf(c.output).raw.dimensions = 2
    assert c.code() == -43
    # per buffer: type, then dimensions; buffers in NAME order ("bilateral_grid" < "input")
    c = calls("bilateral_grid")
    c.buf(c.names.index("input")).raw.type.bits = 64
    c.buf(c.output).raw.dimensions = 1
    assert c.code() == -43
    c = calls("bilateral_grid")
    c.buf(c.names.index("input")).raw.dimensions = 1
    c.buf(c.output).raw.type.bits = 64
    assert c.code() == -3
    # constraint (phase 4) before out-of-bounds (phase 5)
    c = calls("local_laplacian")
    c.buf(c.output).dim(0).stride = 2
    c.buf(0).dim(0).extent -= 1
    assert c.code() == -8
    # out-of-bounds (phase 5) before overflow (phase 6)
    c = calls("local_laplacian")
    c.buf(0).dim(0).extent -= 1
    c.buf(c.output).dim(1).stride = 0x7fffffff
    assert c.code() == -4
    # within phase 5, buffers in name order: "input" (out of bounds) < "output" (negative extent) ...
    c = calls("local_laplacian")
    c.buf(0).dim(0).extent -= 1
    c.buf(c.output).dim(2).extent *= -1
    assert c.code() == -4
    # ... and "blur_y" (negative extent) < "input" (out of bounds)
    c = calls("halide_blur")
    c.buf(0).dim(0).extent -= 1
    c.buf(c.output).dim(1).extent *= -1
    assert c.code() == -28
    # within one buffer and dimension: required region before the sign of the extent
    c = calls("local_laplacian")
    i = c.buf(0)
    i.dim(1).extent = -i.dim(1).extent
    assert c.code() == -4


def test_bounds_query_skips_the_type_check_and_rewrites_the_type(calls, hl):
    """In query mode the reference returns before its type checks and rewrites the query buffers completely
    (src/AddImageChecks.cpp:478-497, :709-713)."""
    q = hl.Buffer.bounds_query(np.float32, 3)          # wrong type on purpose
    out = hl.Buffer(np.zeros((3, 48, 64), np.uint16))
    assert hl.local_laplacian(q, 8, 1.0 / 7.0, 1.0, out) == 0
    assert (q.raw.type.code, q.raw.type.bits) == (1, 16) and q.extents == [64, 48, 3]


def test_default_error_handler_prints_and_aborts(hl):
    """src/runtime/posix_error_handler.cpp:9-21: without a custom handler an error is fatal."""
    code = ("import ctypes as C; lib = C.CDLL(%r); lib.local_laplacian.argtypes = [C.c_void_p, C.c_int, C.c_float, "
            "C.c_float, C.c_void_p]; lib.local_laplacian(None, 8, C.c_float(1), C.c_float(1), None); print('survived')") % hl.LIB_PATH
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == -6 and "survived" not in r.stdout      # SIGABRT
    assert "Error: Buffer argument input is nullptr" in r.stderr


# ---------------------------------------------------------------------------------------------------------------
# with a device
@pytest.mark.gpu
@pytest.mark.parametrize("name", PIPELINES)
def test_valid_call_through_argv_succeeds(calls, name):
    c = calls(name)
    assert c.code() == 0
    out = c.buf(c.output)
    assert out.device_dirty and out.has_device_allocation
    out.copy_to_host()
    assert not out.device_dirty


@pytest.mark.gpu
def test_dirty_flag_and_interface_errors(calls, hl):
    """src/runtime/device_interface.cpp:84-128 (validation order of the runtime) and the copy protocol of a GPU target
    (src/InjectHostDevBufferCopies.cpp:197-217): a GPU-target pipeline never reports -44; a buffer that arrives with
    another API's device allocation is -42 (halide_copy_to_device: "does not support switching interfaces")."""
    c = calls("stencil_chain")
    b = c.buf(0)
    b.raw.flags = hl.FLAG_HOST_DIRTY | hl.FLAG_DEVICE_DIRTY
    assert c.code() == -37                                 # halide_error_code_host_and_device_dirty
    b.raw.flags = hl.FLAG_HOST_DIRTY
    b.raw.device = 0x1000                                  # a device handle without an interface
    assert c.code() == -19                                 # halide_error_code_no_device_interface
    b.raw.device = 0
    b.raw.device_interface = hl.hip_device_interface()     # an interface without a device handle
    assert c.code() == -36                                 # halide_error_code_device_interface_no_device
    b.raw.device_interface = None
    fake = (C.c_void_p * 16)()                             # some other API's interface table
    b.raw.device, b.raw.device_interface = 0x1000, C.addressof(fake)
    b.raw.flags = hl.FLAG_DEVICE_DIRTY
    assert c.code() == -42                                 # halide_error_code_incompatible_device_interface
    b.raw.device, b.raw.device_interface, b.raw.flags = 0, None, hl.FLAG_HOST_DIRTY
    assert c.code() == 0


@pytest.mark.gpu
def test_host_null_with_stale_device_copy(calls, hl):
    """src/AddImageChecks.cpp:648-655 / device_interface.cpp:170-176: a host-dirty input needs a host pointer (-34);
    a device-only buffer (host null, device set, not host-dirty) is a valid argument."""
    c = calls("stencil_chain")
    assert c.code() == 0
    b = c.buf(0)
    host = b.raw.host
    b.raw.host = None
    assert c.code() == 0                                   # device copy is current: nothing to upload
    b.raw.flags |= hl.FLAG_HOST_DIRTY
    assert c.code() == -34                                 # halide_error_code_host_is_null
    b.raw.host = host
    assert c.code() == 0
