"""torch.ops.hlmi.*: zero-copy tensor <-> halide_buffer_t wrappers (reference: src/runtime/HalidePyTorchHelpers.h:28-120,
apps/HelloPyTorch).  Results must equal the oracle exactly, the input tensors must be untouched and still owned by torch."""
import numpy as np
import pytest


def test_cpu_tensors_are_rejected_loudly():
    import torch
    import halide_amd.torch_ops  # noqa: F401
    with pytest.raises(RuntimeError, match="GPU"):
        torch.ops.hlmi.stencil_chain(torch.zeros((8, 8), dtype=torch.uint16))


@pytest.mark.gpu
def test_local_laplacian_op_matches_oracle_without_copies(oracle):
    import torch
    import halide_amd.torch_ops  # noqa: F401
    rng = np.random.default_rng(4)
    inp = rng.integers(0, 65536, (3, 150, 260), dtype=np.uint16)
    t = torch.from_numpy(inp).cuda()
    ptr = t.data_ptr()
    out = torch.ops.hlmi.local_laplacian(t, 8, 1.0 / 7.0, 1.0)
    torch.cuda.synchronize()
    assert out.is_cuda and out.dtype == torch.uint16 and t.data_ptr() == ptr
    assert np.array_equal(out.cpu().numpy(), oracle.local_laplacian(inp, 8, 1.0 / 7.0, 1.0))
    assert np.array_equal(t.cpu().numpy(), inp)
    # a second call on a side stream, right behind torch work on that stream
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        t2 = (t.to(torch.int32) // 2).to(torch.uint16)
        out2 = torch.ops.hlmi.local_laplacian(t2, 8, 1.0 / 7.0, 1.0)
    s.synchronize()
    assert np.array_equal(out2.cpu().numpy(), oracle.local_laplacian(inp // 2, 8, 1.0 / 7.0, 1.0))


@pytest.mark.gpu
def test_conv_and_stencil_ops(oracle):
    import torch
    import halide_amd.torch_ops  # noqa: F401
    rng = np.random.default_rng(5)
    inp = rng.uniform(-1, 1, (2, 9, 11, 64)).astype(np.float32)
    filt = rng.uniform(-1, 1, (64, 3, 3, 128)).astype(np.float32)
    bias = rng.uniform(-1, 1, 128).astype(np.float32)
    got = torch.ops.hlmi.conv_layer(torch.from_numpy(inp).cuda(), torch.from_numpy(filt).cuda(), torch.from_numpy(bias).cuda())
    assert np.array_equal(got.cpu().numpy(), oracle.conv_layer(inp, filt, bias))
    img = rng.integers(0, 65536, (70, 90), dtype=np.uint16)
    got = torch.ops.hlmi.stencil_chain(torch.from_numpy(img).cuda())
    assert np.array_equal(got.cpu().numpy(), oracle.stencil_chain(img))


@pytest.mark.gpu
def test_adjacent_app_ops(oracle):
    """harris (output origin (3, 3)), interpolate, iir_blur, lens_blur, bgu through torch.ops.hlmi: equal to the oracle."""
    import torch
    import halide_amd.torch_ops  # noqa: F401
    rng = np.random.default_rng(6)
    img = rng.random((3, 70, 90), dtype=np.float32)
    got = torch.ops.hlmi.harris(torch.from_numpy(img).cuda())
    assert np.array_equal(got.cpu().numpy(), oracle.harris(img))
    rgba = rng.random((4, 70, 90), dtype=np.float32)
    rgba[3] *= rng.random((70, 90)) > 0.6
    got = torch.ops.hlmi.interpolate(torch.from_numpy(rgba).cuda())
    assert np.array_equal(got.cpu().numpy(), oracle.interpolate(rgba))
    got = torch.ops.hlmi.iir_blur(torch.from_numpy(img).cuda(), 0.25)
    assert np.array_equal(got.cpu().numpy(), oracle.iir_blur(img, np.float32(0.25)))
    left = rng.integers(0, 256, (3, 40, 48), dtype=np.uint8)
    right = np.roll(left, 4, 2)
    got = torch.ops.hlmi.lens_blur(torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), 16, 5, 0.5, 12)
    assert np.array_equal(got.cpu().numpy(), oracle.lens_blur(left, right, 16, 5, 0.5, 12))
    hi = rng.random((3, 96, 128), dtype=np.float32)
    lo = hi[:, ::8, ::8].copy()
    val = (lo * lo * (3 - 2 * lo)).astype(np.float32)
    got = torch.ops.hlmi.bgu(0.125, 16, torch.from_numpy(lo).cuda(), torch.from_numpy(val).cuda(), torch.from_numpy(hi).cuda())
    assert np.array_equal(got.cpu().numpy(), oracle.bgu(0.125, 16, lo, val, hi))


@pytest.mark.gpu
def test_work_on_torchs_default_stream_is_ordered_before_other_streams(hl, oracle):
    """torch's default stream reaches the library as hipStreamLegacy.  An event RECORDED on that handle makes the next
    hipStreamWaitEvent on it crash inside the HIP runtime (ROCm 7.2, scripts/legacy_event_probe.py), so the library names
    the NULL stream in its event calls instead.  Two places order work across streams with events: the per-(levels, alpha)
    remap-table cache of local_laplacian and the last-writer tracking of library-owned allocations."""
    import torch
    import halide_amd.torch_ops  # noqa: F401
    rng = np.random.default_rng(8)
    inp = rng.integers(0, 65536, (3, 90, 140), dtype=np.uint16)
    t = torch.from_numpy(inp).cuda()
    levels, alpha = 5, 0.3125 / 4                      # a pair no other test uses: its table is made HERE, on the default stream
    out = torch.ops.hlmi.local_laplacian(t, levels, alpha, 1.0)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        out2 = torch.ops.hlmi.local_laplacian(t, levels, alpha, 1.0)     # waits for the table's event on another stream
    torch.cuda.synchronize()
    want = oracle.local_laplacian(inp, levels, np.float32(alpha), 1.0)
    assert np.array_equal(out.cpu().numpy(), want) and np.array_equal(out2.cpu().numpy(), want)
    # a library-owned buffer written on the default stream, read by a pipeline on another stream
    img = rng.integers(0, 65536, (70, 90), dtype=np.uint16)
    a, mid, o = hl.Buffer(img), hl.Buffer(np.zeros_like(img)), hl.Buffer(np.zeros_like(img))
    hl.set_stream(1)                                   # hipStreamLegacy
    hl.stencil_chain(a, mid)
    hl.set_stream(side.cuda_stream)
    hl.stencil_chain(mid, o)
    hl.set_stream(None)
    assert np.array_equal(o.numpy(), oracle.stencil_chain(oracle.stencil_chain(img)))
