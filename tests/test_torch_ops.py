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
