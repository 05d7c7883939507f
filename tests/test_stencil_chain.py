"""stencil_chain: 32 chained 5x5 stencils in wrapping u16 (reference:
/root/reference/apps/stencil_chain/stencil_chain_generator.cpp:18-34).  Integer ring arithmetic: the
oracle is exact by construction; the GPU must match bit-for-bit."""
import numpy as np
import pytest


def naive_stage(prev):
    """One stage on the interior of `prev` (shrinks by 2 on each side), straight from the generator text."""
    p = prev.astype(np.uint32)
    h, w = p.shape
    e = np.zeros((h - 4, w - 4), np.uint32)
    for i in range(-2, 3):
        for j in range(-2, 3):
            e = (e + ((i + 3) * (j + 3)) * p[2 + j:h - 2 + j, 2 + i:w - 2 + i]) & 0xFFFF
    return e.astype(np.uint16)


@pytest.mark.parametrize("w,h,stencils", [(7, 5, 3), (40, 33, 32), (1, 1, 32)])
def test_oracle_matches_numpy_restatement(oracle, w, h, stencils):
    rng = np.random.default_rng(w + h)
    inp = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    g = 2 * stencils
    cur = np.pad(inp, g, mode="edge")  # repeat_edge on stage 0 only
    for _ in range(stencils):
        cur = naive_stage(cur)
    assert cur.shape == inp.shape
    assert np.array_equal(oracle.stencil_chain(inp, stencils), cur)


def test_weights_sum_and_linearity(oracle):
    # constant image c -> every stage multiplies by 225 mod 2^16 (sum of weights)
    inp = np.full((20, 30), 7, np.uint16)
    assert np.all(oracle.stencil_chain(inp) == (7 * pow(225, 32, 65536)) % 65536)
    # linearity over the ring: f(a + b) = f(a) + f(b)
    rng = np.random.default_rng(2)
    a = rng.integers(0, 65536, (50, 60), dtype=np.uint16)
    b = rng.integers(0, 65536, (50, 60), dtype=np.uint16)
    lhs = oracle.stencil_chain((a + b).astype(np.uint16))
    rhs = (oracle.stencil_chain(a).astype(np.uint32) + oracle.stencil_chain(b)).astype(np.uint16)
    assert np.array_equal(lhs, rhs)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1536, 2560), (1, 1), (3, 2), (63, 65), (64, 64), (200, 130), (1000, 37)])
def test_hip_matches_oracle(hl, oracle, w, h):
    rng = np.random.default_rng(w * 3 + h)
    inp = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
    hl.stencil_chain(a, o)
    got, want = o.numpy(), oracle.stencil_chain(inp)
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} differ"


@pytest.mark.gpu
def test_hip_full_size_properties(hl, oracle):
    """4K: linearity over Z/2^16 and the constant-image closed form (size-independent properties)."""
    w, h = 3840, 2160
    rng = np.random.default_rng(0)
    a = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    b = rng.integers(0, 65536, (h, w), dtype=np.uint16)

    def run(x):
        bi, bo = hl.Buffer(x), hl.Buffer(np.zeros_like(x))
        hl.stencil_chain(bi, bo)
        return bo.numpy()
    fa, fb, fab = run(a), run(b), run((a + b).astype(np.uint16))
    assert np.array_equal(fab, (fa.astype(np.uint32) + fb).astype(np.uint16))
    assert np.all(run(np.full((h, w), 3, np.uint16)) == (3 * pow(225, 32, 65536)) % 65536)
    assert np.array_equal(fa, oracle.stencil_chain(a))


@pytest.mark.gpu
@pytest.mark.parametrize("x0,y0,w,h", [(30, 20, 50, 40), (31, 21, 51, 41), (7, 0, 110, 90), (0, 3, 64, 31)])
def test_hip_output_window_inside_larger_input(hl, oracle, x0, y0, w, h):
    """Even and odd origins / widths: the dword and the element-wise window moves of the kernel."""
    rng = np.random.default_rng(5)
    inp = rng.integers(0, 65536, (90, 120), dtype=np.uint16)
    full = oracle.stencil_chain(inp)
    a = hl.Buffer(inp)
    o = hl.Buffer(np.zeros((h, w), np.uint16)).set_min(x0, y0)
    hl.stencil_chain(a, o)
    assert np.array_equal(o.numpy(), full[y0:y0 + h, x0:x0 + w])
