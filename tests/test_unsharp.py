"""unsharp: gray -> separable 7-tap Gaussian (sigma 1.5) -> sharpen -> ratio -> recolour, f32 planar
(reference: /root/reference/apps/unsharp/unsharp_generator.cpp:13-52).  GPU == oracle bit for bit."""
import math

import numpy as np
import pytest


def _img(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = 0.5 + 0.3 * np.sin(xx / 17.0) * np.cos(yy / 11.0)
    return np.clip(np.stack([base, np.roll(base, 5, 1) * 0.9, base[::-1] * 0.8]) + rng.normal(0, 0.05, (3, h, w)), 0.02, 1).astype(np.float32)


def test_kernel_taps_are_the_folded_constants(oracle):
    """The simplifier folds exp_f32(constant) with the host's DOUBLE exp, then rounds (src/Simplify_Call.cpp:767-780)."""
    k = oracle.unsharp_kernel()
    den = np.float32(np.sqrt(np.float32(2 * np.float32(3.14159265358979310000))) * np.float32(1.5))
    for i in range(4):
        arg = np.float32(-i * i) / np.float32(4.5)
        assert k[i] == np.float32(np.float32(math.exp(float(arg))) / den)
    assert abs(float(k[0] + 2 * (k[1] + k[2] + k[3])) - 1.0) < 0.03   # a (truncated) normalised Gaussian


def test_oracle_against_float64_reference(oracle):
    inp = _img(40, 30, 1)
    got = oracle.unsharp(inp)
    k = oracle.unsharp_kernel().astype(np.float64)
    gray = 0.299 * inp[0].astype(np.float64) + 0.587 * inp[1] + 0.114 * inp[2]
    pad = np.pad(gray, 3, mode="edge")
    by = sum(k[abs(d)] * pad[3 + d:3 + d + 30, :] for d in range(-3, 4))
    bx = sum(k[abs(d)] * by[:, 3 + d:3 + d + 40] for d in range(-3, 4))
    ref = (2 * gray - bx) / gray * inp.astype(np.float64)
    assert np.max(np.abs(got - ref)) < 1e-5



def test_oracle_matches_float32_numpy_restatement_bit_for_bit(oracle, canon0):
    """Second reading of the generator (:13-52), array at a time in float32, with the folded kernel constants."""
    f32 = np.float32
    inp = _img(45, 37, 4)
    h, w = inp.shape[1:]
    k = oracle.unsharp_kernel()
    pad = np.pad(inp, ((0, 0), (3, 3), (3, 3)), mode="edge")              # repeat_edge (:20)
    g = (f32(0.299) * pad[0] + f32(0.587) * pad[1]) + f32(0.114) * pad[2]
    gy = lambda d: g[3 + d:3 + d + h, :]                                   # gray(x, y + d) for every padded column
    by = ((k[0] * gy(0) + k[1] * (gy(-1) + gy(1))) + k[2] * (gy(-2) + gy(2))) + k[3] * (gy(-3) + gy(3))
    bx_ = lambda d: by[:, 3 + d:3 + d + w]
    bx = ((k[0] * bx_(0) + k[1] * (bx_(-1) + bx_(1))) + k[2] * (bx_(-2) + bx_(2))) + k[3] * (bx_(-3) + bx_(3))
    g0 = g[3:3 + h, 3:3 + w]
    ratio = (f32(2.0) * g0 - bx) / g0
    want = (ratio[None] * inp).astype(f32)
    got = oracle.unsharp(inp)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got != want)} differ"


def _run(hl, inp, out_min=None, out_size=None, in_min=None):
    a = hl.Buffer(inp)
    if in_min:
        a.set_min(*in_min, 0)
    ow, oh = out_size if out_size else (inp.shape[2], inp.shape[1])
    o = hl.Buffer(np.zeros((3, oh, ow), np.float32))
    if out_min:
        o.set_min(*out_min, 0)
    hl.unsharp(a, o)
    return o.numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1536, 2560), (1, 1), (7, 5), (64, 16), (65, 17), (333, 201)])
def test_hip_matches_oracle_bit_for_bit(hl, oracle, w, h):
    inp = _img(w, h, seed=w + h)
    got, want = _run(hl, inp), oracle.unsharp(inp)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got != want)} of {got.size} differ"


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(1536, 2560), (333, 201)])
def test_hip_64_bit_kernel_matches_oracle_bit_for_bit(hl, oracle, monkeypatch, w, h):
    """Rows of 2^29 floats and more take unsharp_tile (flat element numbers, long row products); HLMI_UNSHARP_REF=1 selects it."""
    monkeypatch.setenv("HLMI_UNSHARP_REF", "1")
    inp = _img(w, h, seed=w + h)
    got, want = _run(hl, inp), oracle.unsharp(inp)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{np.count_nonzero(got != want)} of {got.size} differ"


@pytest.mark.gpu
def test_hip_crop_with_nonzero_mins(hl, oracle):
    inp = _img(120, 90, seed=3)
    got = _run(hl, inp, out_min=(17, 9), out_size=(64, 40), in_min=(5, 2))
    want = oracle.unsharp(inp, out_origin=(17, 9), out_size=(64, 40), in_origin=(5, 2))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_hip_output_outside_the_input_is_out_of_bounds(hl):
    inp = _img(32, 32, seed=0)
    with pytest.raises(hl.HalideError) as e:
        _run(hl, inp, out_min=(8, 0), out_size=(32, 32))
    assert e.value.code == -4


def test_bounds_query(hl):
    q = hl.Buffer.bounds_query(np.float32, 3)
    o = hl.Buffer(np.zeros((3, 20, 30), np.float32)).set_min(4, 2, 0)
    hl.unsharp(q, o)
    assert q.mins == [4, 2, 0] and q.extents == [30, 20, 3]
