"""A seeded slice of scripts/fuzz_parity.py inside the driver's `-m gpu` suite: for every pipeline a fixed NUMBER of random
(shape, origin, parameter) cases — not a time budget, so the cases are the same on every box — GPU result vs the oracle, bit
for bit (conv_layer_bf16: its tolerance).  The full sweeps (minutes, 10^5 cases, 8 host threads) stay a script:
profiles/r02c_fuzz_parity.txt."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# cases per pipeline: sized so that the whole slice takes a few seconds of GPU + oracle time
N_CASES = {"local_laplacian": 6, "bilateral_grid": 6, "nl_means": 4, "stencil_chain": 6, "halide_blur": 8, "unsharp": 6,
           "harris": 6, "max_filter": 3, "hist": 6, "interpolate": 3, "iir_blur": 4, "bgu": 3, "lens_blur": 1,
           "camera_pipe": 4, "depthwise_separable_conv": 4, "conv_layer": 3, "conv_layer_bf16": 3}


@pytest.fixture(scope="module")
def fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "scripts", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)     # imports halide_amd (the product) and oracle_lib (the checker)
    return mod


def test_slice_covers_every_fuzzed_pipeline():
    # CPU-runnable: the table above must name exactly the cases the fuzzer defines (reads the script's source only)
    src = open(os.path.join(ROOT, "scripts", "fuzz_parity.py")).read()
    names = {ln[len("def case_"):ln.index("(")] for ln in src.splitlines() if ln.startswith("def case_")}
    assert names == set(N_CASES)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(N_CASES))
def test_seeded_fuzz_slice(fuzz, name):
    rng = np.random.default_rng(20260923 + sum(map(ord, name)))
    for i in range(N_CASES[name]):
        desc, ok = fuzz.CASES[name](rng)
        assert ok, f"{name} case {i}: {desc}"


@pytest.mark.gpu
def test_seeded_fuzz_slice_local_laplacian_on_a_cu_partition(fuzz, on_stream):
    """The headline pipeline's cases again with the calling thread on a frame-queue stream (and, for symmetry, on the device's
    own): partitions switch other defaults on inside local_laplacian (non-temporal frame accesses, level 2 collapsed inside
    ll_up0h, taller units), and that is what bench.py times."""
    rng = np.random.default_rng(20260924)
    for i in range(2 * N_CASES["local_laplacian"]):
        desc, ok = fuzz.CASES["local_laplacian"](rng)
        assert ok, f"local_laplacian case {i} on the {on_stream} stream: {desc}"
