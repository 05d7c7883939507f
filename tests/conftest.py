import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hl():
    """The product library (fails loudly if libhlmi.so was not built)."""
    import halide_amd
    return halide_amd


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib


@pytest.fixture(params=["device", "partition"])
def on_stream(request, hl):
    """Runs the test twice: on the library's own device-wide stream and on a CU-partitioned stream
    (halide_hip_partition_stream: partition 1 of 4, 64 CUs) — bench.py times the headline pipeline on partitions, where
    local_laplacian switches its non-temporal frame accesses, the level-2 collapse inside ll_up0h and its launch geometry on by
    default, so those code paths must face the oracle too, not only the device-wide ones."""
    if request.param == "partition":
        s = hl.partition_stream(1, 4)
        assert s, "the device refused a CU-masked stream"
        hl.set_stream(s)
    yield request.param
    hl.set_stream(None)
