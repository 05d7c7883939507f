"""bench.py's measurement helpers: the clock / power sampler must degrade to None fields where the amdgpu hwmon files are absent (this
container), and hlmi_kernel_timing_only must select exactly one launch of a chain and restore normal operation afterwards."""
import time

import numpy as np
import pytest


def test_clock_sampler_without_hwmon_files_reports_none():
    import bench
    with bench.ClockSampler(0) as cs:
        time.sleep(0.05)
    s = cs.summary()
    assert set(s) == {"sclk_mhz_median", "sclk_mhz_min", "sclk_mhz_max", "power_w_median", "power_w_max", "samples"}
    assert s["samples"] >= 1
    if cs.freq is None:
        assert s["sclk_mhz_median"] is None and s["sclk_mhz_min"] is None
    if cs.power is None:
        assert s["power_w_median"] is None


def test_headline_input_is_the_protocol_input():
    import bench
    assert bench.HEADLINE_KIND == "noise"
    f = bench.synth_frame(3, 64, 48, kind=bench.HEADLINE_KIND)
    assert f.dtype == np.uint16 and f.shape == (3, 48, 64)
    assert np.array_equal(f, bench.synth_frame(3, 64, 48, kind="noise"))          # seeded
    assert f.min() < 2000 and f.max() > 63000                                     # full range


@pytest.mark.gpu
def test_kernel_timing_only_selects_one_launch_and_restores(hl, oracle):
    rng = np.random.default_rng(5)
    inp = rng.integers(0, 65536, (3, 256, 512), dtype=np.uint16)
    a, o = hl.Buffer(inp), hl.Buffer(np.zeros_like(inp))
    hl.local_laplacian(a, 8, 1.0 / 7.0, 1.0, o)
    want = o.numpy().copy()
    hl.kernel_timing_reset()
    hl.kernel_timing(True)
    hl.local_laplacian(a, 8, 1.0 / 7.0, 1.0, o)
    o.device_sync()
    names = [k["name"] for k in hl.kernel_timing_report()]
    assert "ll_down01" in names and len(names) >= 3
    hl.kernel_timing_reset()
    hl.kernel_timing_only("ll_down01")
    try:
        for _ in range(3):
            hl.local_laplacian(a, 8, 1.0 / 7.0, 1.0, o)
        o.device_sync()
        rep = hl.kernel_timing_report()
        assert [k["name"] for k in rep] == ["ll_down01"] and rep[0]["calls"] == 3
    finally:
        hl.kernel_timing_only(None)
        hl.kernel_timing(False)
        hl.kernel_timing_reset()
    hl.local_laplacian(a, 8, 1.0 / 7.0, 1.0, o)       # normal operation again: the complete chain, the right answer
    assert np.array_equal(o.numpy(), want)
    assert np.array_equal(want, oracle.local_laplacian(inp, 8, 1.0 / 7.0, 1.0))
