"""Torch-free per-launch durations of one 4K local_laplacian frame (HIP events around every launch, one stream)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl
import bench

f = bench.synth_frame(1)
a, o = hl.Buffer(f), hl.Buffer(np.zeros_like(f))
for _ in range(3):
    hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
o.device_sync()
hl.kernel_timing_reset()
hl.kernel_timing(True)
for _ in range(10):
    hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
o.device_sync()
hl.kernel_timing(False)
tot = 0.0
for k in hl.kernel_timing_report():
    print(f"{k['name']:20s} {k['avg_ms'] * 1e3:8.1f} us  x{k['calls'] // 10}")
    tot += k["total_ms"] / 10
print(f"sum {tot * 1e3:.1f} us")
