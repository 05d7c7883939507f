"""Per-launch durations of one 4K local_laplacian frame on ONE CU-partitioned stream (64 CUs), the others idle: what each launch
costs a partition when it has the memory system to itself."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl
import bench

nparts = int(sys.argv[1]) if len(sys.argv) > 1 else 4
f = bench.synth_frame(1)
a, o = hl.Buffer(f), hl.Buffer(np.zeros_like(f))
hl.set_stream(hl.partition_stream(0, nparts))
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
    print(f"{k['name']:20s} {k['avg_ms'] * 1e3:8.1f} us")
    tot += k["total_ms"] / 10
print(f"sum {tot * 1e3:.1f} us on one of {nparts} partitions")
hl.set_stream(None)
