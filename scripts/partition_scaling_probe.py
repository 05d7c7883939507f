"""How does the time a CU partition needs for a frame depend on how many of the other partitions are busy?  Frames are dealt to the
first k of 4 partitions; per-partition frame latency = k x (time per frame)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl
import bench

hip = hl.hip_runtime()
nparts = 4
fr = [bench.synth_frame(i) for i in range(8)]
ins = [hl.Buffer(f) for f in fr]
outs = [hl.Buffer(np.zeros_like(f)) for f in fr]
streams = [hl.partition_stream(p, nparts) for p in range(nparts)]
for k in (1, 2, 3, 4):
    def one():
        for i, (a, o) in enumerate(zip(ins, outs)):
            hl.set_stream(streams[i % k])
            hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
    one()
    hip.hipDeviceSynchronize()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        for _ in range(4):
            one()
        hip.hipDeviceSynchronize()
        best = min(best, (time.perf_counter() - t0) / 32)
    hl.set_stream(None)
    print(f"{k} of 4 partitions busy: {best * 1e6:.1f} us per frame = {best * 1e6 * k:.1f} us per frame and partition", flush=True)
