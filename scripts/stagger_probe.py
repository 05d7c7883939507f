"""Do the four CU partitions run in lock step (all in the down kernel, then all in the up kernel)?  Frame time with the
partitions started together vs. desynchronised by a smaller frame on every second partition before the timed loop."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl
import bench

hip = hl.hip_runtime()
nframes, nparts = 8, 4
fr = [bench.synth_frame(i) for i in range(nframes)]
ins = [hl.Buffer(f) for f in fr]
outs = [hl.Buffer(np.zeros_like(f)) for f in fr]
small = [bench.synth_frame(50 + i, 3840, 2160 // 2) for i in range(nparts)]
sin = [hl.Buffer(f) for f in small]
sout = [hl.Buffer(np.zeros_like(f)) for f in small]
streams = [hl.partition_stream(p, nparts) for p in range(nparts)]


def run(stagger, inner=6):
    for i, (a, o) in enumerate(zip(ins, outs)):
        hl.set_stream(streams[i % nparts])
        hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
    for i in range(nparts):
        hl.set_stream(streams[i])
        hl.local_laplacian(sin[i], 8, 1 / 7, 1.0, sout[i])
    hip.hipDeviceSynchronize()
    best = 1e9
    for _ in range(5):
        if stagger:
            for i in stagger:
                hl.set_stream(streams[i])
                hl.local_laplacian(sin[i], 8, 1 / 7, 1.0, sout[i])
        t0 = time.perf_counter()
        for _ in range(inner):
            for i, (a, o) in enumerate(zip(ins, outs)):
                hl.set_stream(streams[i % nparts])
                hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
        hip.hipDeviceSynchronize()
        best = min(best, (time.perf_counter() - t0) / (inner * nframes))
    hl.set_stream(None)
    return best * 1e6


for rep in range(3):
    print(f"in step: {run(None):.1f} us/frame   staggered (1,3): {run((1, 3)):.1f}   staggered (1,2,3 by 1/2/3 halves): {run((1, 2, 2, 3, 3, 3)):.1f}", flush=True)
