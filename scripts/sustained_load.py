"""Keeps the headline loop (4K local_laplacian frames over k of 4 CU partitions) running for `seconds`: a load for clock / power sampling."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl
import bench

hip = hl.hip_runtime()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
fr = [bench.synth_frame(i) for i in range(8)]
ins = [hl.Buffer(f) for f in fr]
outs = [hl.Buffer(np.zeros_like(f)) for f in fr]
streams = [hl.partition_stream(p, 4) for p in range(4)]
print("load starts", flush=True)
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < seconds:
    for _ in range(50):
        for i, (a, o) in enumerate(zip(ins, outs)):
            hl.set_stream(streams[i % k])
            hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
    hip.hipDeviceSynchronize()
    n += 400
dt = time.perf_counter() - t0
print(f"{k} partitions: {n} frames in {dt:.2f} s = {dt / n * 1e6:.1f} us per frame", flush=True)
