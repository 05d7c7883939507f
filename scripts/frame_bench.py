"""Torch-free timing of the headline workload (bench.py's step without the torch plumbing): us per 4K frame on one stream
and on N frame-queue streams.  For quick A/B runs of kernel changes: python scripts/frame_bench.py [frames] [partitions]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl
import bench

hip = hl.hip_runtime()   # the runtime libhlmi.so is bound to
nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nparts = int(sys.argv[2]) if len(sys.argv) > 2 else 4
fr = [bench.synth_frame(i) for i in range(nframes)]
ins = [hl.Buffer(f) for f in fr]
outs = [hl.Buffer(np.zeros_like(f)) for f in fr]


def run(streams, label, reps=5, inner=5):
    for i, (a, o) in enumerate(zip(ins, outs)):
        hl.set_stream(streams[i % len(streams)] if streams else None)
        hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
    hip.hipDeviceSynchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(inner):
            for i, (a, o) in enumerate(zip(ins, outs)):
                if streams:
                    hl.set_stream(streams[i % len(streams)])
                hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
        hip.hipDeviceSynchronize()
        best = min(best, (time.perf_counter() - t0) / (inner * nframes))
    hl.set_stream(None)
    print(f"{label}: {best * 1e6:.1f} us/frame  {3840 * 2160 / best / 1e9:.2f} Gpx/s", flush=True)


for rep in range(2):
    run(None, "1 stream")
    run([hl.partition_stream(p, nparts) for p in range(nparts)], f"{nparts} partitions")
