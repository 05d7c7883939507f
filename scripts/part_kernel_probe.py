"""What limits the headline loop: one kernel of the chain at a time (hlmi_kernel_timing_only), enqueued back to back on 1, 2, 3, 4 of the
four 64-CU partition streams (and on the device-wide stream), us per launch and the implied whole-device rate.  If a kernel scaled with
the CUs it is given, k busy partitions would finish k launches in the time one takes alone.
    python scripts/part_kernel_probe.py [kind]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl
import bench

kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
hip = hl.hip_runtime()
fr = [bench.synth_frame(i, kind=kind) for i in range(8)]
ins = [hl.Buffer(f) for f in fr]
outs = [hl.Buffer(np.zeros_like(f)) for f in fr]
parts = [hl.partition_stream(p, 4) for p in range(4)]


def warm(streams):
    for i, (a, o) in enumerate(zip(ins, outs)):
        hl.set_stream(streams[i % len(streams)] if streams else None)
        hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
    hl.set_stream(None)
    hip.hipDeviceSynchronize()


def rate(streams, reps=6):
    best = 1e9
    for _ in range(3):
        hip.hipDeviceSynchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for i, (a, o) in enumerate(zip(ins, outs)):
                if streams:
                    hl.set_stream(streams[i % len(streams)])
                hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
        hip.hipDeviceSynchronize()
        best = min(best, (time.perf_counter() - t0) / (reps * len(ins)))
    hl.set_stream(None)
    return best * 1e6


for only in (None, "ll_down01", "ll_up0", "ll_down_strip2:2"):
    row = []
    for label, streams in (("device", None), ("1 part", parts[:1]), ("2 parts", parts[:2]), ("3 parts", parts[:3]), ("4 parts", parts)):
        hl.kernel_timing_only(None)
        warm(streams)                     # complete frames in every stream's workspace first
        hl.kernel_timing_only(only)
        row.append(f"{label} {rate(streams):7.1f}")
    hl.kernel_timing_only(None)
    print(f"{only or 'whole chain':18s} us per launch (all streams together): " + " | ".join(row), flush=True)
