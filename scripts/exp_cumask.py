"""Experiment: frames spread over CU-masked streams (one partition of the chip per stream)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halide_amd as hl
import bench

hip = hl.hip_runtime()   # the runtime libhlmi.so is bound to


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    r = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert r == 0, r
    return s.value


def run(nframes, streams, label):
    fr = [bench.synth_frame(i % 4) for i in range(nframes)]
    ins = [hl.Buffer(f) for f in fr]
    outs = [hl.Buffer(np.zeros_like(f)) for f in fr]
    for i, (a, o) in enumerate(zip(ins, outs)):
        hl.set_stream(streams[i % len(streams)] if streams else None)
        hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(5):
            for i, (a, o) in enumerate(zip(ins, outs)):
                hl.set_stream(streams[i % len(streams)] if streams else None)
                hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (5 * nframes))
    hl.set_stream(None)
    print(f"{label}: {best * 1e6:.1f} us/frame  {3840 * 2160 / best / 1e9:.1f} Gpx/s", flush=True)


torch.cuda.init()
ilv = lambda n: [masked_stream(range(p, 256, n)) for p in range(n)]
for rep in range(2):
    run(8, [torch.cuda.Stream().cuda_stream for _ in range(2)], "2 plain streams, 8 frames")
    run(8, ilv(4), "4 interleaved partitions, 8 frames")
    run(16, ilv(4), "4 interleaved partitions, 16 frames")
    run(16, ilv(4) + ilv(4), "4 interleaved partitions x 2 streams, 16 frames")
    run(12, ilv(3)[:3], "3 interleaved partitions (stride 3), 12 frames")
    run(16, [masked_stream([b for b in range(256) if (b % 8) // 2 == p]) for p in range(4)], "4 partitions = XCD pairs {2p,2p+1} if bit%8 is the XCD, 16 frames")
