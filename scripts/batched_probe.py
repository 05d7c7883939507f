"""Frames-in-flight sweep for the small BASELINE configs: partitions x frames, plain streams (scratch measurement)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl

hip = hl.hip_runtime()
rng = np.random.default_rng(0)


def plain_streams(n):
    out = []
    for _ in range(n):
        s = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
        out.append(s.value)
    return out


def bench(calls, streams, rounds=6, samples=4):
    def one():
        for i, c in enumerate(calls):
            hl.set_stream(streams[i % len(streams)] if streams else None)
            c()
        hl.set_stream(None)
    one()
    hip.hipDeviceSynchronize()
    best = 1e9
    for _ in range(samples):
        t0 = time.perf_counter()
        for _ in range(rounds):
            one()
        hip.hipDeviceSynchronize()
        best = min(best, (time.perf_counter() - t0) / (rounds * len(calls)))
    return best * 1e3


W, H = 1920, 1080
img = rng.random((H, W), dtype=np.float32)
nlm = rng.random((3, H, W), dtype=np.float32)
N, Hh, Ww, CI, CO = 16, 56, 56, 128, 128
c_in = rng.uniform(-1, 1, (N, Hh + 2, Ww + 2, CI)).astype(np.float32)
filt, bias = hl.Buffer(rng.uniform(-1, 1, (CI, 3, 3, CO)).astype(np.float32)), hl.Buffer(rng.uniform(-1, 1, CO).astype(np.float32))


def mk(kind, n):
    calls = []
    for i in range(n):
        if kind == "bg":
            a, o = hl.Buffer(np.roll(img, i, 1).copy()), hl.Buffer(np.zeros((H, W), np.float32))
            calls.append(lambda a=a, o=o: hl.bilateral_grid(a, 0.1, o))
        elif kind == "nlm":
            a, o = hl.Buffer(np.roll(nlm, i, 2).copy()), hl.Buffer(np.zeros((3, H, W), np.float32))
            calls.append(lambda a=a, o=o: hl.nl_means(a, 7, 7, 0.12, o))
        else:
            a, o = hl.Buffer(np.roll(c_in, i, 1).copy()), hl.Buffer(np.zeros((N, Hh, Ww, CO), np.float32))
            calls.append(lambda a=a, o=o: hl.conv_layer_bf16(a, filt, bias, o))
    return calls


for kind in sys.argv[1:] or ["bg", "nlm", "conv"]:
    n = 16
    calls = mk(kind, n)
    print(kind, "1 stream", round(bench(calls, None, rounds=3 if kind == "nlm" else 6), 4), flush=True)
    for np_ in (2, 4, 8):
        st = [hl.partition_stream(p, np_) for p in range(np_)]
        print(kind, f"{np_} partitions", round(bench(calls, st, rounds=3 if kind == "nlm" else 6), 4), flush=True)
    for ns in (2, 4, 8):
        print(kind, f"{ns} plain streams", round(bench(calls, plain_streams(ns), rounds=3 if kind == "nlm" else 6), 4), flush=True)
