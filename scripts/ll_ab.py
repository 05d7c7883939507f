"""A/B timing of local_laplacian launch-chain switches in ONE process (the library reads its HLMI_LL_* switches per call):
for every configuration, (a) one call + device sync, the reference's protocol (tools/halide_benchmark.h: min over samples),
(b) frames back to back on one stream, (c) frames spread over four CU-partitioned streams.  us per 3840x2160 frame.
    python scripts/ll_ab.py "HLMI_LL_FUSE_UP2=0" "HLMI_LL_FUSE_UP2=1" "HLMI_LL_NT=1,HLMI_LL_RU=16" ...
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl
import bench

hip = hl.hip_runtime()
kind = os.environ.get("LL_AB_KIND", "smooth")
fr = [bench.synth_frame(i, kind=kind) for i in range(8)]
ins = [hl.Buffer(f) for f in fr]
outs = [hl.Buffer(np.zeros_like(f)) for f in fr]
parts = [hl.partition_stream(p, 4) for p in range(4)]
configs = sys.argv[1:] or [""]
touched = set()


def apply(cfg):
    for k in touched:
        os.environ.pop(k, None)
    for kv in filter(None, cfg.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
        touched.add(k)


def throughput(streams, inner=6, reps=5):
    def once():
        for i, (a, o) in enumerate(zip(ins, outs)):
            if streams:
                hl.set_stream(streams[i % len(streams)])
            hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
    once()
    hip.hipDeviceSynchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(inner):
            once()
        hip.hipDeviceSynchronize()
        best = min(best, (time.perf_counter() - t0) / (inner * len(ins)))
    hl.set_stream(None)
    return best * 1e6


def latency(samples=40):
    a, o = ins[0], outs[0]
    best = 1e9
    for _ in range(samples):
        t0 = time.perf_counter()
        hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
        o.device_sync()
        best = min(best, time.perf_counter() - t0)
    return best * 1e6


for rnd in range(int(os.environ.get("LL_AB_ROUNDS", "2"))):
    for cfg in configs:
        apply(cfg)
        lat = latency()
        t1 = throughput(None)
        t4 = throughput(parts)
        print(f"[{rnd}] {cfg or 'default':60s} call+sync {lat:6.1f}  1-stream {t1:6.1f}  4-partitions {t4:6.1f}  us/frame ({kind})", flush=True)
apply("")
# per-launch HIP-event durations of the last configuration's chain on one stream
if os.environ.get("LL_AB_KERNELS", "1") == "1":
    for cfg in configs:
        apply(cfg)
        hl.kernel_timing_reset()
        hl.kernel_timing(True)
        for _ in range(3):
            for a, o in zip(ins, outs):
                hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
        hip.hipDeviceSynchronize()
        hl.kernel_timing(False)
        rep = hl.kernel_timing_report()
        print(cfg or "default", " ".join(f"{k['name']}={k['avg_ms'] * 1e3:.1f}" for k in rep), flush=True)
        hl.kernel_timing_reset()
