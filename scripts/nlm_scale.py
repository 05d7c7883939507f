"""nl_means time vs number of workgroups (tile 58 x 64): separates the per-workgroup time from the tail of a launch."""
import os
import sys
import time
import ctypes as C

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl

hip = hl.hip_runtime()   # the runtime libhlmi.so is bound to
rng = np.random.default_rng(0)
for tx, ty in [(1, 1), (4, 4), (16, 8), (16, 16), (16, 17), (16, 24), (16, 32), (32, 24), (34, 17), (32, 32)]:
    w, h = 58 * tx, 64 * ty
    img = rng.random((3, h, w), dtype=np.float32)
    a, o = hl.Buffer(img), hl.Buffer(np.zeros_like(img))
    for _ in range(3):
        hl.nl_means(a, 7, 7, 0.12, o)
    hip.hipDeviceSynchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            hl.nl_means(a, 7, 7, 0.12, o)
        hip.hipDeviceSynchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    print(f"{tx * ty:5d} workgroups ({w}x{h}): {best * 1e6:8.1f} us   {best * 1e6 / max(1, -(-tx * ty // 256)):7.1f} us per round of 256", flush=True)
