"""Which kernels' workgroups share a compute unit while four frames are in flight?  (libhlmi_res.so:
`make -C halide_amd/csrc VARIANT=_res EXTRA=-DHLMI_LL_RESIDENCY=1`)

    HLMI_LIB=halide_amd/lib/libhlmi_res.so python scripts/residency_probe.py

Every ll_down01e / ll_up0h workgroup records, when it starts, how many workgroups of the other kernel and of its own were resident
on its CU.  Run with the defaults (one ll_down01e workgroup per CU on frame queues) and with HLMI_LL_D01_PAD_LDS=0 (two)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl
import bench

hip = hl.hip_runtime()
nframes, nq = 8, 4
fr = [bench.synth_frame(i) for i in range(nframes)]
ins = [hl.Buffer(f) for f in fr]
outs = [hl.Buffer(np.zeros_like(f)) for f in fr]
queues = [hl.partition_stream(p, nq) for p in range(nq)]
buf = (C.c_ulonglong * 16)()


def loop(passes):
    t0 = time.perf_counter()
    for _ in range(passes):
        for i, (a, o) in enumerate(zip(ins, outs)):
            hl.set_stream(queues[i % nq])
            hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
    hl.set_stream(None)
    hip.hipDeviceSynchronize()
    return (time.perf_counter() - t0) / (passes * nframes)


loop(4)
assert hl.lib.hlmi_debug_ll_residency(buf) == 1, "not a residency-probe build (HLMI_LIB=halide_amd/lib/libhlmi_res.so)"
t = loop(40)
hl.lib.hlmi_debug_ll_residency(buf)
v = [int(x) for x in buf]
print(f"{t * 1e6:.1f} us per frame with the probe's atomics; HLMI_LL_D01_PAD_LDS={os.environ.get('HLMI_LL_D01_PAD_LDS', '(default)')} "
      f"HLMI_LL_UNITS0={os.environ.get('HLMI_LL_UNITS0', '(default)')} HLMI_LL_RU={os.environ.get('HLMI_LL_RU', '(default)')}")
for kind, name, other in ((0, "ll_down01e", "ll_up0h"), (1, "ll_up0h", "ll_down01e")):
    a, b = v[8 * kind:8 * kind + 4], v[8 * kind + 4:8 * kind + 8]
    n = max(1, sum(a))
    print(f"{name}: {sum(a)} workgroups started; beside 0 / 1 / 2 / 3+ resident {other} workgroups: "
          + " / ".join(f"{x / n:.3f}" for x in a) + f"; beside 0 / 1 / 2 / 3+ of its own kind: " + " / ".join(f"{x / n:.3f}" for x in b))
