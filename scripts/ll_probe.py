"""Phase stamps of ll_down01f / ll_up0f (libhlmi_probe.so, csrc/Makefile VARIANT=_probe EXTRA=-DHLMI_LL_PROBE=1):
HLMI_LIB=halide_amd/lib/libhlmi_probe.so python scripts/ll_probe.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl
import bench

f = bench.synth_frame(1)
a, o = hl.Buffer(f), hl.Buffer(np.zeros_like(f))
buf = (C.c_ulonglong * 32)()
for _ in range(3):
    hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
assert hl.lib.hlmi_debug_ll_probe(buf) == 1, "not a probe build"
N = 5
for _ in range(N):
    hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
hl.lib.hlmi_debug_ll_probe(buf)
v = [int(x) for x in buf]
# s_memtime ticks at a constant 100 MHz on gfx9 (10 ns)
tick = 10.0  # wall_clock64
nw = v[1] / N
if v[1]:
  print(f"down01: waves {nw:.0f}; per wave [ns]: lut fill {v[0] / v[1] * tick:.0f}, prologue {v[2] / v[1] * tick:.0f}, first pair+barrier {v[3] / max(1, v[1]) * tick:.0f}, "
      f"loop {v[4] / v[1] * tick:.0f} for {v[5] / v[1]:.1f} rows, total {v[6] / v[1] * tick:.0f}")
if v[1]:
  print(f"down01 prologue: setup before loads {v[17] / v[1] * tick:.0f}, first four rows' loads outstanding {v[16] / v[1] * tick:.0f}")
if v[10]:
  print(f"up0: waves {v[10] / N:.0f}; per wave [ns]: tile phase {v[8] / v[10] * tick:.0f}, barrier wait {v[9] / v[10] * tick:.0f}, rows {v[11] / v[10] * tick:.0f}, total {v[12] / v[10] * tick:.0f}")
if v[24]:
    w = v[24]
    print(f"down01e: waves {w / N:.0f}; per wave [ns]: steps {v[23] / w:.1f}; plane loop {v[20] / w * tick:.0f}, sel + emission {v[21] / w * tick:.0f}, "
          f"prep + loads {v[22] / w * tick:.0f}; prologue + walk {v[25] / w * tick:.0f}, tail {v[26] / w * tick:.0f}")
    print(f"  per step [ns]: planes {v[20] / v[23] * tick:.0f}, sel + emission {v[21] / v[23] * tick:.0f}, prep + loads {v[22] / v[23] * tick:.0f}")
    if v[28]:
        print(f"  shader clock during the walk: {v[27] / v[28] / 10.0:.0f} MHz (s_memtime cycles per 10 ns s_memrealtime tick), {v[27] / w:.0f} cycles per wave")
