"""Phase stamps of iir_cols_T (probe build): HLMI_LIB=halide_amd/lib/libhlmi_iprobe.so python scripts/iir_probe.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl
rng = np.random.default_rng(0)
a = hl.Buffer(rng.random((3, 2560, 1536), dtype=np.float32)); o = hl.Buffer(np.zeros((3, 2560, 1536), np.float32))
buf = (C.c_ulonglong * 32)()
for _ in range(2): hl.iir_blur(a, 0.1, o)
assert hl.lib.hlmi_debug_iir_probe(buf) == 1
hl.iir_blur(a, 0.1, o); hl.lib.hlmi_debug_iir_probe(buf)
v = [int(x) for x in buf]
print(f"scanner (per pass, ticks): chain {v[0] / v[2]:.0f}, barrier wait {v[1] / v[2]:.0f}")
print(f"loader: fill (incl. wait for its loads) {v[4] / v[7]:.0f}, issue {v[5] / v[7]:.0f}, barrier wait {v[6] / v[7]:.0f}")
print(f"storer: drain {v[8] / v[10]:.0f}, barrier wait {v[9] / v[10]:.0f}")
for name, o in (("forward", 16), ("backward", 21)):
    n = max(v[o + 4], 1)
    print(f"scanner timeline, {name} passes (ticks of 10 ns): wait for tile 0 {v[o] / n:.0f}, first tile {v[o + 1] / n:.0f}, "
          f"steady loop {v[o + 2] / n:.0f}, last tile + drain {v[o + 3] / n:.0f}")
if v[29]:
    print(f"loader at the turn (ticks): 3 issues {v[26] / v[29]:.0f}, reversal incl. its barrier {v[27] / v[29]:.0f}, barrier #1 {v[28] / v[29]:.0f}")
