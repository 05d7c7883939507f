"""The pool of freed device allocations (the counterpart of halide_reuse_device_allocations, src/runtime/cuda.cpp) is
bounded: HLMI_ALLOC_CACHE_MB per device, default 16 GiB.  The reference's pool is not; a caller that never repeats a size
would make an unbounded one hoard the device."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %r)
import halide_amd as hl
hip = hl.hip_runtime()
def free_bytes():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value
img = np.zeros((64, 64), np.uint16)
a, o = hl.Buffer(img), hl.Buffer(np.zeros((62, 62), np.uint16))
hl.halide_blur(a, o)                       # context, arena, caches: everything one-off is paid before the measurement
o.copy_to_host()
before = free_bytes()
for i in range(150):                        # 150 distinct sizes of ~8 MB: 1.2 GB if every freed block were kept
    b = hl.Buffer(np.zeros((1000 + i, 2048), np.float32))
    b.copy_to_device()                      # halide_copy_to_device allocates (src/runtime/device_interface.cpp:141-177)
    b.device_free()
held = before - free_bytes()
print("HELD_MB", held >> 20)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("limit_mb,most_mb", [(64, 200), (16384, 2000)])
def test_freed_allocations_kept_for_reuse_are_bounded(limit_mb, most_mb):
    env = dict(os.environ, HLMI_ALLOC_CACHE_MB=str(limit_mb))
    r = subprocess.run([sys.executable, "-c", CODE % ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    held = int([l for l in r.stdout.splitlines() if l.startswith("HELD_MB")][0].split()[1])
    assert held <= most_mb, f"{held} MB kept with a limit of {limit_mb} MB"
    if limit_mb >= 16384:
        assert held >= 1000, f"only {held} MB kept: the pool is not reusing anything"
