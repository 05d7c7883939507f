"""One warm 4K local_laplacian frame x N on one stream (for rocprofv3 counter passes): python scripts/ll_once.py [N]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl
import bench

f = bench.synth_frame(1, kind=os.environ.get("LL_KIND", bench.HEADLINE_KIND))
a, o = hl.Buffer(f), hl.Buffer(np.zeros_like(f))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
o.device_sync()
