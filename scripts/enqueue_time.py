import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import torch, halide_amd as hl, bench
fr = [bench.synth_frame(i) for i in range(4)]
ins = [hl.Buffer(f) for f in fr]; outs = [hl.Buffer(np.zeros_like(f)) for f in fr]
for a, o in zip(ins, outs): hl.local_laplacian(a, 8, 1/7, 1.0, o)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        for a, o in zip(ins, outs): hl.local_laplacian(a, 8, 1/7, 1.0, o)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("enqueue us/frame", (t1 - t0) / 40 * 1e6, "total us/frame", (t2 - t0) / 40 * 1e6)
