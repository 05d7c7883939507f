"""Is the headline loop host-bound?  Time to ENQUEUE a step's frames (no wait) against the step's wall time, with one host thread
(what bench.py does) and with one host thread per frame queue.  python scripts/host_enqueue_probe.py"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl
import bench

hip = hl.hip_runtime()
nframes, nq, passes = 8, 4, 16
fr = [bench.synth_frame(i) for i in range(nframes)]
ins = [hl.Buffer(f) for f in fr]
outs = [hl.Buffer(np.zeros_like(f)) for f in fr]
queues = [hl.partition_stream(p, nq) for p in range(nq)]


def one_thread():
    t0 = time.perf_counter()
    for _ in range(passes):
        for i, (a, o) in enumerate(zip(ins, outs)):
            hl.set_stream(queues[i % nq])
            hl.local_laplacian(a, 8, 1 / 7, 1.0, o)
    t1 = time.perf_counter()
    hl.set_stream(None)
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t0


def per_queue_threads():
    def work(q):
        hl.set_stream(queues[q])
        for _ in range(passes):
            for i in range(q, nframes, nq):
                hl.local_laplacian(ins[i], 8, 1 / 7, 1.0, outs[i])
        hl.set_stream(None)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(q,)) for q in range(nq)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    t1 = time.perf_counter()
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t0


n = passes * nframes
for name, fn in (("one host thread", one_thread), ("a host thread per queue", per_queue_threads)):
    fn()
    best = min((fn() for _ in range(5)), key=lambda x: x[1])
    print(f"{name}: enqueue {best[0] / n * 1e6:.1f} us per frame, step {best[1] / n * 1e6:.1f} us per frame", flush=True)
