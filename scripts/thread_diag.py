"""Per-operation wall times of tests/test_threads.py's shared-stream case (4 host threads, one stream)."""
import sys, os, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl

def main(nthreads=4, reps=3):
    rng = np.random.default_rng(7)
    frames = [rng.integers(0, 65536, (3, 200 + 16 * i, 320), dtype=np.uint16) for i in range(nthreads)]
    gray = [rng.random((180, 250 + 8 * i), dtype=np.float32) for i in range(nthreads)]
    log = []
    def worker(i):
        for rep in range(reps):
            t = [time.perf_counter()]
            a, o = hl.Buffer(frames[i]), hl.Buffer(np.zeros_like(frames[i])); t.append(time.perf_counter())
            hl.local_laplacian(a, 8, 1.0 / 7.0, 1.0, o); t.append(time.perf_counter())
            b, p = hl.Buffer(gray[i]), hl.Buffer(np.zeros_like(gray[i])); t.append(time.perf_counter())
            hl.bilateral_grid(b, 0.1, p); t.append(time.perf_counter())
            o.numpy(); t.append(time.perf_counter())
            p.numpy(); t.append(time.perf_counter())
            del a, o, b, p; t.append(time.perf_counter())
            log.append((i, rep, [round((t[k + 1] - t[k]) * 1e3, 2) for k in range(len(t) - 1)]))
    t0 = time.perf_counter()
    ths = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
    [t.start() for t in ths]; [t.join() for t in ths]
    print(f"threads={nthreads} total {time.perf_counter() - t0:.2f} s; per op ms [mk, ll, mk, bg, ll.numpy, bg.numpy, del]")
    for e in sorted(log): print(e)

if __name__ == "__main__":
    main(1, 2)
    main(4, 3)
