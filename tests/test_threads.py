"""Re-entrancy: the reference's generated pipelines may be called concurrently from several host threads
(test/generator/gpu_multi_context_threaded_aottest.cpp, variable_num_threads_aottest.cpp).  Here calls that share a
(device, stream) share one scratch arena, so their launches must be enqueued one call at a time; threads with their own
stream (halide_hip_set_stream) overlap on the GPU.  Either way every result must be the oracle's, bit for bit."""
import threading

import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("own_streams", [False, True])
def test_concurrent_calls_from_host_threads(hl, oracle, own_streams):
    import torch
    rng = np.random.default_rng(7)
    frames = [rng.integers(0, 65536, (3, 200 + 16 * i, 320), dtype=np.uint16) for i in range(4)]
    gray = [rng.random((180, 250 + 8 * i), dtype=np.float32) for i in range(4)]
    want_ll = [oracle.local_laplacian(f, 8, 1.0 / 7.0, 1.0) for f in frames]
    want_bg = [oracle.bilateral_grid(g, 0.1) for g in gray]
    errors = []

    def worker(i):
        try:
            stream = torch.cuda.Stream() if own_streams else None
            if stream is not None:
                hl.set_stream(stream.cuda_stream)
            for rep in range(6):
                a, o = hl.Buffer(frames[i]), hl.Buffer(np.zeros_like(frames[i]))
                hl.local_laplacian(a, 8, 1.0 / 7.0, 1.0, o)
                b, p = hl.Buffer(gray[i]), hl.Buffer(np.zeros_like(gray[i]))
                hl.bilateral_grid(b, 0.1, p)
                if not np.array_equal(o.numpy(), want_ll[i]):
                    errors.append(f"thread {i} rep {rep}: local_laplacian differs")
                if not np.array_equal(p.numpy().view(np.uint32), want_bg[i].view(np.uint32)):
                    errors.append(f"thread {i} rep {rep}: bilateral_grid differs")
            if stream is not None:
                hl.set_stream(None)
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {i}: {e!r}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.gpu
def test_partition_streams_are_distinct_and_compute_correctly(hl, oracle):
    """halide_hip_partition_stream: library-owned streams confined to disjoint CU partitions (bench.py spreads the
    frames of a step over four of them).  Results on a partition must be the oracle's; handles are cached."""
    streams = [hl.partition_stream(p, 4) for p in range(4)]
    assert all(streams) and len(set(streams)) == 4
    assert hl.partition_stream(2, 4) == streams[2]
    assert hl.partition_stream(4, 4) is None and hl.partition_stream(0, 0) is None
    rng = np.random.default_rng(1)
    frames = [rng.integers(0, 65536, (3, 180, 256), dtype=np.uint16) for _ in range(4)]
    outs = []
    for f, s in zip(frames, streams):
        hl.set_stream(s)
        a, o = hl.Buffer(f), hl.Buffer(np.zeros_like(f))
        hl.local_laplacian(a, 8, 1.0 / 7.0, 1.0, o)
        outs.append((a, o))
    hl.set_stream(None)
    for f, (a, o) in zip(frames, outs):
        assert np.array_equal(o.numpy(), oracle.local_laplacian(f, 8, 1.0 / 7.0, 1.0))
