"""Re-entrancy: the reference's generated pipelines may be called concurrently from several host threads
(test/generator/gpu_multi_context_threaded_aottest.cpp, variable_num_threads_aottest.cpp).  Here calls that share a
(device, stream) share one scratch arena, so their launches must be enqueued one call at a time; threads with their own
stream (halide_hip_set_stream) overlap on the GPU.  Either way every result must be the oracle's, bit for bit."""
import threading

import numpy as np
import pytest


@pytest.mark.gpu
@pytest.mark.timeout(180)
@pytest.mark.parametrize("own_streams", [False, True])
def test_concurrent_calls_from_host_threads(hl, oracle, own_streams):
    # NB: round 1's "224 s" of this test on the shared stream was the first `import torch` of the session on a fresh box
    # (the image pages in for minutes) plus the oracle, not the library: scripts/thread_diag.py times the same 4-thread
    # loop at 20 ms.  The caller-owned streams are therefore made with the HIP runtime directly, not through torch.
    import ctypes
    hip = hl.hip_runtime()   # the runtime the library is bound to (not dlopen by name: torch bundles another copy)
    rng = np.random.default_rng(7)
    frames = [rng.integers(0, 65536, (3, 200 + 16 * i, 320), dtype=np.uint16) for i in range(4)]
    gray = [rng.random((180, 250 + 8 * i), dtype=np.float32) for i in range(4)]
    want_ll = [oracle.local_laplacian(f, 8, 1.0 / 7.0, 1.0) for f in frames]
    want_bg = [oracle.bilateral_grid(g, 0.1) for g in gray]
    errors = []

    def worker(i):
        try:
            stream = None
            if own_streams:
                stream = ctypes.c_void_p()
                assert hip.hipStreamCreateWithFlags(ctypes.byref(stream), 1) == 0  # hipStreamNonBlocking
                hl.set_stream(stream.value)
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
                assert hip.hipStreamSynchronize(stream) == 0  # kept alive: the runtime's caches may still name it
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {i}: {e!r}")

    import time
    t0 = time.perf_counter()
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    elapsed = time.perf_counter() - t0
    assert not errors, errors
    # 48 small calls: re-entrancy that "works at 0.1 calls/s" is not re-entrancy (round 1: 224 s on the shared stream)
    assert elapsed < 20.0, f"{elapsed:.1f} s for 48 small calls from 4 threads"


@pytest.mark.gpu
def test_partition_streams_are_distinct_and_compute_correctly(hl, oracle):
    """halide_hip_partition_stream: library-owned frame-queue streams (bench.py spreads the
    frames of a step over four of them).  Results on such a stream must be the oracle's; handles are cached."""
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
    # read back with the thread's stream reset: copy_to_host orders itself behind the stream that PRODUCED each output
    for f, (a, o) in zip(frames, outs):
        assert np.array_equal(o.numpy(), oracle.local_laplacian(f, 8, 1.0 / 7.0, 1.0))


@pytest.mark.gpu
@pytest.mark.parametrize("layout,nparts", [(0, 5), (1, 6), (2, 7), (3, 9)])
def test_every_cu_mask_layout_computes_the_oracles_frame(hl, oracle, monkeypatch, layout, nparts):
    """HLMI_PART_MASK: 3 = the full mask (default: a queue of its own on the whole device), 1 / 2 = real CU partitions (the same
    CU slots on every XCD / contiguous slots), 0 = rounds 3-5's every-nparts-th bit.  The layout is read when a
    (part, nparts) stream is first made, so each case takes an nparts nothing else in the suite uses.  A confined queue
    must still produce the oracle's frame (launch geometry sized for its share, all workgroups on the CUs it has)."""
    monkeypatch.setenv("HLMI_PART_MASK", str(layout))
    streams = [hl.partition_stream(p, nparts) for p in (0, nparts - 1)]
    assert all(streams) and streams[0] != streams[1]
    rng = np.random.default_rng(20 + layout)
    f = rng.integers(0, 65536, (3, 272, 512), dtype=np.uint16)
    want = oracle.local_laplacian(f, 8, 1.0 / 7.0, 1.0)
    for s in streams:
        a, o = hl.Buffer(f), hl.Buffer(np.zeros_like(f))
        hl.set_stream(s)
        hl.local_laplacian(a, 8, 1.0 / 7.0, 1.0, o)
        hl.set_stream(None)
        assert np.array_equal(o.numpy(), want)


@pytest.mark.gpu
def test_result_produced_on_another_stream_is_complete_when_read_back(hl, oracle):
    """`set_stream(S); pipeline(); set_stream(None); copy_to_host()` must wait for S — at a size where a
    missing dependency would be visible (a 4K frame is ~0.15 ms of kernels; the D2H starts within microseconds)."""
    s = hl.partition_stream(1, 2)
    rng = np.random.default_rng(3)
    f = rng.integers(0, 65536, (3, 1080, 1920), dtype=np.uint16)
    want = oracle.local_laplacian(f, 8, 1.0 / 7.0, 1.0)
    for _ in range(3):
        a, o = hl.Buffer(f), hl.Buffer(np.zeros_like(f))
        hl.set_stream(s)
        hl.local_laplacian(a, 8, 1.0 / 7.0, 1.0, o)
        hl.set_stream(None)
        assert np.array_equal(o.numpy(), want)


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_frames_in_flight_on_partition_streams_match_single_calls(hl, oracle):
    """bench_apps.py's throughput mode: distinct frames of bilateral_grid / nl_means / conv_layer_bf16 in flight on four
    frame-queue streams (two frames per stream, back to back, no host synchronisation in between).  Every output must be
    what the same call produces alone on the default stream — and, for the bit-exact pipelines, the oracle's."""
    streams = [hl.partition_stream(p, 4) for p in range(4)]
    assert all(streams)
    rng = np.random.default_rng(11)
    nf = 8
    gray = [rng.random((270, 480), dtype=np.float32) for _ in range(nf)]
    rgb = [rng.random((3, 96, 160), dtype=np.float32) for _ in range(nf)]
    cin = [rng.uniform(-1, 1, (2, 18, 18, 128)).astype(np.float32) for _ in range(nf)]
    cf = rng.uniform(-1, 1, (128, 3, 3, 128)).astype(np.float32)
    cb = rng.uniform(-1, 1, 128).astype(np.float32)
    filt, bias = hl.Buffer(cf), hl.Buffer(cb)

    def run_all(use_streams):
        outs = []
        for i in range(nf):
            if use_streams:
                hl.set_stream(streams[i % 4])
            bg_o = hl.Buffer(np.zeros_like(gray[i]))
            hl.bilateral_grid(hl.Buffer(gray[i]), 0.1, bg_o)
            nl_o = hl.Buffer(np.zeros_like(rgb[i]))
            hl.nl_means(hl.Buffer(rgb[i]), 7, 7, 0.12, nl_o)
            cv_o = hl.Buffer(np.zeros((2, 16, 16, 128), np.float32))
            hl.conv_layer_bf16(hl.Buffer(cin[i]), filt, bias, cv_o)
            outs.append((bg_o, nl_o, cv_o))
        hl.set_stream(None)
        return [tuple(b.numpy().copy() for b in t) for t in outs]

    single = run_all(False)
    for rep in range(2):
        batched = run_all(True)
        for i in range(nf):
            for k, name in enumerate(("bilateral_grid", "nl_means", "conv_layer_bf16")):
                assert np.array_equal(batched[i][k].view(np.uint32), single[i][k].view(np.uint32)), f"rep {rep} frame {i}: {name} differs"
    for i in (0, nf - 1):
        assert np.array_equal(single[i][0].view(np.uint32), oracle.bilateral_grid(gray[i], 0.1).view(np.uint32))
        assert np.array_equal(single[i][1].view(np.uint32), oracle.nl_means(rgb[i], 7, 7, 0.12).view(np.uint32))
