"""hlmi_run_batch — the in-process frame sharder (frames are independent units; one host thread and one
HIP stream per device, device chosen per thread with halide_set_gpu_device, no data-path collective).  Reference
pattern: test/generator/gpu_multi_context_threaded_aottest.cpp (one thread per context running the same pipeline),
src/runtime/HalideRuntime.h:1014-1019 (halide_set_gpu_device).

On a one-GPU box the N>1 path is exercised by listing device 0 more than once (each listing gets its own worker thread
and stream); with more devices visible every device takes a share and results come back from the device that produced
them."""
import numpy as np
import pytest


def _frames(hl, n, h=120, w=200, seed=0):
    rng = np.random.default_rng(seed)
    imgs = [rng.integers(0, 65536, (3, h + 8 * (i % 3), w), dtype=np.uint16) for i in range(n)]
    ins = [hl.Buffer(a) for a in imgs]
    outs = [hl.Buffer(np.zeros_like(a)) for a in imgs]
    return imgs, ins, outs


def test_errors_of_worker_threads_come_back_to_the_caller(hl):
    """A frame whose arguments fail the prologue stops the batch with that frame's code (-8); on a box without a GPU
    the workers cannot even get a stream and the batch fails with -29 — loudly, never by computing elsewhere."""
    import torch
    imgs, ins, outs = _frames(hl, 3)
    ins[1].dim(0).stride = 2
    with pytest.raises(hl.HalideError) as e:
        hl.run_batch("local_laplacian", [(i, 8, 1.0 / 7.0, 1.0, o) for i, o in zip(ins[1:], outs[1:])], devices=[0])
    if torch.cuda.is_available():
        assert e.value.code == -8 and "stride.0" in str(e.value)
    else:
        assert e.value.code == -29
    with pytest.raises(TypeError):
        hl.run_batch("local_laplacian", [(ins[0], 8, outs[0])], devices=[0])


def test_empty_batch_is_a_no_op(hl):
    assert hl.run_batch("local_laplacian", [], devices=[0, 0]) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("devices,spd", [([0], 1), ([0, 0], 1), ([0, 0, 0], 1), ([0], 4)])
def test_batch_on_aliased_devices_matches_the_oracle(hl, oracle, devices, spd):
    imgs, ins, outs = _frames(hl, 7, seed=len(devices) + spd)
    assert hl.run_batch("local_laplacian", [(i, 8, 1.0 / 7.0, 1.0, o) for i, o in zip(ins, outs)], devices=devices,
                        streams_per_device=spd) == 0
    for img, o in zip(imgs, outs):
        assert o.device_dirty
        assert np.array_equal(o.numpy(), oracle.local_laplacian(img, 8, 1.0 / 7.0, 1.0))


@pytest.mark.gpu
def test_batch_over_every_visible_device(hl, oracle):
    """With N devices visible frame i runs on device i % N and its output lives there; copy_to_host finds it."""
    n = hl.device_count()
    assert n >= 1
    imgs, ins, outs = _frames(hl, 2 * n + 1, h=96, w=160, seed=5)
    assert hl.run_batch("local_laplacian", [(i, 8, 1.0 / 7.0, 1.0, o) for i, o in zip(ins, outs)]) == 0
    for img, o in zip(imgs, outs):
        assert np.array_equal(o.numpy(), oracle.local_laplacian(img, 8, 1.0 / 7.0, 1.0))
    if n > 1:
        # a buffer resident on device 1 handed to a call that runs on device 0 is refused, not silently read over xGMI
        hl.set_gpu_device(0)
        with pytest.raises(hl.HalideError) as e:
            hl.local_laplacian(ins[1], 8, 1.0 / 7.0, 1.0, hl.Buffer(np.zeros_like(imgs[1])))
        hl.set_gpu_device(-1)
        assert e.value.code == -42


@pytest.mark.gpu
def test_batch_of_float_frames_nl_means(hl, oracle):
    """BASELINE configs[3] in miniature: a batch of nl_means frames over two workers."""
    rng = np.random.default_rng(2)
    imgs = [rng.random((3, 40, 56), dtype=np.float32) for _ in range(5)]
    ins, outs = [hl.Buffer(a) for a in imgs], [hl.Buffer(np.zeros_like(a)) for a in imgs]
    assert hl.run_batch("nl_means", [(i, 7, 7, 0.12, o) for i, o in zip(ins, outs)], devices=[0, 0]) == 0
    for img, o in zip(imgs, outs):
        assert np.array_equal(o.numpy().view(np.uint32), oracle.nl_means(img, 7, 7, 0.12).view(np.uint32))
