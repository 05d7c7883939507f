"""N>1 path on CPU: frame sharding + the collectives the multi-GPU bench uses, world_size 2 over gloo.

The GPU pipeline cannot run here, so the per-frame `process` of this test is the CPU oracle (test infrastructure);
what is under test is the sharding/collective logic of halide_amd/sharding.py that bench.py drives with RCCL."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frames(n):
    rng = np.random.default_rng(42)
    return [rng.integers(0, 65536, (34, 50), dtype=np.uint16) for _ in range(n)]


def _worker(rank, world, port, n_frames, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import oracle_lib
    from halide_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = _frames(n_frames)
    mine = sharding.run_batch(frames, oracle_lib.blur, rank, world)
    assert sorted(mine) == sharding.shard(n_frames, rank, world)
    digests = sharding.gather_digests({i: sharding.digest64(v) for i, v in mine.items()}, n_frames, dist)
    slowest = sharding.max_over_ranks(0.25 * (rank + 1), dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, digests, slowest))


def test_shard_partitions_the_batch():
    from halide_amd import sharding
    for n in (0, 1, 5, 8, 33):
        for world in (1, 2, 3, 8):
            parts = [sharding.shard(n, r, world) for r in range(world)]
            assert sorted(i for p in parts for i in p) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        sharding.shard(4, 2, 2)


def test_two_ranks_over_gloo_match_single_process():
    import torch.multiprocessing as mp
    import oracle_lib
    from halide_amd import sharding
    n_frames, world = 5, 2
    want = [sharding.digest64(oracle_lib.blur(f)) for f in _frames(n_frames)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, digests, slowest in results:
        assert digests == want          # every rank sees the digests of the whole batch
        assert slowest == 0.5           # max over ranks of 0.25*(rank+1)


def _sg_worker(rank, world, port, n_frames, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from halide_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def process(t):
        calls.append(int(t[0, 0].item()))
        return t * 2 + 1

    like = torch.zeros((6, 5), dtype=torch.float32)
    frames = [torch.full((6, 5), float(i)) for i in range(n_frames)] if rank == 0 else None
    res = sharding.scatter_process_gather(frames, process, dist, root=0, like=like, n_items=n_frames)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, sorted(calls), None if res is None else [float(r[0, 0].item()) for r in res]))


def test_scatter_process_gather_over_gloo():
    """configs[3]'s exchange pattern: the batch starts and ends on rank 0, every rank computes its round-robin share."""
    import torch.multiprocessing as mp
    n_frames, world = 7, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sg_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {r: (calls, res) for r, calls, res in (q.get(timeout=180) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][0] == [0, 2, 4, 6] and results[1][0] == [1, 3, 5]      # who computed what
    assert results[0][1] == [2.0 * i + 1 for i in range(n_frames)] and results[1][1] is None


def test_scatter_process_gather_single_process():
    import torch
    from halide_amd import sharding
    out = sharding.scatter_process_gather([torch.full((2, 2), float(i)) for i in range(3)], lambda t: t + 1)
    assert [float(o[0, 0]) for o in out] == [1.0, 2.0, 3.0]


def test_scatter_process_gather_32_frames_over_8_ranks():
    """BASELINE.json configs[3] at its real shape: a batch of 32 frames on rank 0, 8 ranks, every rank computes its 4."""
    import torch.multiprocessing as mp
    n_frames, world = 32, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sg_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {r: (calls, res) for r, calls, res in (q.get(timeout=600) for _ in range(world))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(world):
        assert results[r][0] == list(range(r, n_frames, world))      # who computed what: round robin, 4 frames each
    assert results[0][1] == [2.0 * i + 1 for i in range(n_frames)]   # the batch, in order, back on rank 0 only
    assert all(results[r][1] is None for r in range(1, world))


def test_eight_ranks_over_gloo_match_single_process():
    import torch.multiprocessing as mp
    import oracle_lib
    from halide_amd import sharding
    n_frames, world = 19, 8          # not a multiple of the world size: ranks 0..2 take three frames, the others two
    want = [sharding.digest64(oracle_lib.blur(f)) for f in _frames(n_frames)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, digests, slowest in results:
        assert digests == want
        assert slowest == 0.25 * world
