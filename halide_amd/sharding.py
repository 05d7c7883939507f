"""Frame sharding for batched pipelines over the GPUs of one node (one process per GPU).

Frames are independent units (every AOT entry point is a pure function of its own buffers), so the N>1 path has
no data-path collective: rank r owns frames r, r+world, r+2*world, ... of the batch, produces them in its own
HBM, and `torch.distributed` (RCCL on GPUs, gloo in the CPU tests) is used only for
  * the timing protocol of bench.py (barrier, max-over-ranks of the elapsed time),
  * an optional integrity exchange: every rank contributes one 64-bit digest per frame it produced and all
    ranks receive the digests of the whole batch (`gather_digests`), 8 bytes per frame.
The reference has no counterpart (it has no distributed layer at all).
"""
from __future__ import annotations

import hashlib
from typing import Callable, Dict, Iterable, List

import numpy as np


def shard(n_items: int, rank: int, world: int) -> List[int]:
    """Indices of the batch items rank `rank` of `world` owns (round-robin, so any prefix of the batch is balanced)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, n_items, world))


def digest64(arr: np.ndarray) -> int:
    """Order-sensitive 64-bit digest of an array's bytes (first 8 bytes of SHA-256)."""
    return int.from_bytes(hashlib.sha256(np.ascontiguousarray(arr).tobytes()).digest()[:8], "little", signed=True)


def max_over_ranks(value: float, dist=None, device="cpu") -> float:
    """The bench contract's max over ranks of a per-rank elapsed time."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_digests(local: Dict[int, int], n_items: int, dist=None, device="cpu") -> List[int]:
    """All-gather of per-frame digests: returns the digest of every item of the batch, on every rank."""
    world = 1 if dist is None or not dist.is_initialized() else dist.get_world_size()
    if world == 1:
        return [local[i] for i in range(n_items)]
    import torch
    rank = dist.get_rank()
    per_rank = (n_items + world - 1) // world
    mine = torch.zeros(per_rank, dtype=torch.int64, device=device)
    for slot, idx in enumerate(shard(n_items, rank, world)):
        mine[slot] = local[idx]
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    out = [0] * n_items
    for r in range(world):
        for slot, idx in enumerate(shard(n_items, r, world)):
            out[idx] = int(gathered[r][slot].item())
    return out


def run_batch(frames: Iterable, process: Callable, rank: int, world: int) -> Dict[int, object]:
    """Apply `process(frame)` to the frames this rank owns; returns {batch index: result}."""
    frames = list(frames)
    return {i: process(frames[i]) for i in shard(len(frames), rank, world)}


def scatter_process_gather(frames, process: Callable, dist=None, root: int = 0, like=None, n_items: int | None = None):
    """The one workload of BASELINE.json with a real exchange step (configs[3]: a batch of nl_means frames that
    originates and terminates on ONE GPU): rank `root` holds `frames` (a list of equally shaped tensors), every rank
    processes its round-robin share, the results return to `root`.

    Point-to-point transfers (`isend` / `irecv`: RCCL over xGMI with the "nccl" backend, one link per peer) instead
    of a scatter/gather collective, because they pipeline: all receives are posted up front, rank r computes frame k
    while frames k+1.. are still arriving and returns each result as soon as it exists; `root` works on its own share
    meanwhile.  xGMI is point-to-point, so the seven peers of a node load seven different links of `root`.

    `process(tensor) -> tensor` must return a tensor shaped like its input.  Non-root ranks pass `frames=None`,
    `like` = a tensor with the frames' shape/dtype/device and `n_items`.  Returns the list of results on `root`
    (batch order), None elsewhere.  world == 1: plain local loop."""
    world = 1 if dist is None or not dist.is_initialized() else dist.get_world_size()
    if world == 1:
        return [process(f) for f in frames]
    import torch
    rank = dist.get_rank()
    if rank == root:
        n = len(frames)
        sends = [dist.isend(frames[i], dst=(i % world), tag=i) for i in range(n) if i % world != root]
        results = [None] * n
        recvs = []
        for i in range(n):
            if i % world != root:
                results[i] = torch.empty_like(frames[i])
                recvs.append(dist.irecv(results[i], src=(i % world), tag=n + i))
        for i in shard(n, root, world):  # own share, overlapped with the transfers
            results[i] = process(frames[i])
        for w in sends + recvs:
            w.wait()
        return results
    n = int(n_items)
    mine = shard(n, rank, world)
    bufs = {i: torch.empty_like(like) for i in mine}
    recvs = {i: dist.irecv(bufs[i], src=root, tag=i) for i in mine}
    sends, keep = [], []
    for i in mine:
        recvs[i].wait()
        out = process(bufs[i])
        keep.append(out)
        sends.append(dist.isend(out, dst=root, tag=n + i))
    for w in sends:
        w.wait()
    return None
