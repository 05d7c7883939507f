#!/usr/bin/env python3
"""bench_batch.py — BASELINE.json configs[3]: a batch of 32 nl_means frames (7x7 search / 7x7 patch, f32 1920x1080x3)
that originates and ends on GPU 0 and is processed by all N GPUs of the node.

    python bench_batch.py [--gpus N]        # starts its own N ranks (halide_amd/launcher.py); also runs under an external
                                            # `python -m torch.distributed.run --nproc-per-node N ... bench_batch.py --gpus N`

This is the one workload of the suite with a real exchange step: frames travel rank 0 -> rank r and
results travel back as point-to-point RCCL transfers over xGMI, pipelined against the compute
(`halide_amd.sharding.scatter_process_gather`); a 1080p frame is 24.9 MB (≈0.16 ms on one 153 GB/s link) against
≈0.28 ms of nl_means, so the links of rank 0 are not the bound up to 8 GPUs.  The timed region covers scatter +
compute + gather ("scaling": "strong": the batch is fixed).  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
W, H, BATCH = 1920, 1080, 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    args = ap.parse_args()
    import torch
    from halide_amd import launcher, sharding
    launcher.ensure_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:])   # N > 1 and no launcher yet: start N ranks, exit
    rank, local_rank, world = launcher.check_world(args.gpus)
    stub = launcher.stub_mode()    # launcher self-test (tests/test_launcher.py): gloo, CPU tensors, identity instead of nl_means
    if not stub:
        import halide_amd.torch_ops  # noqa: F401  (registers torch.ops.hlmi.*)
        if not torch.cuda.is_available():
            raise SystemExit("bench_batch.py needs HIP devices (no CPU fallback)")
        torch.cuda.set_device(local_rank)
    dist, dev, ranks_seen = launcher.init_process_group(local_rank, world)
    shape = (3, 8, 16) if stub else (3, H, W)
    like = torch.empty(shape, dtype=torch.float32, device=dev)
    frames = None
    if rank == 0:
        g = torch.Generator(device=dev).manual_seed(0)
        frames = [torch.rand(shape, generator=g, device=dev) for _ in range(BATCH)]

    def process(t):
        return t.clone() if stub else torch.ops.hlmi.nl_means(t, 7, 7, 0.12)

    def step():
        return sharding.scatter_process_gather(frames, process, dist, root=0, like=like, n_items=BATCH)

    def barrier():
        if not stub:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0, dist, dev)
    if rank == 0:
        assert len(out) == BATCH and all(o.shape == shape for o in out)
        per_batch = elapsed / args.steps
        print(json.dumps({"metric": "megapixels/sec nl_means 7x7/7x7 f32 1920x1080x3, batch of 32 frames from and to GPU 0",
                          "value": round(BATCH * W * H / per_batch / 1e6, 1), "unit": "Mpx/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(per_batch * 1e3, 3), "higher_is_better": True,
                          "scaling": "strong", "dtype": "f32", "data": "stub" if stub else "synthetic",
                          "config": {"workload": "stub" if stub else "apps/nl_means patch 7 search 7 sigma 0.12", "batch": BATCH,
                                     "rccl_ranks": ranks_seen,
                                     "exchange": "isend/irecv per frame, rank 0 <-> rank r (RCCL over xGMI), pipelined with compute"}}),
              flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
