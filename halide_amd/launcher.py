"""One process per GPU, started by the benchmark itself.

`python bench.py --gpus N` must be the WHOLE command: when it is not already running under a launcher
(`WORLD_SIZE` unset) and N > 1, `ensure_ranks` re-executes the script under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>`,
one rank per GPU, and exits with the launcher's status.  It refuses loudly when fewer than N devices are visible,
and `check_world` refuses a launcher whose WORLD_SIZE disagrees with `--gpus` — so the JSON line's `n_gpus` can only
ever be the N that was asked for.

The reference has no distributed layer; its multi-device pattern is one host thread + context per
device (`test/generator/gpu_multi_context_threaded_aottest.cpp`, `halide_set_gpu_device`
`src/runtime/HalideRuntime.h:1014-1019`), which `hlmi_run_batch` implements in-process; the benchmark contract asks
for one PROCESS per GPU over RCCL, which is what this module arranges.

Launcher self-test (tests/test_launcher.py): with `HLMI_BENCH_STUB=1` the scripts replace the pipeline call by a no-op and
the "nccl" backend by "gloo", so that the launch / rendezvous / reduction / JSON path runs on a box without GPUs.  A
stub line says `"data": "stub"` and carries no roofline: it can never be mistaken for a measurement.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Tuple


def stub_mode() -> bool:
    return os.environ.get("HLMI_BENCH_STUB", "") == "1"


def backend() -> str:
    """"nccl" IS RCCL on ROCm; gloo only in the launcher self-test."""
    return "gloo" if stub_mode() else "nccl"


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_devices() -> int:
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def ensure_ranks(n_gpus: int, script: str, argv: List[str]) -> None:
    """Returns when this process is a rank (or N == 1); otherwise launches N ranks of `script argv` and exits."""
    if n_gpus < 1:
        raise SystemExit(f"--gpus {n_gpus}: need at least one GPU")
    if "WORLD_SIZE" in os.environ or n_gpus == 1:
        return
    if not stub_mode():
        have = visible_devices()
        if have < n_gpus:
            raise SystemExit(f"--gpus {n_gpus} asked for, but only {have} HIP device(s) are visible: refusing to run a smaller job "
                             f"under the same name")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    sys.stdout.flush()
    sys.stderr.flush()
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def check_world(n_gpus: int) -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the launcher's environment; world must be the N that --gpus named."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != n_gpus:
        raise SystemExit(f"launched with WORLD_SIZE={world} but --gpus {n_gpus}: the two must agree (run "
                         f"`python {os.path.basename(sys.argv[0])} --gpus {world}` and let it start its own ranks)")
    if not stub_mode():
        have = visible_devices()
        if have <= local_rank:
            raise SystemExit(f"rank {rank}: local rank {local_rank} has no HIP device ({have} visible); no CPU fallback exists")
    return rank, local_rank, world


def init_process_group(local_rank: int, world: int):
    """torch.distributed over RCCL (gloo in the stub), or None for a single rank.  Returns (dist, device, ranks_seen):
    every rank adds 1 to an all-reduced counter — proof that the backend really spans `world` ranks."""
    import torch
    device = "cpu" if stub_mode() else torch.device("cuda", local_rank)
    if world == 1:
        return None, device, 1
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if stub_mode():
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=device)
    ones = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(ones)
    return dist, device, int(ones.item())
