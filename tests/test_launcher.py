"""`python bench.py --gpus N` is the whole command: the script starts its own N ranks (halide_amd/launcher.py).

CPU-runnable: HLMI_BENCH_STUB=1 replaces the pipeline call by a no-op and RCCL by gloo, everything else — the re-exec under
torch.distributed.run, the rendezvous on 127.0.0.1, the all-reduced rank count, barrier, max-over-ranks, the JSON line — is
the code path of the real run.  The reference has no distributed layer; the contract tested here is this
repository's bench contract."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *argv, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS"):   # a caller's OMP setting would be kept
        env.pop(k, None)
    env["HLMI_BENCH_STUB"] = "1"
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, script), *argv], env=env, capture_output=True, text=True,
                          timeout=timeout)


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


@pytest.mark.parametrize("script", ["bench.py", "bench_batch.py"])
def test_self_launch_two_ranks(script):
    p = _run(script, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stdout + p.stderr
    d = _json_line(p.stdout)
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["data"] == "stub"
    assert d["steps"] == 3 and d["warmup"] == 1


def test_single_rank_needs_no_launcher():
    p = _run("bench.py", "--gpus", "1", "--steps", "2")
    assert p.returncode == 0, p.stdout + p.stderr
    d = _json_line(p.stdout)
    assert d["n_gpus"] == 1 and d["config"]["rccl_ranks"] == 1


def test_world_size_must_agree_with_gpus():
    # a launcher that started 1 rank while --gpus says 2 must be refused, not silently benchmarked as one GPU
    p = _run("bench.py", "--gpus", "2", "--steps", "2", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0
    assert "WORLD_SIZE=1" in p.stderr and "--gpus 2" in p.stderr


def test_refuses_more_gpus_than_visible():
    # the real (non-stub) path: this box has fewer than 64 HIP devices, so the launcher must refuse before starting anything
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "HLMI_BENCH_STUB"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0
    assert "HIP device(s) are visible" in p.stderr


def test_self_launch_eight_ranks_the_target_world_size():
    """The node this is built for has 8 GPUs: the whole launch path at world size 8 (gloo stub), every rank bound to its own
    local rank (= HIP device index in the real run) and the host threads of the CPU legs divided between the ranks."""
    p = _run("bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    d = _json_line(p.stdout)
    assert d["n_gpus"] == 8 and d["config"]["rccl_ranks"] == 8
    table = sorted(d["config"]["rank_table"])
    assert [t[0] for t in table] == list(range(8)) and [t[1] for t in table] == list(range(8))   # rank r -> device r, no sharing
    want_threads = max(1, (os.cpu_count() or 1) // 8)
    assert all(t[2] == want_threads for t in table), table
    # weak scaling: 8 ranks x FRAMES_PER_STEP frames, rank 0 owns every 8th
    assert d["config"]["frames_of_rank0"] == list(range(0, 64, 8))
