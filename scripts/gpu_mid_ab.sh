#!/bin/bash
# ll_mid (ll_down_multi + ll_up_multi in one launch) against the two launches: parity, then alternating A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06mid
python -m pytest tests/test_local_laplacian.py tests/test_threads.py tests/test_fuzz_slice.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r06mid/pytest.log
timeout 300 python scripts/fuzz_parity.py --only local_laplacian --seconds 40 --seed 71 2>&1 | tail -3 | tee -a gpurun_out/r06mid/pytest.log
timeout 300 python scripts/fuzz_parity.py --only local_laplacian --seconds 40 --seed 72 --frame-queue 2>&1 | tail -3 | tee -a gpurun_out/r06mid/pytest.log
F="--no-cpu-baseline --no-other-configs --no-variants --no-ceiling --steps 40 --warmup 3"
run() {
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  v=$(env "${envs[@]}" timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'], d['ms_per_call_one_stream']['noise'])")
  echo "${envs[*]} $* : $v" | tee -a gpurun_out/r06mid/mid_ab.txt
}
for round in 1 2 3; do
  run HLMI_LL_FUSE_MID=0 -- --partitions 4
  run HLMI_LL_FUSE_MID=1 -- --partitions 4
done
run HLMI_LL_FUSE_MID=0 -- --partitions 0 --streams 1
run HLMI_LL_FUSE_MID=1 -- --partitions 0 --streams 1
