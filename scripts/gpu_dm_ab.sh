#!/bin/bash
# ll_down_multi's tile edge at level 7 (4 = default: 486 workgroups; 8: 135; 2: 1.9 k): frame rate with four frames in flight, one call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06mid
F="--no-cpu-baseline --no-other-configs --no-variants --no-ceiling --steps 40 --warmup 3"
L=$GRAFT_REPO_ROOT/halide_amd/lib
run() {
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  v=$(env "${envs[@]}" timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'], d['ms_per_call_one_stream']['noise'])")
  echo "${envs[*]} $* : $v" | tee -a gpurun_out/r06mid/dm_ab.txt
}
HLMI_LIB=$L/libhlmi_dm8.so python -m pytest tests/test_local_laplacian.py -m gpu -x -q -k "matches_oracle" 2>&1 | tail -2
for round in 1 2 3; do
  run HLMI_LIB=$L/libhlmi.so -- --partitions 4
  run HLMI_LIB=$L/libhlmi_dm8.so -- --partitions 4
  run HLMI_LIB=$L/libhlmi_dm2.so -- --partitions 4
done
