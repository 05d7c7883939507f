#!/bin/bash
# wave priority (s_setprio 3) for ll_down01e or for ll_up0h, four frames in flight, alternating, 40 steps per point
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06mask
F="--no-cpu-baseline --no-other-configs --no-variants --no-ceiling --steps 40 --warmup 3"
L=$GRAFT_REPO_ROOT/halide_amd/lib
run() {
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  v=$(env "${envs[@]}" timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'], d['ms_per_call_one_stream']['noise'])")
  echo "${envs[*]} $* : $v" | tee -a gpurun_out/r06mask/prio_ab.txt
}
for round in 1 2 3; do
  run HLMI_LIB=$L/libhlmi.so -- --partitions 4
  run HLMI_LIB=$L/libhlmi_dprio.so -- --partitions 4
  run HLMI_LIB=$L/libhlmi_uprio.so -- --partitions 4
done
