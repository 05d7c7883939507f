#!/bin/bash
# defaults now: one ll_down01e workgroup per CU on frame queues, 640 units, RU 32.  How many ll_up0h workgroups sit beside it?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06mask
F="--no-cpu-baseline --no-other-configs --no-variants --no-ceiling --steps 40 --warmup 3"
run() {
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  v=$(env "${envs[@]}" timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'])")
  echo "${envs[*]} $* : $v" | tee -a gpurun_out/r06mask/occ_ab6.txt
}
for round in 1 2; do
  run A=0 -- --partitions 4
  run HLMI_LL_RU=26 -- --partitions 4
  run HLMI_LL_RU=20 -- --partitions 4
  run HLMI_LL_RU=16 -- --partitions 4
  run HLMI_LL_RU=36 -- --partitions 4
done
run HLMI_LL_UNITS0=320 -- --partitions 4
run HLMI_LL_UNITS0=448 -- --partitions 4
run HLMI_LL_FUSE_UP2=0 -- --partitions 4
run A=0 -- --partitions 4
