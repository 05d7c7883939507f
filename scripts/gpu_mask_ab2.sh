#!/bin/bash
# N unmasked queues (HLMI_PART_MASK=3) x the CU count the launch geometry is sized for
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06mask
F="--no-cpu-baseline --no-other-configs --no-variants --no-ceiling --steps 10 --warmup 2"
run() {
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  v=$(env "${envs[@]}" timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'])")
  echo "${envs[*]} $* : $v" | tee -a gpurun_out/r06mask/mask_ab2.txt
}
run HLMI_PART_MASK=0 -- --partitions 4
for g in 32 48 64 80 96; do run HLMI_PART_MASK=3 HLMI_PART_GEOM_CUS=$g -- --partitions 4; done
for n in 3 5 6 8; do run HLMI_PART_MASK=3 HLMI_PART_GEOM_CUS=64 -- --partitions $n; done
run HLMI_PART_MASK=3 HLMI_PART_GEOM_CUS=64 -- --partitions 2 --streams-per-partition 2
run HLMI_PART_MASK=3 HLMI_PART_GEOM_CUS=48 -- --partitions 6
run HLMI_PART_MASK=0 -- --partitions 4
