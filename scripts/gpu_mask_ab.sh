#!/bin/bash
# real CU partitions (HLMI_PART_MASK=1/2: the same CU slots on every XCD) against round 3-5's mask layout (0: in effect four unmasked queues)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06mask
F="--no-cpu-baseline --no-other-configs --no-variants --no-ceiling --steps 10 --warmup 2"
run() {  # env..., then -- bench flags
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  v=$(env "${envs[@]}" timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'], d['clock_state_timed_region'])")
  echo "${envs[*]} $* : $v" | tee -a gpurun_out/r06mask/mask_ab.txt
}
run HLMI_PART_MASK=0 -- --partitions 4
run HLMI_PART_MASK=1 -- --partitions 4
run HLMI_PART_MASK=2 -- --partitions 4
run HLMI_PART_MASK=1 -- --partitions 2
run HLMI_PART_MASK=1 -- --partitions 8
run HLMI_PART_MASK=1 -- --partitions 4 --streams-per-partition 2
run HLMI_PART_MASK=1 -- --partitions 2 --streams-per-partition 2
run HLMI_PART_MASK=1 HLMI_PART_GEOM_CUS=256 -- --partitions 4
run HLMI_PART_MASK=0 HLMI_PART_GEOM_CUS=256 -- --partitions 4
run HLMI_PART_MASK=0 HLMI_PART_GEOM_CUS=128 -- --partitions 4
run HLMI_PART_MASK=0 -- --partitions 4
run HLMI_PART_MASK=1 -- --partitions 4
