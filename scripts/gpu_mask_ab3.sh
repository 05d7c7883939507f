#!/bin/bash
# throughput-mode launch geometry of ll_down01e (units) and ll_up0h (RU) on four queues
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06mask
F="--no-cpu-baseline --no-other-configs --no-variants --no-ceiling --steps 10 --warmup 2"
run() {
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  v=$(env "${envs[@]}" timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'])")
  echo "${envs[*]} $* : $v" | tee -a gpurun_out/r06mask/mask_ab3.txt
}
run HLMI_PART_MASK=3 -- --partitions 4
for u in 256 384 512 640 768 896 1024 1536 2048; do run HLMI_PART_MASK=3 HLMI_LL_UNITS0=$u -- --partitions 4; done
for r in 8 12 24 32; do run HLMI_PART_MASK=3 HLMI_LL_RU=$r -- --partitions 4; done
run HLMI_PART_MASK=3 HLMI_LL_UNITS0=512 HLMI_LL_RU=24 -- --partitions 4
run HLMI_PART_MASK=3 -- --partitions 4
