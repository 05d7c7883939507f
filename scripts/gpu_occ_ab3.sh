#!/bin/bash
# alternating A/B, 40 steps each, four rounds: ll_down01e at two / one workgroup per CU with four frames in flight
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06mask
F="--no-cpu-baseline --no-other-configs --no-variants --no-ceiling --steps 40 --warmup 3"
run() {
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  v=$(env "${envs[@]}" timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'])")
  echo "${envs[*]} $* : $v" | tee -a gpurun_out/r06mask/occ_ab3.txt
}
for round in 1 2 3 4; do
  run A=0 -- --partitions 4
  run HLMI_LL_D01_PAD_LDS=4096 -- --partitions 4
  run HLMI_LL_D01_PAD_LDS=4096 HLMI_LL_RU=32 -- --partitions 4
  run HLMI_LL_D01_PAD_LDS=4096 HLMI_LL_RU=32 HLMI_LL_UNITS0=640 -- --partitions 4
  run HLMI_LL_D01_PAD_LDS=12288 HLMI_LL_RU=32 -- --partitions 4
done
