#!/bin/bash
# ll_down01e at one workgroup per CU (unused LDS) with four frames in flight and on one stream
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06mask
F="--no-cpu-baseline --no-other-configs --no-variants --no-ceiling --steps 10 --warmup 2"
run() {
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  v=$(env "${envs[@]}" timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'])")
  echo "${envs[*]} $* : $v" | tee -a gpurun_out/r06mask/occ_ab.txt
}
run A=0 -- --partitions 4
run HLMI_LL_D01_PAD_LDS=4096 -- --partitions 4
run HLMI_LL_D01_PAD_LDS=4096 HLMI_LL_UNITS0=256 -- --partitions 4
run HLMI_LL_D01_PAD_LDS=4096 HLMI_LL_UNITS0=1024 -- --partitions 4
run A=0 -- --partitions 0 --streams 1
run HLMI_LL_D01_PAD_LDS=4096 -- --partitions 0 --streams 1
run A=0 -- --partitions 4
run HLMI_LL_D01_PAD_LDS=4096 -- --partitions 4
