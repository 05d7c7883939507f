#!/bin/bash
# one ll_down01e workgroup per CU: units x rows-per-wave sweep, 40 steps per point
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06mask
F="--no-cpu-baseline --no-other-configs --no-variants --no-ceiling --steps 40 --warmup 3"
run() {
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  v=$(env "${envs[@]}" timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'])")
  echo "${envs[*]} $* : $v" | tee -a gpurun_out/r06mask/occ_ab4.txt
}
run A=0 -- --partitions 4
for u in 512 576 640 704 768 896; do
  for r in 32 40 48; do
    run HLMI_LL_D01_PAD_LDS=4096 HLMI_LL_RU=$r HLMI_LL_UNITS0=$u -- --partitions 4
  done
done
run HLMI_LL_D01_PAD_LDS=4096 HLMI_LL_RU=64 HLMI_LL_UNITS0=640 -- --partitions 4
run HLMI_LL_D01_PAD_LDS=4096 HLMI_LL_RU=28 HLMI_LL_UNITS0=640 -- --partitions 4
run HLMI_LL_D01_PAD_LDS=8192 HLMI_LL_RU=32 HLMI_LL_UNITS0=640 -- --partitions 4
run A=0 -- --partitions 4
