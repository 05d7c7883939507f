#!/bin/bash
# with one ll_down01e workgroup per CU (640 units, 32 rows per ll_up0h wave): workgroups of ll_up0h per CU, units near 640, smooth input
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06mask
F="--no-cpu-baseline --no-other-configs --no-variants --no-ceiling --steps 40 --warmup 3"
run() {
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  v=$(env "${envs[@]}" timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'])")
  echo "${envs[*]} $* : $v" | tee -a gpurun_out/r06mask/occ_ab5.txt
}
B="HLMI_LL_D01_PAD_LDS=4096 HLMI_LL_RU=32 HLMI_LL_UNITS0=640"
run A=0 -- --partitions 4
run $B -- --partitions 4
for p in 4096 8192 16384 30000 60000; do run $B HLMI_LL_UP0_PAD_LDS=$p -- --partitions 4; done
run $B -- --partitions 4
for u in 600 620 660 680; do run HLMI_LL_D01_PAD_LDS=4096 HLMI_LL_RU=32 HLMI_LL_UNITS0=$u -- --partitions 4; done
run $B -- --partitions 4
run $B -- --partitions 5
run $B -- --partitions 3
run $B -- --partitions 2 --streams-per-partition 2
run A=0 -- --partitions 4
