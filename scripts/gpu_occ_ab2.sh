#!/bin/bash
# ll_down01e at one workgroup per CU (unused LDS): how much LDS to leave, how many queues, which geometry
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06mask
F="--no-cpu-baseline --no-other-configs --no-variants --no-ceiling --steps 10 --warmup 2"
run() {
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  v=$(env "${envs[@]}" timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'])")
  echo "${envs[*]} $* : $v" | tee -a gpurun_out/r06mask/occ_ab2.txt
}
run A=0 -- --partitions 4
for p in 2048 4096 16384 32768 49152 65536; do run HLMI_LL_D01_PAD_LDS=$p -- --partitions 4; done
for n in 3 5 6 8; do run HLMI_LL_D01_PAD_LDS=4096 HLMI_PART_GEOM_CUS=64 -- --partitions $n; done
for u in 384 640 768; do run HLMI_LL_D01_PAD_LDS=4096 HLMI_LL_UNITS0=$u -- --partitions 4; done
for r in 8 16 32; do run HLMI_LL_D01_PAD_LDS=4096 HLMI_LL_RU=$r -- --partitions 4; done
run HLMI_LL_D01_PAD_LDS=4096 HLMI_LL_NT=0 -- --partitions 4
run A=0 -- --partitions 4
run HLMI_LL_D01_PAD_LDS=4096 -- --partitions 4
