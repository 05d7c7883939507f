#!/bin/bash
# why are the library's queue streams faster than caller-made plain streams?  same launch geometry (HLMI_STREAM_SHARE=4) on both
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06mask
F="--no-cpu-baseline --no-other-configs --no-variants --no-ceiling --steps 10 --warmup 2"
run() {
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  v=$(env "${envs[@]}" timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'])")
  echo "${envs[*]} $* : $v" | tee -a gpurun_out/r06mask/mask_ab4.txt
}
run A=0 -- --partitions 4
run A=0 -- --partitions 0 --streams 4
run HLMI_STREAM_SHARE=4 -- --partitions 0 --streams 4
run HLMI_STREAM_SHARE=4 GPU_MAX_HW_QUEUES=8 -- --partitions 0 --streams 4
run HLMI_STREAM_SHARE=4 GPU_MAX_HW_QUEUES=2 -- --partitions 0 --streams 4
run HLMI_STREAM_SHARE=4 -- --partitions 0 --streams 3
run HLMI_STREAM_SHARE=4 -- --partitions 0 --streams 2
run GPU_MAX_HW_QUEUES=8 -- --partitions 4
run GPU_MAX_HW_QUEUES=8 -- --partitions 6
run A=0 -- --partitions 4
