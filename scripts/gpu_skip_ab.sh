#!/bin/bash
# what each launch of the chain costs the frame rate with four frames in flight: the frame rate without it (HLMI_SKIP_LAUNCH: wrong pixels)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06mask
F="--no-cpu-baseline --no-other-configs --no-variants --no-ceiling --steps 40 --warmup 3"
run() {
  local envs=() ; while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  v=$(env "${envs[@]}" timeout 300 python bench.py $F "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'])")
  echo "${envs[*]} $* : $v" | tee -a gpurun_out/r06mask/skip_ab.txt
}
for round in 1 2; do
run A=0 -- --partitions 4
run HLMI_SKIP_LAUNCH=ll_down_strip2:2 -- --partitions 4
run HLMI_SKIP_LAUNCH=ll_down_multi:4 -- --partitions 4
run HLMI_SKIP_LAUNCH=ll_up_multi:3 -- --partitions 4
run HLMI_SKIP_LAUNCH=ll_down_strip2:2,ll_down_multi:4,ll_up_multi:3 -- --partitions 4
run HLMI_SKIP_LAUNCH=ll_up0 -- --partitions 4
run HLMI_SKIP_LAUNCH=ll_down01 -- --partitions 4
run HLMI_SKIP_LAUNCH=ll_down01,ll_up0 -- --partitions 4
done
