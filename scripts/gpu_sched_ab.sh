#!/bin/bash
# frame scheduling A/B of the headline loop (noise input): CU partitions vs plain streams
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06sched
F="--no-cpu-baseline --no-other-configs --no-variants --no-ceiling --steps 10 --warmup 2"
for cfg in "--partitions 4" "--partitions 2" "--partitions 8" "--partitions 4 --streams-per-partition 2" "--partitions 2 --streams-per-partition 2" "--partitions 0 --streams 2" "--partitions 0 --streams 3" "--partitions 0 --streams 4" "--partitions 0 --streams 8" "--partitions 4"; do
  v=$(timeout 300 python bench.py $F $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'], d['clock_state_timed_region'])")
  echo "$cfg : $v" | tee -a gpurun_out/r06sched/sched.txt
done
