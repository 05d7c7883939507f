#!/bin/bash
# Only the single-stream bench + rocprofv3 kernel stats + PMC traffic of scripts/gpu_full.sh (into the same tag directory).
TAG=${1:-full}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== bench 1 stream"; timeout 900 python bench.py --partitions 0 --streams 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_1stream.json
cd /tmp
CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --partitions 0 --streams 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- $CMD > $OUT/pmc_$c.log 2>&1
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.csv" -size +3M -delete
