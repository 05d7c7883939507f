#!/bin/bash
# rocprofv3 kernel durations for bgu (and whatever --only names)
ONLY=${1:-bgu}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$ONLY; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- bash -c "cd $R && python bench_apps.py --only $ONLY --samples 1" > $OUT/kt.log 2>&1)
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/kt/*kernel_stats.csv | cut -d, -f1-8 | head -12
