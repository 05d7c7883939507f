#!/bin/bash
# kernel timeline of the headline loop (four frames in flight): rocprofv3 kernel trace -> scripts/timeline_overlap.py
cd $GRAFT_REPO_ROOT; OUT=$GRAFT_REPO_ROOT/gpurun_out/r06tl; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-variants --no-other-configs --no-ceiling $*"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python scripts/timeline_overlap.py "$f" | tee $OUT/overlap.txt
find $OUT -name "*.csv" -size +8M -delete
