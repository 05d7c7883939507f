#!/bin/bash
# rocprofv3 session for bench.py: kernel trace + stats, then PMC passes (FETCH_SIZE, WRITE_SIZE separately).
# Usage: bash scripts/gpu_prof.sh [tag] [pmc: 0|1]
TAG=${1:-prof}; PMC=${2:-1}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-variants --partitions 0 --streams 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); echo "== kernel stats ($f)"; cut -d, -f1-4,6,7 "$f" | sed -E 's/\(anonymous namespace\):://g; s/\([^"]*\)//' | head -24
if [ "$PMC" = "1" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- $CMD > $OUT/pmc_$c.log 2>&1
    f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
    echo "== $c per kernel (mean per dispatch, as reported)"; python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, len(v), sum(v)/len(v))
PY
  done
fi
# keep the merge small
find $OUT -name "*.csv" -size +3M -delete
