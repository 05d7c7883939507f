#!/bin/bash
# quick parity + timing of named pipelines: TESTS="tests/test_x.py ..." ONLY="a,b" bash scripts/gpu_quick.sh tag
TAG=${1:-q}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest $TESTS -m gpu -q --tb=short -x 2>&1 | tail -12 | tee $OUT/pytest.log
timeout 300 python bench_apps.py --only $ONLY 2>/dev/null | grep pipeline | cut -c1-700 | tee $OUT/bench.jsonl
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $R/bench_apps.py --only $ONLY --samples 2 > $OUT/kt.log 2>&1
python3 - $OUT/kt/kt_kernel_stats.csv <<'PY' | tee $OUT/kstats.txt
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    print(f"{n:45s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1000:8.2f}us min={float(r['MinNs'])/1000:8.2f} max={float(r['MaxNs'])/1000:8.2f}")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.csv" -size +1M -delete
