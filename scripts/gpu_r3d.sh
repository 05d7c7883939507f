#!/bin/bash
TAG=${1:-r3d}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_layer.py -m gpu -q --tb=short -x 2>&1 | tail -8 | tee $OUT/pytest.log
timeout 300 python bench_apps.py --only conv_layer_bf16 2>/dev/null | grep pipeline | cut -c1-330 | tee $OUT/bench_new.jsonl
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $R/bench_apps.py --only ${ONLY:-conv_layer_bf16} > $OUT/kt.log 2>&1
python3 - $OUT/kt/kt_kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    print(f"{n:45s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1000:8.2f}us min={float(r['MinNs'])/1000:8.2f} max={float(r['MaxNs'])/1000:8.2f}")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.csv" -size +2M -delete
