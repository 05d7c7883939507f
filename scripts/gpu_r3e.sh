#!/bin/bash
TAG=${1:-r3e}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_layer.py -m gpu -q --tb=short -x 2>&1 | tail -5 | tee $OUT/pytest.log
cd /tmp
for a in ${ABLS:-0 2 7}; do
  HLMI_CONVP_ABL=$a timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$a -o kt -- python $R/bench_apps.py --only conv_layer_bf16 --samples 2 > $OUT/kt$a.log 2>&1
  python3 - $OUT/kt$a/kt_kernel_stats.csv $a <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    if 'conv3x3' in n: print(f"abl={sys.argv[2]} {n:30s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1000:8.2f}us min={float(r['MinNs'])/1000:8.2f}")
PY
done | tee $OUT/abl.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.csv" -size +1M -delete
