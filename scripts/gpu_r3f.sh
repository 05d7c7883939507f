#!/bin/bash
TAG=${1:-r3f}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for st in 0 1 2 3; do
  HLMI_CONVP_STAG=$st timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$st -o kt -- python $R/bench_apps.py --only conv_layer_bf16 --samples 2 > $OUT/kt$st.log 2>&1
  python3 - $OUT/kt$st/kt_kernel_stats.csv $st <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    if 'conv3x3' in n: print(f"stag={sys.argv[2]} {n:30s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1000:8.2f}us min={float(r['MinNs'])/1000:8.2f}")
PY
done | tee $OUT/stag.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.csv" -size +1M -delete
