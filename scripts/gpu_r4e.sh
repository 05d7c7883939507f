#!/bin/bash
# partition-count sweep with the re-cut kernels + PMC traffic of the frame
TAG=${1:-r4e}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for P in 2 3 4 5 6 8; do
  timeout 300 python bench.py --steps 6 --warmup 2 --partitions $P --no-cpu-baseline --no-other-configs --no-ceiling --no-variants 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('partitions $P', d['value'], d['config']['frame_ms'])" | tee -a $OUT/partitions.log
done
timeout 300 python bench.py --steps 6 --warmup 2 --partitions 4 --streams-per-partition 2 --no-cpu-baseline --no-other-configs --no-ceiling --no-variants 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('partitions 4 x 2', d['value'], d['config']['frame_ms'])" | tee -a $OUT/partitions.log
cd /tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-other-configs --no-ceiling --partitions 0 --streams 1"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- $CMD > $OUT/pmc_$c.log 2>&1
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0].replace('void ','')[:40]
    acc[(n,r['Grid_Size'],r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in acc.items():
    if k[0].startswith('__amd'): continue
    print(k, round(sum(v)/len(v)), len(v))
PY
done 2>&1 | tee $OUT/traffic.log
find $OUT -name "*.csv" -size +2M -delete
