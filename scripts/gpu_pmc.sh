#!/bin/bash
# rocprofv3 PMC pass over bench.py.  Usage: bash scripts/gpu_pmc.sh tag "CTR1 CTR2 ..." ["CTRa CTRb ..." ...]
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --partitions 0 --streams 1"
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0].replace('void ','')[:34]
    acc[(n,r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if k[0].startswith('__amd'): continue
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
  find $OUT/p$i -name "*.csv" -size +2M -delete
done
