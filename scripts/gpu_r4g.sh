#!/bin/bash
# A/B of run-time switches: frame time (1 stream, 4 partitions), kernel times, PMC traffic.  Usage: bash scripts/gpu_r4g.sh tag "VAR=val VAR2=val" ...
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for v in "$@"; do
  echo "== $v" | tee -a $OUT/ab.log
  (export $v; timeout 300 python scripts/kernel_times.py 2>&1 | grep -E "ll_down01|ll_up0|sum"; timeout 300 python scripts/frame_bench.py 8 4 2>&1 | tail -2
   cd /tmp; CMD="python $R/scripts/kernel_times.py"
   for c in FETCH_SIZE WRITE_SIZE; do
     timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- $CMD > $OUT/pmc_$c.log 2>&1
     f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
     python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0].replace('void ','')[:24]
    acc[(n,r['Grid_Size'],r['Counter_Name'])].append(float(r['Counter_Value']))
tot=0
for k,v in acc.items():
    if k[0].startswith('__amd'): continue
    m=sum(v)/len(v)*1024*(2 if k[2]=='FETCH_SIZE' else 1)/1e6; tot+=m
    if m>20: print(k[0],k[2],round(m,1),'MB')
print('frame total', k[2], round(tot,1), 'MB')
PY
     rm -rf $OUT/pmc_$c
   done) 2>&1 | tee -a $OUT/ab.log
done
