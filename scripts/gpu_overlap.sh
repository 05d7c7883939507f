#!/bin/bash
# How much do kernels of the two streams actually overlap?  rocprofv3 kernel trace of the default bench, summarised on the box.
TAG=${1:-ovl}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r['Kernel_Name'].find('ll_') >= 0]
ev = []
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    ev.append((s, 1, r)); ev.append((e, -1, r))
ev.sort(key=lambda t: (t[0], t[1]))
# take the last 60 % of the trace (timed region), measure time with 0 / 1 / >=2 kernels in flight
t0 = ev[int(len(ev) * 0.4)][0]
depth, last, hist = 0, None, collections.Counter()
for t, d, r in ev:
    if last is not None and t > t0:
        hist[min(depth, 2)] += t - max(last, t0)
    depth += d; last = t
tot = sum(hist.values())
print({k: round(v / tot, 3) for k, v in sorted(hist.items())}, "total_us", tot / 1e3)
# per kernel: mean duration alone vs in the 2-stream run
dur = collections.defaultdict(list)
for r in rows:
    if int(r['Start_Timestamp']) > t0:
        dur[r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0][:28] + ':' + r['Grid_Size']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(k, len(v), round(sum(v) / len(v) / 1e3, 1))
PY
find $OUT -name "*.csv" -size +1M -delete
