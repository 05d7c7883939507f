#!/usr/bin/env python3
"""Which kernels run at the same time?  Reads a rocprofv3 kernel-trace CSV (Start_Timestamp / End_Timestamp / Queue_Id /
Kernel_Name) and prints, for the steady part of the trace: the share of wall time with k kernels in flight, the share of wall
time each COMBINATION of kernel kinds is in flight, and per kind its mean duration alone in the trace vs here.

    python scripts/timeline_overlap.py <kernel_trace.csv>"""
import collections
import csv
import sys


def kind(name):
    for k in ("ll_down01e", "ll_up0h", "ll_down_strip2", "ll_down_multi", "ll_up_multi", "ll_remap_lut"):
        if k in name:
            return {"ll_down01e": "D", "ll_up0h": "U", "ll_down_strip2": "s", "ll_down_multi": "m", "ll_up_multi": "u", "ll_remap_lut": "l"}[k]
    return None


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = kind(r["Kernel_Name"])
        if k is None:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id", "?")))
rows.sort()
if not rows:
    sys.exit("no local_laplacian kernels in the trace")
# steady part: the last 60 % of the dispatches
rows = rows[int(len(rows) * 0.4):]
t0, t1 = rows[0][0], max(r[1] for r in rows)
events = []
for s, e, k, q in rows:
    events.append((s, 1, k))
    events.append((e, -1, k))
events.sort()
live = collections.Counter()
by_n = collections.Counter()
by_combo = collections.Counter()
prev = events[0][0]
for t, d, k in events:
    dt = t - prev
    if dt > 0:
        n = sum(live.values())
        by_n[n] += dt
        combo = "".join(sorted(k2 * c for k2, c in live.items() if c > 0))
        by_combo[combo] += dt
    live[k] += d
    prev = t
total = sum(by_n.values())
print(f"steady window {total / 1e3:.1f} us, {len(rows)} dispatches, queues: {sorted(set(r[3] for r in rows))}")
print("kernels in flight -> share of wall time:", {n: round(v / total, 3) for n, v in sorted(by_n.items())})
print("combination (D = ll_down01e, U = ll_up0h, s / m / u = strip2 / down_multi / up_multi) -> share of wall time:")
for c, v in sorted(by_combo.items(), key=lambda x: -x[1])[:24]:
    print(f"  {c or '(idle)':10s} {v / total:6.3f}")
dur = collections.defaultdict(list)
for s, e, k, q in rows:
    dur[k].append(e - s)
print("mean duration here (us):", {k: round(sum(v) / len(v) / 1e3, 1) for k, v in dur.items()})
nD = len(dur["D"])
print(f"frames: {nD}, wall per frame {total / 1e3 / max(nD, 1):.1f} us")
