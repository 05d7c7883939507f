#!/bin/bash
# Round 3, first GPU pass: parity suite, the new bench line (graph on / off), the membench sweep, counter list, FETCH_SIZE calibration.
TAG=${1:-r3a}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "Marketing Name" > $OUT/rocminfo.txt; nproc >> $OUT/rocminfo.txt; lscpu | grep -m1 "Model name" >> $OUT/rocminfo.txt
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 --tb=short --durations=6 2>&1 | tail -40 | tee $OUT/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py > $OUT/bench.out 2>&1; tail -1 $OUT/bench.out | tee $OUT/bench.json | cut -c1-1500
echo "== bench graph off"; HLMI_LL_GRAPH=0 timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-variants 2>&1 | tail -1 | tee $OUT/bench_nograph.json | cut -c1-400
echo "== bench graph on (same flags)"; timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-variants 2>&1 | tail -1 | tee $OUT/bench_graph.json | cut -c1-400
echo "== 1 stream graph on/off"
timeout 600 python bench.py --partitions 0 --streams 1 --no-cpu-baseline --no-other-configs --no-variants 2>&1 | tail -1 | tee $OUT/bench_1stream.json | cut -c1-300
HLMI_LL_GRAPH=0 timeout 600 python bench.py --partitions 0 --streams 1 --no-cpu-baseline --no-other-configs --no-variants 2>&1 | tail -1 | tee $OUT/bench_1stream_nograph.json | cut -c1-300
echo "== membench sweep"
timeout 600 python - <<'PY' 2>&1 | tee $OUT/membench.log
import json, os, halide_amd as hl
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", os.environ.get("TAG", "r3a"))
for nb in (1 << 30, 1 << 28):
    sw = hl.membench_sweep(nb, 10)
    json.dump(sw, open(os.path.join(out, f"membench_sweep_{nb >> 20}MB.json"), "w"))
    print(nb >> 20, "MB best:", sw["best"], "memcpy_d2d", sw["memcpy_d2d_gbs"])
    top = sorted(sw["copy"], key=lambda r: -r["gbs"])[:8]
    print("  top copy:", top)
print("naive:", hl.membench_naive(1 << 30, 10))
print("widths:", hl.membench_widths(1 << 30, 4))
PY
echo "== counters"
(cd /tmp && timeout 120 rocprofv3 --list-avail > $OUT/avail_full.txt 2>&1); grep -o "TCC_[A-Za-z0-9_]*" $OUT/avail_full.txt | sort -u | tr '\n' ' ' | cut -c1-3000
echo
W="python -c \"import halide_amd as hl; print(hl.membench_widths(1<<30, 2))\""
PMC_CMD="$W" bash scripts/gpu_pmc_cmd.sh $TAG/pmc_w "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" 2>&1 | grep -E "mb_width|Error|error" | tee $OUT/pmc_widths.log
PMC_CMD="python scripts/ll_once.py 3" bash scripts/gpu_pmc_cmd.sh $TAG/pmc_ll "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" 2>&1 | grep -E "ll_|Error|error" | tee $OUT/pmc_ll.log
find $OUT -name "*.csv" -size +2M -delete
ls $OUT
