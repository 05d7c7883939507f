#!/bin/bash
# Round evidence: all GPU parity tests, smoke, bench (with cpu baseline), rocprofv3 kernel stats + PMC traffic.
# Usage: bash scripts/gpu_full.sh tag
TAG=${1:-full}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "Marketing Name" > $OUT/rocminfo.txt; nproc >> $OUT/rocminfo.txt; lscpu | grep -m1 "Model name" >> $OUT/rocminfo.txt
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --tb=short --durations=8 2>&1 | tail -30 | tee $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -1 | tee $OUT/bench.json | cut -c1-600
echo "== bench with the driver's flags"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-variants 2>&1 | tail -1 | tee $OUT/bench_driver_flags.json | cut -c1-300
echo "== bench_apps"; timeout 900 python bench_apps.py 2>/dev/null | grep pipeline | tee $OUT/bench_apps.jsonl | cut -c1-160
echo "== bench 1 stream"; timeout 900 python bench.py --partitions 0 --streams 1 --no-cpu-baseline --no-other-configs 2>&1 | tail -1 | tee $OUT/bench_1stream.json
echo "== bench --gpus 2 on this box (must refuse unless 2 devices are visible)"; timeout 300 python bench.py --gpus 2 --steps 2 > $OUT/bench_gpus2.log 2>&1; echo "exit $?" >> $OUT/bench_gpus2.log; tail -2 $OUT/bench_gpus2.log
cd /tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-other-configs --no-ceiling --partitions 0 --streams 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- $CMD > $OUT/pmc_$c.log 2>&1
done
# the same counters with the launch geometry of a frame queue (HLMI_STREAM_SHARE=4: the caller's one stream is sized like one of four
# queues — one ll_down01e workgroup per CU, 640 units, 32 rows per ll_up0h wave, non-temporal frame accesses): what the headline loop moves
for c in FETCH_SIZE WRITE_SIZE; do
  HLMI_STREAM_SHARE=4 timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmcq_$c -o pmc -- $CMD > $OUT/pmcq_$c.log 2>&1
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.csv" -size +3M -delete
ls -la $OUT $OUT/kt | head -30
