#!/bin/bash
# One self-contained GPU session: parity tests, smoke, bench, rocprofv3 kernel stats.
# Usage (from the repo root on the GPU box): bash scripts/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx9" > $OUT/rocminfo.txt
nproc > $OUT/nproc.txt; lscpu | grep -m1 "Model name" >> $OUT/nproc.txt
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -45 | tee $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.log
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 3 2>&1 | tail -5 | tee $OUT/bench.log
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o ll -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1)
find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats.csv"); do head -30 $f; done
