#!/bin/bash
# round 2, call A: thread diagnosis first (bounded), then the whole GPU suite
TAG=${1:-r2a}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== thread diag"; timeout 300 python scripts/thread_diag.py 2>&1 | tail -40 | tee $OUT/thread_diag.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 --tb=short --durations=12 -p no:cacheprovider 2>&1 | tail -150 | tee $OUT/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.log
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench.json
