#!/bin/bash
# Other pipelines: parity tests of the named files + bench_apps.  Usage: bash scripts/gpu_apps.sh tag [pytest files...]
TAG=${1:-apps}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
[ $# -gt 0 ] && { echo "== pytest $@"; timeout 1200 python -m pytest "$@" -m gpu -q -x --tb=short 2>&1 | tail -15 | tee $OUT/pytest.log; }
echo "== bench_apps"; timeout 900 python bench_apps.py 2>&1 | tee $OUT/bench_apps.jsonl
exit 0
