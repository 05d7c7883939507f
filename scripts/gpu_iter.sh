#!/bin/bash
# Iteration session: local_laplacian parity, bench, env A/B.  Usage: bash scripts/gpu_iter.sh tag ["VAR=val" ...]
TAG=${1:-it}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest local_laplacian"; timeout 900 python -m pytest tests/test_local_laplacian.py -m gpu -q -x --tb=short 2>&1 | tail -15 | tee $OUT/pytest_ll.log
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench.json
[ $# -gt 0 ] && bash scripts/gpu_abl.sh $TAG "$@"
exit 0
