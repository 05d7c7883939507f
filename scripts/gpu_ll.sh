#!/bin/bash
# local_laplacian session: parity first, then bench + A/B of run-time knobs.
# Usage: bash scripts/gpu_ll.sh tag ["VAR=val" ...]
TAG=${1:-ll}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest local_laplacian"; timeout 900 python -m pytest tests/test_local_laplacian.py -m gpu -q --maxfail=6 --tb=short 2>&1 | tail -40 | tee $OUT/pytest_ll.log
echo "== bench (default knobs)"; timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench.log
bash scripts/gpu_abl.sh $TAG "$@"
