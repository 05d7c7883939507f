#!/bin/bash
# local_laplacian session: parity first, then bench + A/B of the run-time knobs.  Usage: bash scripts/gpu_ll.sh [tag]
TAG=${1:-ll}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest local_laplacian"; timeout 900 python -m pytest tests/test_local_laplacian.py -m gpu -q --maxfail=6 --tb=short 2>&1 | tail -60 | tee $OUT/pytest_ll.log
echo "== bench (default knobs)"; timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -3 | tee $OUT/bench.log
for v in "HLMI_LL_TY0=4" "HLMI_LL_TY0=16" "HLMI_LL_RU=4" "HLMI_LL_RU=16" "HLMI_LL_NO_DPP=1" "HLMI_LL_TYB=8"; do
  echo "== bench $v"; (export $v; timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$v', d['value'], d['config']['frame_ms'], d['roofline']['kernel_ms_per_frame'])
except Exception as e: print('$v', 'FAILED', l[-300:])
") | tee -a $OUT/variants.log
done
