#!/bin/bash
# A/B of run-time switches, timing only.  Usage: bash scripts/gpu_r4h.sh tag "VAR=val VAR2=val" ...
TAG=$1; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for v in "$@"; do
  echo "== $v" | tee -a $OUT/ab.log
  (export $v; timeout 300 python scripts/kernel_times.py 2>&1 | grep -E "ll_down01|ll_up0|sum"; timeout 300 python scripts/frame_bench.py 8 4 2>&1 | tail -2) 2>&1 | tee -a $OUT/ab.log
done
