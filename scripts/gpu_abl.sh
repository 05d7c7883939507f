#!/bin/bash
# A/B timings of run-time switches (env vars).  Usage: bash scripts/gpu_abl.sh tag "VAR=val ..." ...
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for v in "$@"; do
  (export $v; timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('$v', d['value'], d['config']['frame_ms'], d['roofline']['kernel_ms_per_frame'])
except Exception as e: print('$v', 'FAILED', l[-300:])
") | tee -a $OUT/abl.log
done
