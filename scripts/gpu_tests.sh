#!/bin/bash
# GPU parity tests only. Usage: bash scripts/gpu_tests.sh [tag] [pytest args...]
TAG=${1:-t}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 2000 python -m pytest tests -m gpu -q --maxfail=10 --tb=short --durations=15 "$@" 2>&1 | tail -120 | tee $OUT/pytest_gpu.log
