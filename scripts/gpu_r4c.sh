#!/bin/bash
# quick iteration: parity (local_laplacian tests + fuzz slice) then kernel times and frame rate.  Usage: bash scripts/gpu_r4c.sh tag [lib-suffix ...]
TAG=${1:-r4c}; shift
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest local_laplacian"; timeout 1200 python -m pytest tests/test_local_laplacian.py tests/test_fuzz_slice.py -m gpu -q -x --tb=short 2>&1 | tail -8 | tee $OUT/pytest_ll.log
for v in "" "$@"; do
  echo "== variant '$v'"; HLMI_LIB=$R/halide_amd/lib/libhlmi$v.so timeout 300 python scripts/kernel_times.py 2>&1 | grep -E "ll_down01|ll_up0|sum" | tee -a $OUT/variants.log
  HLMI_LIB=$R/halide_amd/lib/libhlmi$v.so timeout 300 python scripts/frame_bench.py 8 4 2>&1 | tail -2 | tee -a $OUT/variants.log
done
