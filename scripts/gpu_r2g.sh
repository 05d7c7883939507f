#!/bin/bash
TAG=${1:-r2g}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_local_laplacian.py -m gpu -q --tb=short -p no:cacheprovider --maxfail=5 2>&1 | tail -8 | tee $OUT/pytest_ll.log
echo "== kernel times"; timeout 120 python scripts/kernel_times.py 2>&1 | tail -14 | tee $OUT/kt.log
echo "== kernel times level in SGPRs"; HLMI_LIB=$R/halide_amd/lib/libhlmi_lsgpr.so timeout 120 python scripts/kernel_times.py 2>&1 | grep -E "down01|up0" | tee -a $OUT/kt.log
for ru in 8 32; do echo "== RU=$ru"; HLMI_LL_RU=$ru timeout 120 python scripts/kernel_times.py 2>&1 | grep -E "up0" | tee -a $OUT/kt.log; done
echo "== frame_bench"; timeout 200 python scripts/frame_bench.py 2>&1 | tail -4 | tee $OUT/fb.log
