#!/bin/bash
# round 2, call D: parity of ll_down01f with LDS seam exchange, then timings
TAG=${1:-r2d}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity (local_laplacian tests)"; timeout 900 python -m pytest tests/test_local_laplacian.py -m gpu -q --tb=short -p no:cacheprovider --maxfail=5 2>&1 | tail -25 | tee $OUT/pytest_ll.log
echo "== kernel times exch"; timeout 120 python scripts/kernel_times.py 2>&1 | tail -14 | tee $OUT/kt_exch.log
echo "== kernel times no exch"; HLMI_LL_D01_EXCH=0 timeout 120 python scripts/kernel_times.py 2>&1 | grep down01 | tee $OUT/kt_noexch.log
fb() { echo "== frame_bench $1"; shift; env "$@" timeout 200 python scripts/frame_bench.py 2>&1 | tail -2; }
fb "exch default" A=1 | tee $OUT/fb.log
fb "no exch" HLMI_LL_D01_EXCH=0 | tee -a $OUT/fb.log
for u in 1024 1536 2560 3072; do fb "exch UNITS0=$u" HLMI_LL_UNITS0=$u | tee -a $OUT/fb.log; done
fb "unfused" HLMI_LL_FUSE_D2=0 | tee -a $OUT/fb.log
fb "exch default again" A=1 | tee -a $OUT/fb.log
