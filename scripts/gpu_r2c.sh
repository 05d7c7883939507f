#!/bin/bash
# round 2, call C: parity of the fused level 0->1->2 kernel, then A/B: fusion on/off, unit counts, non-temporal variants
TAG=${1:-r2c}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity (local_laplacian tests)"; timeout 600 python -m pytest tests/test_local_laplacian.py tests/test_torch_ops.py tests/test_batch.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -25 | tee $OUT/pytest_ll.log
echo "== kernel times fused"; timeout 120 python scripts/kernel_times.py 2>&1 | tail -14 | tee $OUT/kt_fused.log
echo "== kernel times unfused"; HLMI_LL_FUSE_D2=0 timeout 120 python scripts/kernel_times.py 2>&1 | tail -14 | tee $OUT/kt_unfused.log
fb() { echo "== frame_bench $1"; shift; env "$@" timeout 200 python scripts/frame_bench.py 2>&1 | tail -2; }
fb "fused default" A=1 | tee $OUT/fb.log
fb "unfused" HLMI_LL_FUSE_D2=0 | tee -a $OUT/fb.log
for u in 1024 1536 3072 4096; do fb "fused UNITS0=$u" HLMI_LL_UNITS0=$u | tee -a $OUT/fb.log; done
for v in 2 4 6 7; do fb "fused nt$v" HLMI_LIB=$R/halide_amd/lib/libhlmi_nt$v.so | tee -a $OUT/fb.log; done
fb "fused default again" A=1 | tee -a $OUT/fb.log
