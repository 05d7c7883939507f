#!/bin/bash
# A/B of a compile-time variant of the library on local_laplacian: bash scripts/gpu_ll_variant_ab.sh <tag> <variant lib basename> [pytest -k]
TAG=$1; VAR=$2; K=${3:-local_laplacian}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -k "$K" 2>&1 | tail -4 | tee $OUT/pytest.log
for rnd in 1 2 3; do
  for kind in noise smooth; do
    LL_AB_KERNELS=0 LL_AB_KIND=$kind LL_AB_ROUNDS=1 timeout 300 python scripts/ll_ab.py 2>&1 | grep default | sed "s/^/new   /" | tee -a $OUT/ab.txt
    HLMI_LIB=$R/halide_amd/lib/$VAR LL_AB_KERNELS=0 LL_AB_KIND=$kind LL_AB_ROUNDS=1 timeout 300 python scripts/ll_ab.py 2>&1 | grep default | sed "s/^/old   /" | tee -a $OUT/ab.txt
  done
done
PMC_CMD="python scripts/ll_once.py" bash scripts/gpu_pmc_cmd.sh $TAG/pmc "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" 2>&1 | grep ll_down01e | tee $OUT/pmc.txt
