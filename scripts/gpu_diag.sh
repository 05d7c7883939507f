#!/bin/bash
# Diagnostics session: VALU ubench, baseline bench, ablations, SQ counters.  Usage: bash scripts/gpu_diag.sh tag
TAG=${1:-diag}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== ubench"; (cd scripts/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o /tmp/valu_rate 2>&1 | tail -2; timeout 120 /tmp/valu_rate) | tee $OUT/ubench.log
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench.json
bash scripts/gpu_abl.sh $TAG HLMI_LL_D0F=0 HLMI_LL_UP0_OLD=1 HLMI_LL_FUSE_FROM=8
bash scripts/gpu_pmc.sh $TAG "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" 2>&1 | tee $OUT/pmc.log
