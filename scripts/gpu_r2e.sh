#!/bin/bash
# round 2, call E: ablations of ll_down01f (what bounds it) + SQ counters of the two big kernels
TAG=${1:-r2e}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for v in "" _abl1 _abl2 _abl4 _abl8 _abl6 _abl15; do
  echo "== kernel times lib$v"; HLMI_LIB=$R/halide_amd/lib/libhlmi$v.so timeout 120 python scripts/kernel_times.py 2>&1 | grep -E "down01|up0" | tee -a $OUT/abl.log
done
PMC_CMD="python scripts/ll_once.py 3" bash scripts/gpu_pmc_cmd.sh $TAG/pmc "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_I8 SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS" 2>&1 | grep -E "down01|up0f|^\(" | tee $OUT/pmc.log
