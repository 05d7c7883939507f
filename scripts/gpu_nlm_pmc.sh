#!/bin/bash
TAG=${1:-nlmpmc}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for v in w4 w8 lds; do
  export HLMI_NLM_ROWS8= HLMI_NLM_LDS=
  unset HLMI_NLM_ROWS8 HLMI_NLM_LDS
  [ $v = w8 ] && export HLMI_NLM_ROWS8=1
  [ $v = lds ] && export HLMI_NLM_LDS=1
  echo "== $v"
  PMC_CMD="python bench_apps.py --only nl_means --samples 1" bash scripts/gpu_pmc_cmd.sh $TAG/$v "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU" 2>&1 | grep -E "nlm"
done | tee $OUT/pmc.txt
