#!/bin/bash
# Round 4: where does ll_down01e's time go?  Compile-time ablations (libhlmi_e*.so, HLMI_D01E_ABL) + PMC counters of the shipped build.
TAG=${1:-r4b}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for v in "" _e1 _e2 _e8 _e16 _e24; do
  echo "== variant '$v'"; HLMI_LIB=$R/halide_amd/lib/libhlmi$v.so timeout 300 python scripts/kernel_times.py 2>&1 | grep -E "ll_down01|ll_up0|sum" | tee -a $OUT/variants.log
  HLMI_LIB=$R/halide_amd/lib/libhlmi$v.so timeout 300 python scripts/frame_bench.py 8 4 2>&1 | tail -2 | tee -a $OUT/variants.log
done
bash scripts/gpu_pmc.sh $TAG "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR" 2>&1 | grep -E "ll_down01|ll_up0" | tee $OUT/pmc.log
