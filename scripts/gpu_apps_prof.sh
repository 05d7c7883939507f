#!/bin/bash
# rocprofv3 evidence for the non-headline pipelines: kernel stats of one bench_apps.py pass + counters for nlm_7x7, conv3x3_bf16_lin, bg_*
TAG=${1:-apps_prof}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- bash -c "cd $R && python bench_apps.py --samples 1" > $OUT/kt.log 2>&1)
find $OUT -name "*kernel_trace.csv" -delete
PMC_CMD="python bench_apps.py --only nl_means,bilateral_grid,conv_layer_bf16,conv_layer --samples 1" bash scripts/gpu_pmc_cmd.sh $TAG/pmc \
  "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" \
  "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE" 2>&1 | grep -E "^\(" | tee $OUT/pmc.log
