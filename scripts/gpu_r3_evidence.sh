#!/bin/bash
# Round-3 evidence: gpu_full (tests, smoke, bench lines, kernel stats, PMC traffic) + apps profile + ceiling sweep + width calibration
TAG=${1:-r03a}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
bash scripts/gpu_full.sh $TAG
cd $R
echo "== apps: kernel stats + counters"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_apps -o kt -- bash -c "cd $R && python bench_apps.py --samples 1" > $OUT/kt_apps.log 2>&1)
find $OUT -name "*kernel_trace.csv" -delete
PMC_CMD="python bench_apps.py --only nl_means,bilateral_grid,conv_layer_bf16,stencil_chain --samples 1" bash scripts/gpu_pmc_cmd.sh $TAG/pmc_apps \
  "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE" 2>&1 | grep -E "^\(" | tee $OUT/apps_pmc.txt | cut -c1-200
echo "== HBM ceiling sweep + access-width calibration"
timeout 600 python - <<PY 2>&1 | tee $OUT/membench.log
import json, os, halide_amd as hl
out = "$OUT"
for nb in (1 << 30, 1 << 28):
    sw = hl.membench_sweep(nb, 10)
    json.dump(sw, open(os.path.join(out, f"membench_sweep_{nb >> 20}MB.json"), "w"))
    print(nb >> 20, "MB best:", sw["best"], "memcpy_d2d", sw["memcpy_d2d_gbs"])
print("naive:", hl.membench_naive(1 << 30, 10))
print("widths:", hl.membench_widths(1 << 30, 4))
PY
W="python -c \"import halide_amd as hl; print(hl.membench_widths(1<<30, 2))\""
PMC_CMD="$W" bash scripts/gpu_pmc_cmd.sh $TAG/pmc_w "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" 2>&1 | grep -E "mb_width" | tee $OUT/pmc_widths.txt
PMC_CMD="python scripts/ll_once.py 3" bash scripts/gpu_pmc_cmd.sh $TAG/pmc_ll "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" 2>&1 | grep -E "ll_" | tee $OUT/pmc_ll_tcc.txt
find $OUT -name "*.csv" -size +3M -delete
ls $OUT | head -60
