#!/bin/bash
# Round evidence: gpu_full (tests, smoke, bench lines, kernel stats, PMC traffic) + apps profile + ceiling sweep; `python scripts/make_profiles.py <tag> <round>`
# turns the directory into the tracked summaries under profiles/ (and stamps profiles/traffic.json with the kernel source's hash)
TAG=${1:-r05a}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
bash scripts/gpu_full.sh $TAG
cd $R
echo "== apps: kernel stats + counters"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_apps -o kt -- bash -c "cd $R && python bench_apps.py --samples 1 --no-batched" > $OUT/kt_apps.log 2>&1)
find $OUT -name "*kernel_trace.csv" -delete
PMC_CMD="python bench_apps.py --only nl_means,bilateral_grid,conv_layer_bf16,stencil_chain,camera_pipe --samples 1 --no-batched" bash scripts/gpu_pmc_cmd.sh $TAG/pmc_apps \
  "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE" 2>&1 | grep -E "^\(" | tee $OUT/apps_pmc.txt | cut -c1-200
echo "== local_laplacian: instruction counters of the two big kernels (both canonical forms)"
LLCMD="python scripts/ll_once.py"
PMC_CMD="$LLCMD" bash scripts/gpu_pmc_cmd.sh $TAG/pmc_ll "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" 2>&1 | grep -E "^\(" | tee $OUT/ll_pmc_fma.txt | cut -c1-200
if [ -f $R/halide_amd/lib/libhlmi_nofma.so ]; then
  PMC_CMD="HLMI_LIB=$R/halide_amd/lib/libhlmi_nofma.so $LLCMD" bash scripts/gpu_pmc_cmd.sh $TAG/pmc_ll_nofma "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" 2>&1 | grep -E "^\(" | tee $OUT/ll_pmc_nofma.txt | cut -c1-200
  echo "== full GPU suite against the canon-0 build"
  HLMI_LIB=$R/halide_amd/lib/libhlmi_nofma.so timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --tb=short 2>&1 | tail -8 | tee $OUT/pytest_gpu_nofma.log
fi
echo "== HBM ceiling sweep + access-width calibration"
timeout 600 python - <<PY 2>&1 | tee $OUT/membench.log
import json, os, halide_amd as hl
out = "$OUT"
for nb in (1 << 30, 1 << 28):
    sw = hl.membench_sweep(nb, 10)
    json.dump(sw, open(os.path.join(out, f"membench_sweep_{nb >> 20}MB.json"), "w"))
    print(nb >> 20, "MB best:", sw["best"], "memcpy_d2d", sw["memcpy_d2d_gbs"])
print("naive:", hl.membench_naive(1 << 30, 10))
PY
find $OUT -name "*.csv" -size +3M -delete
ls $OUT | head -60
