R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/conv; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_layer.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
timeout 300 python bench_apps.py --only conv_layer_bf16 2>/dev/null | grep pipeline | tee $OUT/bench.json
PMC_CMD="python bench_apps.py --only conv_layer_bf16 --samples 1" bash scripts/gpu_pmc_cmd.sh conv/pmc "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" 2>&1 | grep -E "conv3x3" | tee $OUT/pmc.log
