#!/bin/bash
# conv_layer_bf16 (new kernel), stencil_chain (register window), bilateral_grid (histogram): parity + timings + kernel stats
TAG=${1:-r3c}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_layer.py tests/test_stencil_chain.py tests/test_bilateral_grid.py tests/test_fuzz_slice.py tests/test_dropin_drivers.py -m gpu -q --tb=short -x 2>&1 | tail -30 | tee $OUT/pytest.log
ONLY=bilateral_grid,stencil_chain,conv_layer_bf16
timeout 300 python bench_apps.py --only $ONLY 2>/dev/null | grep pipeline | tee $OUT/bench_new.jsonl
HLMI_CONV_OLD=1 HLMI_SC_LDS=1 timeout 300 python bench_apps.py --only stencil_chain,conv_layer_bf16 2>/dev/null | grep pipeline | tee $OUT/bench_old.jsonl
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $R/bench_apps.py --only $ONLY > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-4,6-7 "$f" | head -14
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.csv" -size +2M -delete
