#!/bin/bash
# failing tests of r3a in detail + the new bilateral histogram
TAG=${1:-r3b}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_threads.py tests/test_torch_ops.py tests/test_unsharp.py -m gpu -q --tb=long 2>&1 | tail -120 | tee $OUT/pytest_a.log
timeout 600 python -m pytest tests/test_local_laplacian.py tests/test_bilateral_grid.py tests/test_fuzz_slice.py -m gpu -q --tb=short 2>&1 | tail -30 | tee $OUT/pytest_b.log
timeout 300 python bench_apps.py --only bilateral_grid 2>/dev/null | grep pipeline | tee $OUT/bench_bg.jsonl
