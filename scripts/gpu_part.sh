#!/bin/bash
export TMPDIR=/tmp
echo "== parity"; timeout 600 python -m pytest tests/test_local_laplacian.py tests/test_threads.py tests/test_batch.py tests/test_torch_ops.py -m gpu -q --tb=short -p no:cacheprovider --maxfail=5 2>&1 | tail -3
fb() { echo "== frame_bench $1"; shift; env "$@" timeout 200 python scripts/frame_bench.py 8 4 2>&1 | tail -2; }
fb "lut cache" A=1; fb "no lut cache" HLMI_LL_NO_LUT_CACHE=1; fb "lut cache" A=1; fb "no lut cache" HLMI_LL_NO_LUT_CACHE=1
