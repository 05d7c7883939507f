#!/bin/bash
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_local_laplacian.py -m gpu -q --tb=short -p no:cacheprovider --maxfail=5 2>&1 | tail -12
echo "== kernel times half"; timeout 120 python scripts/kernel_times.py 2>&1 | grep -E "down01|sum"
echo "== kernel times full-width"; HLMI_LL_D01H=0 timeout 120 python scripts/kernel_times.py 2>&1 | grep -E "down01|sum"
for u in 2048 3072 6144; do echo "== half UNITS0H=$u"; HLMI_LL_UNITS0H=$u timeout 120 python scripts/kernel_times.py 2>&1 | grep -E "down01"; done
fb() { echo "== frame_bench $1"; shift; env "$@" timeout 200 python scripts/frame_bench.py 8 4 2>&1 | tail -2; }
fb "half" A=1; fb "full" HLMI_LL_D01H=0; fb "half" A=1; fb "full" HLMI_LL_D01H=0
