#!/bin/bash
# Randomised GPU-vs-oracle soak at the round's final kernel sources (scripts/fuzz_parity.py): the headline pipeline biased to its fast
# path, the pipelines whose kernels changed this round, every pipeline for a few seconds, and the 8-host-thread stress.
# Usage: bash scripts/gpu_soak.sh tag
TAG=${1:-soak}; OUT=gpurun_out/$TAG; mkdir -p $OUT
{
  echo "Round 6 fuzz / stress runs on MI355X (scripts/fuzz_parity.py; library at $(cat $GRAFT_REPO_ROOT/.soak_head 2>/dev/null))."
  echo; echo "== python scripts/fuzz_parity.py --only local_laplacian --seconds 150 --seed 51"
  timeout 400 python scripts/fuzz_parity.py --only local_laplacian --seconds 150 --seed 51 2>&1 | tail -3
  echo; echo "== python scripts/fuzz_parity.py --only local_laplacian --seconds 100 --seed 56 --frame-queue   (the throughput geometry: one ll_down01e workgroup per CU)"
  timeout 400 python scripts/fuzz_parity.py --only local_laplacian --seconds 100 --seed 56 --frame-queue 2>&1 | tail -3
  echo; echo "== python scripts/fuzz_parity.py --only bilateral_grid,depthwise_separable_conv --seconds 25 --seed 52"
  timeout 300 python scripts/fuzz_parity.py --only bilateral_grid,depthwise_separable_conv --seconds 25 --seed 52 2>&1 | tail -4
  echo; echo "== python scripts/fuzz_parity.py --seconds 5 --seed 53   (all 17 pipelines)"
  timeout 600 python scripts/fuzz_parity.py --seconds 5 --seed 53 2>&1 | tail -19
  echo; echo "== python scripts/fuzz_parity.py --threads 8 --seconds 90 --seed 54   (8 host threads, every second one on a stream of its own)"
  timeout 400 python scripts/fuzz_parity.py --threads 8 --seconds 90 --seed 54 2>&1 | tail -19
} > $OUT/fuzz_parity.txt 2>&1
tail -50 $OUT/fuzz_parity.txt
if [ -f $GRAFT_REPO_ROOT/halide_amd/lib/libhlmi_nofma.so ]; then
  { echo; echo "== the same sweep against the canon-0 build: HLMI_LIB=libhlmi_nofma.so python scripts/fuzz_parity.py --seconds 4 --seed 55"
    HLMI_LIB=$GRAFT_REPO_ROOT/halide_amd/lib/libhlmi_nofma.so timeout 600 python scripts/fuzz_parity.py --seconds 4 --seed 55 2>&1 | tail -19; } >> $OUT/fuzz_parity.txt 2>&1
  tail -20 $OUT/fuzz_parity.txt
fi
