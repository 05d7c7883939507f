#!/bin/bash
# Round 6: the two canonical float forms side by side on one box.  Parity of both builds (libhlmi.so = fma canon,
# libhlmi_nofma.so = -DHLMI_CANON_FMA=0) against the matching oracle form, then A/B timings alternating the two libraries.
# Usage: bash scripts/gpu_canon_ab.sh <tag> [pytest -k expression]
TAG=${1:-r06ab}; KEXPR=${2:-"local_laplacian or nl_means or bilateral"}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
NOFMA=$R/halide_amd/lib/libhlmi_nofma.so
echo "== parity, fma canon (default library)"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --tb=short -k "$KEXPR" 2>&1 | tail -25 | tee $OUT/pytest_fma.log
echo "== parity, canon 0 (libhlmi_nofma.so)"
HLMI_LIB=$NOFMA timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --tb=short -k "$KEXPR" 2>&1 | tail -25 | tee $OUT/pytest_nofma.log
echo "== local_laplacian A/B (us per 4K frame), alternating libraries"
for rnd in 1 2; do
  for kind in noise smooth; do
    LL_AB_KIND=$kind LL_AB_ROUNDS=1 timeout 300 python scripts/ll_ab.py 2>&1 | sed "s/^/fma   /" | tee -a $OUT/ll_ab.txt
    HLMI_LIB=$NOFMA LL_AB_KIND=$kind LL_AB_ROUNDS=1 timeout 300 python scripts/ll_ab.py 2>&1 | sed "s/^/nofma /" | tee -a $OUT/ll_ab.txt
  done
done
echo "== apps A/B"
for rnd in 1 2; do
  timeout 600 python bench_apps.py --only nl_means,bilateral_grid --samples 10 2>&1 | sed "s/^/fma   /" | tee -a $OUT/apps_ab.txt | cut -c1-400
  HLMI_LIB=$NOFMA timeout 600 python bench_apps.py --only nl_means,bilateral_grid --samples 10 2>&1 | sed "s/^/nofma /" | tee -a $OUT/apps_ab.txt | cut -c1-400
done
