#!/bin/bash
# round 2, call B: GPU suite, then A/B of ll_up:1 fused into ll_up0f (HLMI_LL_FUSE_UP1) and non-temporal frame accesses (libhlmi_nt.so)
TAG=${1:-r2b}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q --maxfail=15 --tb=short --durations=8 -p no:cacheprovider -x 2>&1 | tail -60 | tee $OUT/pytest_gpu.log
echo "== frame_bench fused (default)"; timeout 200 python scripts/frame_bench.py 2>&1 | tail -4 | tee $OUT/fb_fused.log
echo "== frame_bench unfused"; HLMI_LL_FUSE_UP1=0 timeout 200 python scripts/frame_bench.py 2>&1 | tail -4 | tee $OUT/fb_unfused.log
echo "== frame_bench fused RU=8"; HLMI_LL_RU=8 timeout 200 python scripts/frame_bench.py 2>&1 | tail -4 | tee $OUT/fb_fused_ru8.log
echo "== frame_bench fused RU=32"; HLMI_LL_RU=32 timeout 200 python scripts/frame_bench.py 2>&1 | tail -4 | tee $OUT/fb_fused_ru32.log
echo "== frame_bench NT fused"; HLMI_LIB=$R/halide_amd/lib/libhlmi_nt.so timeout 200 python scripts/frame_bench.py 2>&1 | tail -4 | tee $OUT/fb_nt.log
echo "== frame_bench NT unfused"; HLMI_LL_FUSE_UP1=0 HLMI_LIB=$R/halide_amd/lib/libhlmi_nt.so timeout 200 python scripts/frame_bench.py 2>&1 | tail -4 | tee $OUT/fb_nt_unfused.log
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -1 | tee $OUT/bench.json
echo "== bench 1 stream"; timeout 600 python bench.py --no-cpu-baseline --no-variants --partitions 0 --streams 1 2>&1 | tail -1 | tee $OUT/bench_1stream.json
