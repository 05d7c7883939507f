#!/bin/bash
# Round 4 iteration: parity of the re-cut local_laplacian dataflow (ll_down01e / ll_up0h) + A/B bench against the round-3 pair.
# Usage: bash scripts/gpu_r4a.sh tag
TAG=${1:-r4a}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest local_laplacian"; timeout 1200 python -m pytest tests/test_local_laplacian.py tests/test_fuzz_slice.py -m gpu -q -x --tb=short 2>&1 | tail -25 | tee $OUT/pytest_ll.log
for E in 1 0; do
  echo "== bench EMIT=$E 4 partitions"; HLMI_LL_EMIT=$E timeout 600 python bench.py --steps 10 --no-cpu-baseline --no-other-configs --no-ceiling 2>&1 | tail -1 | tee $OUT/bench_emit$E.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'], d['config']['variants']); print(d['roofline']['kernel_ms_per_frame'])"
  echo "== bench EMIT=$E 1 stream"; HLMI_LL_EMIT=$E timeout 600 python bench.py --steps 10 --partitions 0 --streams 1 --no-cpu-baseline --no-other-configs --no-ceiling --no-variants 2>&1 | tail -1 | tee $OUT/bench1_emit$E.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['config']['frame_ms'])"
done
