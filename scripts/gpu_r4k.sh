#!/bin/bash
TAG=${1:-r4k}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_threads.py -m gpu -q -x --tb=short 2>&1 | tail -5
timeout 900 python bench.py --steps 10 2>&1 | tail -1 | tee $OUT/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','value_uniform_noise','value_natural_tiled','ms_per_step')}); print(d['roofline']['frac'], d['roofline']['frac_moved'], d['roofline']['traffic_source']); print(d['config']['variants'])
for c in d['config']['other_configs']: print(c['pipeline'], c['ms_per_call'], c['roofline']['frac'], c.get('batched'), c.get('ms_per_call_filter_uncached'))
print(d['cpu_baseline'])"
