#!/bin/bash
timeout 900 python -m pytest tests/test_nl_means.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -5
for th in 32 16 64; do echo "packed TH=$th"; HLMI_NLM_TH=$th timeout 300 python bench_apps.py --only nl_means 2>/dev/null | grep -o '"ms_per_call": [0-9.]*'; done
echo scalar; HLMI_NLM_SCALAR=1 timeout 300 python bench_apps.py --only nl_means 2>/dev/null | grep -o '"ms_per_call": [0-9.]*'
