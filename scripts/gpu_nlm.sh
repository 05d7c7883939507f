timeout 600 python -m pytest tests/test_nl_means.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
for th in 16 32; do echo "== TH=$th"; HLMI_NLM_TH=$th timeout 300 python bench_apps.py --only nl_means 2>/dev/null | grep -o '"ms_per_call": [0-9.]*'; done
