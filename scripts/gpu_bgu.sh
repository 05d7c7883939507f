#!/bin/bash
# bgu: parity, drop-in driver, RunGen, bench line
timeout 900 python -m pytest tests/test_bgu.py tests/test_iir_blur.py tests/test_entry_protocol.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -12
timeout 600 python -m pytest tests/test_dropin_drivers.py tests/test_reference_consumers.py -m gpu -q --tb=short -p no:cacheprovider -k "bgu" 2>&1 | tail -8
timeout 300 python bench_apps.py --only bgu,iir_blur 2>/dev/null | grep pipeline | cut -c1-600
HLMI_BGU_DIRECT=1 timeout 300 python bench_apps.py --only bgu 2>/dev/null | grep -o '"pipeline": "[a-z_]*".*"ms_per_call": [0-9.]*'
