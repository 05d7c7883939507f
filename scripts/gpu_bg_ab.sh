cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06bg
timeout 900 python -m pytest tests/test_bilateral_grid.py tests/test_device_math.py -m gpu -q --tb=short 2>&1 | tail -15 | tee gpurun_out/r06bg/pytest.log
for i in 1 2; do
timeout 300 python bench_apps.py --only bilateral_grid --samples 20 2>&1 | grep pipeline | sed "s/^/one /" | tee -a gpurun_out/r06bg/ab.txt | cut -c1-420
HLMI_BG_TWO_LAUNCH=1 timeout 300 python bench_apps.py --only bilateral_grid --samples 20 2>&1 | grep pipeline | sed "s/^/two /" | tee -a gpurun_out/r06bg/ab.txt | cut -c1-420
done
