timeout 600 python -m pytest tests/test_camera_pipe.py tests/test_bilateral_grid.py tests/test_dropin_drivers.py tests/test_reference_consumers.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -6
timeout 300 python bench_apps.py --only bilateral_grid,camera_pipe 2>/dev/null | grep pipeline | cut -c1-330
HLMI_BG_UNFUSED=1 HLMI_CP_NO_SETUP_CACHE=1 timeout 300 python bench_apps.py --only bilateral_grid,camera_pipe 2>/dev/null | grep -o '"pipeline": "[a-z_]*".*"ms_per_call": [0-9.]*'
