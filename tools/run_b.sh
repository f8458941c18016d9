python -m pytest tests/test_gpu_kernels.py tests/test_gpu_anerf.py tests/test_gpu_configs.py tests/test_gpu_modules.py tests/test_gpu_entry_points.py -q 2>&1 | tail -8
