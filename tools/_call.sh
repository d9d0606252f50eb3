timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "small_cin" 2>&1 | grep -v '^$' | tail -3
