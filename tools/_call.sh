timeout 900 python -m pytest tests/test_gpu_plus1.py tests/test_gpu_kernels.py -x -q -k "nobody_wrote or splitk or small_cin or variants" 2>&1 | grep -v '^$' | tail -5
B="python bench.py --no-cpu-baseline --no-extras"
$B 2>&1 | tail -1 | cut -c1-500
