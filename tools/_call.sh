timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_plus1.py -x -q -k "subpixel or plus" 2>&1 | grep -v '^$' | tail -3
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 60 --warmup 10"
for i in 1 2 3; do
U3D_TUNE=22:1 $B 2>/dev/null | python -c "import sys,json; print('old  ', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
$B 2>/dev/null | python -c "import sys,json; print('once ', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done
