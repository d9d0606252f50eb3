timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "small_cin" 2>&1 | grep -v '^$' | tail -3
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -k "golden" 2>&1 | grep -v '^$' | tail -2
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10"
for i in 1 2 3; do
U3D_TUNE=20:-1 $B 2>/dev/null | python -c "import sys,json; print('old ', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
$B 2>/dev/null | python -c "import sys,json; print('va  ', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done
