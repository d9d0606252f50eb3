timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | grep -v '^$' | tail -3
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_plus1.py -x -q -k "golden or replica or plus or subpixel" 2>&1 | grep -v '^$' | tail -2
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10"
for i in 1 2 3 4; do
U3D_STAT_REPS=1 $B 2>/dev/null | python -c "import sys,json; print('reps1 ', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
$B 2>/dev/null | python -c "import sys,json; print('reps8 ', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done
