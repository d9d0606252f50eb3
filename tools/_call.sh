timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad or replica or groupnorm" 2>&1 | grep -v '^$' | tail -2
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -k golden 2>&1 | grep -v '^$' | tail -2
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10"
for i in 1 2 3; do $B 2>/dev/null | python -c "import sys,json; print('job0 ', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; done
