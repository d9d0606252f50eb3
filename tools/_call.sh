python tools/_ab_reduce.py 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "wgrad" 2>&1 | grep -E "passed|failed|rror" | tail -2
ms() { python -c "import sys,json; print('$1', json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])['ms_per_step'])"; }
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10"
for i in 1 2 3; do
U3D_TUNE=23:2 $B 2>/dev/null | ms "4-byte "
$B 2>/dev/null | ms "16-byte"
done
