timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | grep -v '^$' | tail -2
L="--layers enc0.c2,enc1.c1,dec2.c2 --only fwd,dgrad --iters 10"
echo BASE; U3D_LIB_PATH=/root/repo/ab_libs/base.so python tools/layer_bench.py $L 2>/dev/null | tail -7
echo SWAP; python tools/layer_bench.py $L 2>/dev/null | tail -7
echo BASE b1; U3D_LIB_PATH=/root/repo/ab_libs/base.so python tools/layer_bench.py $L --batch 1 2>/dev/null | tail -7
echo SWAP b1; python tools/layer_bench.py $L --batch 1 2>/dev/null | tail -7
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 60 --warmup 10"
for i in 1 2 3; do
U3D_LIB_PATH=/root/repo/ab_libs/base.so $B 2>/dev/null | python -c "import sys,json; print('base ', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
$B 2>/dev/null | python -c "import sys,json; print('swap ', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done
