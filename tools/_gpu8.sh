set -u
O=gpurun_out/r04h
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x 2>&1 | grep -v amdgpu | tail -6
for k in 0 2 1; do
  U3D_TUNE=4:$k python tools/layer_bench.py --only dgrad --iters 10 --layers enc0.c2 2>&1 | grep "enc0.c2" | sed "s/^/key4=$k /"
done | tee $O/layer_n16_ab.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_boundary.py -q -x 2>&1 | grep -v amdgpu | tail -4
for t in "4:0" "4:2"; do
  U3D_TUNE=$t timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['roofline']['families']; print('$t', d['value'], d['ms_per_step'], d['roofline']['frac'], f['u3d_conv3d'])"
done | tee $O/bench_n16_ab.txt
