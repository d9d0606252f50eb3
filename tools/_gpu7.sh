set -u
O=gpurun_out/r04g
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "small_cin or pack" 2>&1 | grep -v amdgpu | tail -3
timeout 600 python -m pytest tests/test_gpu_model.py -q -x 2>&1 | grep -v amdgpu | tail -3
for t in "13:0" "13:1"; do
  U3D_TUNE=$t timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['roofline']['families']; print('$t', d['value'], d['ms_per_step'], f['u3d_conv3d_small_cin_fwd'], f['u3d_pack_weights_batch'])"
done | tee $O/bench_smallc_ab.txt
