U3D_TUNE=22:1 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv3d_fwd or dgrad or many_tiles" 2>&1 | grep -v '^$' | tail -4
echo BASE; python tools/layer_bench.py --only fwd,dgrad --iters 10 2>&1 | grep -v amdgpu
echo W3; U3D_TUNE=22:1 python tools/layer_bench.py --only fwd,dgrad --iters 10 2>&1 | grep -v amdgpu
