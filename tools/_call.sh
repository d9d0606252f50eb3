U3D_TUNE=22:1 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv3d_fwd or dgrad or many_tiles or subpixel" 2>&1 | grep -v '^$' | tail -6
echo BASE; python tools/layer_bench.py --only fwd,dgrad --iters 10 2>&1 | grep -v amdgpu | grep -E "enc0.c2|enc1|enc2|dec0.c2|dec1.c2|dec2.c2|total"
echo OVL; U3D_TUNE=22:1 python tools/layer_bench.py --only fwd,dgrad --iters 10 2>&1 | grep -v amdgpu | grep -E "enc0.c2|enc1|enc2|dec0.c2|dec1.c2|dec2.c2|total"
