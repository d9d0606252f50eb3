timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad_job or groupnorm" 2>&1 | grep -v '^$' | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "golden" 2>&1 | grep -v '^$' | tail -3
out=gpurun_out/r06w; mkdir -p $out; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 60 --warmup 10"
for i in 1 2 3; do
U3D_WGRAD_JOB=0 $B 2>/dev/null | python -c "import sys,json; print('sep  ', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
U3D_WGRAD_JOB=1 $B 2>/dev/null | python -c "import sys,json; print('job  ', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done
