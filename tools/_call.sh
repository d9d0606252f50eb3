timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad" 2>&1 | grep -v '^$' | tail -2
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -k golden 2>&1 | grep -v '^$' | tail -2
B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10"
for i in 1 2 3; do
U3D_TUNE=23:1 $B 2>/dev/null | python -c "import sys,json; print('g4 ', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
$B 2>/dev/null | python -c "import sys,json; print('g1 ', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done
out=gpurun_out/r06z; mkdir -p $out; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-roofline --no-extras"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- $B --steps 3 --warmup 3 > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1); mkdir -p $out/t; cp "$f" $out/t/x_kernel_trace.csv; rm -rf $out/trace
python tools/gap_analysis.py $out/t --list > $out/step_launches.txt; grep ' wgrad_reduce' $out/step_launches.txt
