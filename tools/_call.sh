timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "head" 2>&1 | grep -v '^$' | tail -2
out=gpurun_out/r06p; mkdir -p $out; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-roofline --no-extras"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- $B --steps 3 --warmup 3 > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1); mkdir -p $out/t; cp "$f" $out/t/x_kernel_trace.csv; rm -rf $out/trace
python tools/gap_analysis.py $out/t --list > $out/step_launches.txt; grep -n 'small\|head_' $out/step_launches.txt
