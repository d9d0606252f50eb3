timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "cell_packer" 2>&1 | grep -v '^$' | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "golden" 2>&1 | grep -v '^$' | tail -3
out=gpurun_out/r06v; mkdir -p $out; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-roofline --no-extras"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- $B --steps 3 --warmup 3 > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1); mkdir -p $out/t; cp "$f" $out/t/x_kernel_trace.csv; rm -rf $out/trace
python tools/gap_analysis.py $out/t --list > $out/step_launches.txt; grep -n 'pack_\|head_fwd' $out/step_launches.txt
