timeout 900 python -m pytest tests/test_gpu_ragged.py tests/test_gpu_plus1.py -x -q 2>&1 | grep -v '^$' | tail -5
out=gpurun_out/r06m; mkdir -p $out; export TMPDIR=/tmp
M="python tools/model_bench.py --name UNet3D --f-maps 32 --levels 4 --batch 1 --steps 10 --warmup 3"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- $M --patch 80,170,170 --no-events --steps 3 > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1); mkdir -p $out/t; cp "$f" $out/t/x_kernel_trace.csv; rm -rf $out/trace
python tools/gap_analysis.py $out/t --list > $out/step_launches_170.txt; head -3 $out/step_launches_170.txt; grep -n 'chan_stats\|conv3d_mfma_kernel\|wgrad_kernel<true, false' $out/step_launches_170.txt
