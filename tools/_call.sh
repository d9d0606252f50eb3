timeout 1500 python -m pytest tests/test_optim.py tests/test_gpu_kernels.py -x -q -k "adam or groupnorm or maxpool" 2>&1 | grep -v '^$' | tail -8
out=gpurun_out/r06j; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-roofline --no-extras"
$B > $out/new1.log 2>&1; U3D_TUNE=18:1 $B > $out/k18.log 2>&1; $B > $out/new2.log 2>&1
for f in new1 k18 new2; do python - $out/$f.log $f <<'PY'
import sys,json
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        r=json.loads(ln); print(sys.argv[2], r['value'], r['ms_per_step'])
PY
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- $B --steps 3 --warmup 3 > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1); mkdir -p $out/t; cp "$f" $out/t/x_kernel_trace.csv; rm -rf $out/trace
python tools/gap_analysis.py $out/t --list > $out/step_launches.txt; grep -n 'maxpool2_fwd\|adam\|_stats' $out/step_launches.txt
