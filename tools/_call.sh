set -u
out=gpurun_out/r06f; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "groupnorm or maxpool or small_cin" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "golden or cpu_oracle" 2>&1 | tail -8
B="python bench.py --no-cpu-baseline --no-roofline --no-extras"
for i in 1 2; do
$B > $out/new$i.log 2>&1
U3D_TUNE=18:1 $B > $out/twopass$i.log 2>&1
done
for f in new1 twopass1 new2 twopass2; do python - $out/$f.log $f <<'PY'
import sys,json
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        r=json.loads(ln); print(sys.argv[2], r['value'], r['ms_per_step'])
PY
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- $B --steps 3 --warmup 3 > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1); mkdir -p $out/t; cp "$f" $out/t/x_kernel_trace.csv; rm -rf $out/trace
python tools/gap_analysis.py $out/t --list > $out/step_launches.txt; head -3 $out/step_launches.txt
