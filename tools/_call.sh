timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "cell_packer" 2>&1 | grep -v '^$' | tail -8
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "golden" 2>&1 | grep -v '^$' | tail -4
out=gpurun_out/r06n; mkdir -p $out; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-roofline --no-extras"
for i in 1 2; do $B > $out/new$i.log 2>&1; U3D_PACK_ELEMENTWISE=1 $B > $out/oldpack$i.log 2>&1; done
for f in new1 oldpack1 new2 oldpack2; do python - $out/$f.log $f <<'PY'
import sys,json
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        r=json.loads(ln); print(sys.argv[2], r['value'], r['ms_per_step'])
PY
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- $B --steps 3 --warmup 3 > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_trace.csv" | head -1); mkdir -p $out/t; cp "$f" $out/t/x_kernel_trace.csv; rm -rf $out/trace
python tools/gap_analysis.py $out/t --list > $out/step_launches.txt; grep -n 'pack_' $out/step_launches.txt
