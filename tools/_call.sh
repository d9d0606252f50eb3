timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "head" 2>&1 | grep -v '^$' | tail -6
out=gpurun_out/r06u; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-roofline --no-extras"
for i in 1 2; do $B > $out/new$i.log 2>&1; U3D_TUNE=21:1 $B > $out/vec$i.log 2>&1; done
for f in new1 vec1 new2 vec2; do python - $out/$f.log $f <<'PY'
import sys,json
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        r=json.loads(ln); print(sys.argv[2], r['value'], r['ms_per_step'])
PY
done
