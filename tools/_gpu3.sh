set -u
O=gpurun_out/r04c
mkdir -p $O
L=pytorch-3dunet_amd/pytorch3dunet_amd/lib
for v in base burst rb9 both spread all; do
  if [ $v = base ]; then unset U3D_LIB_PATH; else export U3D_LIB_PATH=$PWD/$L/libu3d_hip_$v.so; fi
  python tools/layer_bench.py --only fwd,dgrad --iters 10 2>&1 | grep -v amdgpu > $O/layer_$v.txt
  echo "== $v: $(grep 'total fwd' $O/layer_$v.txt) | $(grep 'total dgrad' $O/layer_$v.txt)"
done
unset U3D_LIB_PATH
python - <<'PY'
import re,glob
rows={}
for v in ("base","burst","rb9","both","spread","all"):
    for l in open(f"gpurun_out/r04c/layer_{v}.txt"):
        m=re.match(r"(\S+)\s+(\d+)->\s*(\d+) @\S+: fwd\s+([\d.]+) ms.*dgrad\s+([\d.]+) ms", l)
        if m: rows.setdefault(m.group(1),{})[v]=(float(m.group(4)),float(m.group(5)))
print("layer      " + "  ".join(f"{v:>13s}" for v in ("base","burst","rb9","both","spread","all")))
for k,d in rows.items():
    print(f"{k:10s} " + "  ".join(f"{d[v][0]:.3f}/{d[v][1]:.3f}" if v in d else "      -      " for v in ("base","burst","rb9","both","spread","all")))
PY
# correctness of the most aggressive variant
U3D_LIB_PATH=$PWD/$L/libu3d_hip_all.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
for v in base both all; do
  if [ $v = base ]; then unset U3D_LIB_PATH; else export U3D_LIB_PATH=$PWD/$L/libu3d_hip_$v.so; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v['ms_per_step'] for k,v in list(d['roofline']['families'].items())[:3]})"
done
