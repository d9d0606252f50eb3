B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10"
L=pytorch-3dunet_amd/pytorch3dunet_amd/lib
ms() { python -c "import sys,json; print('$1', json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])['ms_per_step'])"; }
for i in 1 2 3 4; do
$B 2>/dev/null | ms base
U3D_LIB_PATH=$L/libu3d_hip_idx32.so $B 2>/dev/null | ms idx32
done
cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
for v in base idx32; do
if [ $v = idx32 ]; then export U3D_LIB_PATH=$L/libu3d_hip_idx32.so; fi
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r06ab/ix$v -- python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 5 --warmup 3 > /dev/null 2>&1
db=$(find gpurun_out/r06ab/ix$v -name "*.db" | head -1)
echo "== $v"
python tools/prof_summary.py stats "$db" 8 | grep -E "gn_bwd_apply|maxpool2_bwd_merge|total kernel" | cut -c1-160
rm -rf gpurun_out/r06ab/ix$v
done
