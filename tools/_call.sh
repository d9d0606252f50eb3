true
mkdir -p gpurun_out/r06ab; cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
for v in 1 0; do
export U3D_SE_QUADS=$v
U3D_BF16=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r06ab/se$v -- python tools/predict_bench.py --reps 6 > gpurun_out/r06ab/se$v.log 2>&1
db=$(find gpurun_out/r06ab/se$v -name "*.db" | head -1)
echo "== U3D_SE_QUADS=$v"; tail -1 gpurun_out/r06ab/se$v.log | cut -c1-200
python tools/prof_summary.py stats "$db" 7 | grep -E "se_apply|nearest_add|maxpool2_fwd|smallc|total kernel" | cut -c1-60,100-180
rm -rf gpurun_out/r06ab/se$v
done
