cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
for v in 0 1; do
U3D_PACK_BOTH=$v timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r06ab/pk$v -- python tools/model_bench.py --bf16 --act-bf16 --no-events --steps 5 --warmup 2 > /dev/null 2>&1
db=$(find gpurun_out/r06ab/pk$v -name "*.db" | head -1)
echo "== U3D_PACK_BOTH=$v"
python tools/prof_summary.py stats "$db" 7 | grep -E "pack_weights|total kernel" | cut -c1-200
rm -rf gpurun_out/r06ab/pk$v
done
