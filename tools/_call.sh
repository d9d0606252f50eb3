cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
for cfg in 2048 1024 768 512; do
export U3D_EXP_HB_BLOCKS=$cfg
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r06ab/c1 -- python tools/model_bench.py --bf16 --act-bf16 --no-events --steps 5 --warmup 2 > /dev/null 2>&1
db=$(find gpurun_out/r06ab/c1 -name "*.db" | head -1)
echo "== head bwd cap $cfg"
python tools/prof_summary.py stats "$db" 7 | grep -E "head_bwd|total kernel" | cut -c1-30,60-200
rm -rf gpurun_out/r06ab/c1
done
