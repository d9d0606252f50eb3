cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
for cfg in "2048 2048" "1536 1536" "768 4096"; do
set -- $cfg
export U3D_EXP_C1_BLOCKS=$1 U3D_EXP_C1B_BLOCKS=$2
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r06ab/c1 -- python tools/model_bench.py --bf16 --act-bf16 --no-events --steps 5 --warmup 2 > /dev/null 2>&1
db=$(find gpurun_out/r06ab/c1 -name "*.db" | head -1)
echo "== fwd cap $1, bwd cap $2"
python tools/prof_summary.py stats "$db" 7 | grep -E "conv1x1_smallc|head_bwd|total kernel" | cut -c1-40,100-170
rm -rf gpurun_out/r06ab/c1
done
