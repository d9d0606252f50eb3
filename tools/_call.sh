timeout 900 python -m pytest tests/test_gpu_b16.py tests/test_gpu_res.py tests/test_gpu_configs.py -q 2>&1 | grep -E "passed|failed|rror" | tail -4
cd /tmp && export TMPDIR=/tmp; cd - >/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r06ab/c1 -- python tools/model_bench.py --bf16 --act-bf16 --no-events --steps 5 --warmup 2 > /dev/null 2>&1
db=$(find gpurun_out/r06ab/c1 -name "*.db" | head -1)
python tools/prof_summary.py stats "$db" 7 | grep -E "head_bwd|smallc|total kernel" | cut -c1-30,60-200
rm -rf gpurun_out/r06ab/c1
