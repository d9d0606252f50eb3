mkdir -p gpurun_out/r06e
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/r06e/r06e_gputests.log
cat gpurun_out/r06e/r06e_gputests.log
U3D_PROFILES_CORE=1 bash tools/run_profiles.sh r06e
bash tools/run_profiles_cfg4.sh r06e_cfg4 --act-bf16
python tools/model_bench.py --bf16 --act-bf16 --no-events --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06e/cfg4_bare.jsonl
python tools/model_bench.py --bf16 --act-bf16 --checkpoint --no-events --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/r06e/cfg4_bare.jsonl
python tools/model_bench.py --bf16 --act-bf16 --checkpoint --checkpoint-levels 2 --no-events --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/r06e/cfg4_bare.jsonl
cat gpurun_out/r06e/cfg4_bare.jsonl
