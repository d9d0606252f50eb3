timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_b16.py tests/test_gpu_res.py tests/test_gpu_bf16.py -q 2>&1 | grep -E "passed|failed|rror" | tail -6
for i in 1 2; do
U3D_CKPT_RERUN_LAST=1 python tools/model_bench.py --bf16 --act-bf16 --checkpoint --no-events --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c40-160
python tools/model_bench.py --bf16 --act-bf16 --checkpoint --no-events --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c40-160
done
python tools/model_bench.py --bf16 --act-bf16 --checkpoint --checkpoint-levels 2 --no-events --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c40-160
python tools/model_bench.py --bf16 --act-bf16 --no-events --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c40-160
