timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "bf16_wgrad_job or wgrad_job" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_b16.py tests/test_gpu_res.py tests/test_gpu_bf16.py -q 2>&1 | grep -E "passed|failed|Error|error" | tail -8
