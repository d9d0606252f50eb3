timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | grep -v '^$' | grep -v 'RCCL\|HIP version\|ROCm\|Hostname\|Librccl' | tail -4
python bench.py --steps 60 --warmup 10 2>/dev/null | tail -1
