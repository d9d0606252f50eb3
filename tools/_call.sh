timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | grep -v '^$' | grep -v 'RCCL\|HIP version\|ROCm\|Hostname\|Librccl' | tail -6
