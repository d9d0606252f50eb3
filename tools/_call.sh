U3D_POISON=1 timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_graph.py -k "not hip_graph and not graph" 2>&1 | grep -v '^$' | tail -12
