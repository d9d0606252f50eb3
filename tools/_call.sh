python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-300
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee gpurun_out/r06f_gputests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-260
