python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
python bench.py --gpus 1 --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-330
