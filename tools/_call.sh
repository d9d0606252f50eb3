B="python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 100 --warmup 10"
for i in 1 2 3; do
for t in "19:512" "19:1024" "19:2048"; do
U3D_TUNE=$t $B 2>/dev/null | python -c "import sys,json; print('$t ', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done
done
