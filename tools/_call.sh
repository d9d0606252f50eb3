ms() { python -c "import sys,json; print('$1', json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])['ms_per_step'])"; }
B="python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5"
rocm-smi --showuniqueid 2>/dev/null | grep -E "Unique ID" | head -1
for i in 1 2 3 4 5; do
U3D_BENCH_BRACKET_PHASE=0 $B 2>/dev/null | ms "phase 0"
$B 2>/dev/null | ms "phase 5"
$B --no-roofline 2>/dev/null | ms "no events"
done
$B --steps 100 --warmup 10 2>/dev/null | ms "100 steps "
