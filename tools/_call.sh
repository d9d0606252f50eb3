for i in 1 2; do
for w in 3 5 10; do
python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps20 warmup$w ', d['ms_per_step'])"
done
done
