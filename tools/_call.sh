for p in 80,170,170 80,168,168 64,128,128; do
python tools/model_bench.py --name UNet3D --f-maps 32 --levels 4 --patch $p --batch 1 --steps 10 --warmup 3 --no-events 2>/dev/null | tail -1 | cut -c1-300
U3D_STAT_REPS=1 python tools/model_bench.py --name UNet3D --f-maps 32 --levels 4 --patch $p --batch 1 --steps 10 --warmup 3 --no-events 2>/dev/null | tail -1 | cut -c1-300
done
python tools/model_bench.py --bf16 --act-bf16 --no-events --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-300
