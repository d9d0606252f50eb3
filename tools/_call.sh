M="python tools/model_bench.py --name UNet3D --f-maps 32 --levels 4 --batch 1 --steps 15 --warmup 5 --no-events"
for p in 80,170,170 80,168,168 64,128,128; do $M --patch $p 2>/dev/null | tail -1 | cut -c1-200; done
for p in 112,234,234 112,232,232; do $M --patch $p --forward-only 2>/dev/null | tail -1 | cut -c1-200; done
