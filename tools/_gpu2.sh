set -u
O=gpurun_out/r04b
mkdir -p $O
export TMPDIR=/tmp
# 1. correctness of the new kernel variants first (kernel tests + model tests), then the rest of the suite
(time timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_boundary.py tests/test_gpu_graph.py -x -q) > $O/gputests_a.log 2>&1
tail -4 $O/gputests_a.log
# 2. same-box A/B of the two kernel changes, per layer
python tools/layer_bench.py --only fwd,dgrad --iters 10 > $O/layer_sw.txt 2>&1
U3D_TUNE=14:1 python tools/layer_bench.py --only fwd,dgrad --iters 10 > $O/layer_old.txt 2>&1
python tools/layer_bench.py --only wgrad --iters 10 > $O/layer_wg_flags.txt 2>&1
U3D_TUNE=15:1 python tools/layer_bench.py --only wgrad --iters 10 > $O/layer_wg_barrier.txt 2>&1
paste -d'|' $O/layer_sw.txt $O/layer_old.txt | grep -v amdgpu | cut -c1-230
paste -d'|' $O/layer_wg_flags.txt $O/layer_wg_barrier.txt | grep -v amdgpu | cut -c1-200
# 3. whole step, four combinations
for t in "0:0" "14:1" "15:1" "14:1,15:1"; do
  U3D_TUNE=$t timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['value'], d['ms_per_step'], d['roofline']['frac'], {k:v['ms_per_step'] for k,v in list(d['roofline']['families'].items())[:6]})"
done | tee $O/bench_ab.txt
# 4. CU-masked streams: launch cost, per-layer effect
tools/bin/cu_mask_probe > $O/cu_mask_probe.txt 2>&1
grep "per launch" $O/cu_mask_probe.txt
python tools/layer_bench.py --only fwd,wgrad --iters 10 --layers enc0.c2,dec1.c1,dec2.c2 --reserve 8 2>&1 | grep -v amdgpu | tee $O/layer_reserve8.txt
# 5. the rest of the GPU suite (incl. the slow config-4 full-size pin)
(time timeout 2000 python -m pytest tests -m gpu -x -q --durations=8 --deselect tests/test_gpu_kernels.py --deselect tests/test_gpu_model.py --deselect tests/test_gpu_boundary.py --deselect tests/test_gpu_graph.py) > $O/gputests_b.log 2>&1
tail -15 $O/gputests_b.log
