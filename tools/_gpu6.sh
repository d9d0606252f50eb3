set -u
O=gpurun_out/r04f
mkdir -p $O
export TMPDIR=/tmp
(time timeout 2400 python -m pytest tests -m gpu -q --durations=5) > $O/gputests.log 2>&1
grep -v amdgpu $O/gputests.log | tail -12
timeout 900 python tools/overlap_probe.py --slots 0 > $O/overlap_probe.txt 2>&1
timeout 300 python tools/overlap_probe.py --slots 0 --standin torch --only 4 >> $O/overlap_probe.txt 2>&1
grep -v amdgpu $O/overlap_probe.txt
timeout 300 python bench.py > $O/bench.json.log 2> $O/bench.err
tail -1 $O/bench.json.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['families']['u3d_pack_weights_batch'], d.get('extra_reference_step_order',{}).get('vs_value'))"
