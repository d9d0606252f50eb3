set -u
O=gpurun_out/r04d
mkdir -p $O
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_boundary.py tests/test_gpu_graph.py tests/test_gpu_res.py -x -q) > $O/gputests_a.log 2>&1
grep -v amdgpu $O/gputests_a.log | tail -4
timeout 900 python tools/overlap_probe.py --slots 0 16 32 > $O/overlap_probe_link.txt 2>&1
timeout 300 python tools/overlap_probe.py --slots 0 --standin torch --only 4 >> $O/overlap_probe_link.txt 2>&1
grep -v amdgpu $O/overlap_probe_link.txt
timeout 300 python bench.py > $O/bench.json.log 2> $O/bench.err
tail -1 $O/bench.json.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('extra_reference_step_order'), d['extra_fp32_split']['value'])"
python tools/predict_bench.py > $O/cfg5_predict.jsonl 2>> $O/bench.err
U3D_BF16=1 python tools/predict_bench.py >> $O/cfg5_predict.jsonl 2>> $O/bench.err
cat $O/cfg5_predict.jsonl | cut -c1-900
python tools/model_bench.py --bf16 > $O/cfg4_model_bench.jsonl 2>> $O/bench.err
python tools/model_bench.py --bf16 --checkpoint >> $O/cfg4_model_bench.jsonl 2>> $O/bench.err
cut -c1-700 $O/cfg4_model_bench.jsonl
