set -u
O=gpurun_out/r04e
mkdir -p $O
export TMPDIR=/tmp
(time timeout 2400 python -m pytest tests -m gpu -q --durations=6) > $O/gputests.log 2>&1
grep -v amdgpu $O/gputests.log | tail -25
timeout 900 python tools/overlap_probe.py --slots 0 > $O/overlap_probe_link.txt 2>&1
timeout 300 python tools/overlap_probe.py --slots 0 --standin torch --only 4 >> $O/overlap_probe_link.txt 2>&1
grep -v amdgpu $O/overlap_probe_link.txt
python tools/predict_bench.py > $O/cfg5_predict.jsonl 2>> $O/bench.err
U3D_BF16=1 python tools/predict_bench.py >> $O/cfg5_predict.jsonl 2>> $O/bench.err
cat $O/cfg5_predict.jsonl | cut -c1-1200
