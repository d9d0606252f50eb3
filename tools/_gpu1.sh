set -u
mkdir -p gpurun_out/r04a
export TMPDIR=/tmp
tools/bin/cu_mask_probe > gpurun_out/r04a/cu_mask_probe.txt 2>&1
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15) > gpurun_out/r04a/gputests.log 2>&1
tail -5 gpurun_out/r04a/gputests.log
timeout 300 python bench.py > gpurun_out/r04a/bench.json.log 2> gpurun_out/r04a/bench.err
tail -1 gpurun_out/r04a/bench.json.log | cut -c1-600
timeout 600 python tools/overlap_probe.py --reserve 0 8 16 > gpurun_out/r04a/overlap_probe.txt 2>&1
cat gpurun_out/r04a/overlap_probe.txt | grep -v amdgpu.ids
cat gpurun_out/r04a/cu_mask_probe.txt
