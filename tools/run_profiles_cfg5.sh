#!/bin/bash
# BASELINE config 5 (ResidualUNetSE3D f_maps=64 on a (3,96,192,192) volume, sliding-window inference: tools/predict_bench.py) in fp32
# and bf16: the bench lines (whole-volume time + roofline of the dominant MFMA family from HIP events) and the rocprofv3 kernel
# stats of the same commands.  Run from the repo root on the GPU box:   bash tools/run_profiles_cfg5.sh <tag>
set -u
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python tools/predict_bench.py > $out/${tag}_cfg5_predict_bench.jsonl 2> $out/cfg5.err
U3D_BF16=1 python tools/predict_bench.py >> $out/${tag}_cfg5_predict_bench.jsonl 2>> $out/cfg5.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/stats32 -- python tools/predict_bench.py --reps 2 > $out/stats32.log 2>&1
db=$(find $out/stats32 -name "*.db" | head -1)
python tools/prof_summary.py stats "$db" 3 > $out/${tag}_cfg5_fp32_kernel_stats.md
U3D_BF16=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/stats16 -- python tools/predict_bench.py --reps 2 > $out/stats16.log 2>&1
db=$(find $out/stats16 -name "*.db" | head -1)
python tools/prof_summary.py stats "$db" 3 > $out/${tag}_cfg5_bf16_kernel_stats.md
rm -rf $out/stats32 $out/stats16
cut -c1-400 $out/${tag}_cfg5_predict_bench.jsonl
