#!/bin/bash
# BASELINE config 4 (ResidualUNet3D f_maps=64, 1x80x160x160, compute_dtype bf16) under rocprofv3: kernel stats, HBM traffic
# (separate --pmc passes) and SQ counters of the same command.  Run from the repo root on the GPU box:
#   bash tools/run_profiles_cfg4.sh <tag> [extra model_bench flags]
set -u
tag=${1:-rXX}
shift
extra="$*"
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python tools/model_bench.py --bf16 $extra"
# a box whose first step faults or hangs costs minutes per command below: stop here instead
timeout -k 5 90 $B --no-events --steps 1 --warmup 1 > $out/sanity.log 2>&1 || { echo "sanity step failed on this box"; tail -3 $out/sanity.log; exit 1; }
timeout -k 10 150 $B --steps 15 --warmup 5 > $out/${tag}_cfg4_model_bench.jsonl 2> $out/cfg4.err
# (15 + 5 steps: the first two steps after an idle phase run on clocks ramping up from idle — profiles/r06f_step_family_times.txt; the 3 + 2 steps
# of rounds 4-6 were mostly such steps)
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/stats -- $B --no-events --steps 15 --warmup 5 > $out/stats.log 2>&1
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B --no-events --steps 1 --warmup 1 > $out/fetch.log 2>&1
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $B --no-events --steps 1 --warmup 1 > $out/write.log 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d $out/sq -- $B --no-events --steps 1 --warmup 1 > $out/sq.log 2>&1
db=$(find $out/stats -name "*.db" | head -1)
python tools/prof_summary.py stats "$db" 20 > $out/${tag}_cfg4_kernel_stats.md
python tools/prof_summary.py traffic $out/fetch $out/write 2 > $out/${tag}_cfg4_pmc_traffic.json
python tools/prof_summary.py pmc $out/sq > $out/${tag}_cfg4_pmc_sq.md
rm -rf $out/stats $out/fetch $out/write $out/sq
head -c 600 $out/${tag}_cfg4_model_bench.jsonl; echo
