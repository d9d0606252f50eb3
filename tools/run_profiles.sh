#!/bin/bash
# The measurement set committed under profiles/: bench line, rocprofv3 kernel stats, HBM traffic (separate --pmc passes) and
# SQ counters of the same command.  Run from the repo root on the GPU box:  bash tools/run_profiles.sh <tag>
set -u
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-roofline --no-extras"
python bench.py > $out/bench.json.log 2> $out/bench.err
# (the stats pass runs the driver's own K / W: the first two steps after an idle phase run on clocks ramping up from idle — with 3 + 3 steps,
# as until the end of round 6, most of the averaged launches were such steps: u3d_conv3d read 0.78-0.79 of peak where every steady-state step
# of the same trace reads 0.83, profiles/r06f_step_family_times.txt)
timeout 300 rocprofv3 --kernel-trace --stats -d $out/stats -- $B --steps 20 --warmup 5 > $out/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- $B --steps 20 --warmup 5 > $out/trace.log 2>&1
python tools/step_family_times.py $out/trace > $out/${tag}_step_family_times.txt
python tools/gap_analysis.py $out/trace --list > $out/${tag}_step_launches.txt
rm -rf $out/trace
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B --steps 2 --warmup 1 > $out/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $B --steps 2 --warmup 1 > $out/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d $out/sq -- $B --steps 2 --warmup 1 > $out/sq.log 2>&1
db=$(find $out/stats -name "*.db" | head -1)
python tools/prof_summary.py stats "$db" 25 > $out/${tag}_bench_kernel_stats.md
python tools/prof_summary.py traffic $out/fetch $out/write 3 > $out/${tag}_pmc_traffic.json
python tools/prof_summary.py pmc $out/sq > $out/${tag}_pmc_sq.md
python tools/prof_summary.py tables "$db" 25 $out/sq $out/bench.json.log > $out/${tag}_tables.md
cp $out/bench.json.log $out/${tag}_bench.json.log
rm -rf $out/stats $out/fetch $out/write $out/sq
tail -1 $out/bench.json.log | cut -c1-400
# (U3D_PROFILES_CORE=1: only the bench line and its rocprofv3 / PMC evidence — the GPU budget of a round is 90 minutes)
[ "${U3D_PROFILES_CORE:-0}" = "1" ] && exit 0
# opt-in paths, layer rates and accuracy probes quoted in DESIGN_HISTORY.md 4.7 / 4.9
python tools/bf16_bench.py > $out/${tag}_bf16_layer_bench.txt 2>&1
python tools/bf16_bench.py --fmaps 32 --patch 64,128,128 --batch 2 --levels 1 > $out/${tag}_cfg2_fullres_layer_bench.txt 2>&1
python tools/f32s_error_probe.py > $out/${tag}_f32s_error_probe.txt 2>&1
python tools/model_bench.py --bf16 > $out/${tag}_cfg4_model_bench.jsonl 2>> $out/bench.err
python tools/model_bench.py --bf16 --checkpoint >> $out/${tag}_cfg4_model_bench.jsonl 2>> $out/bench.err
python tools/model_bench.py --split >> $out/${tag}_cfg4_model_bench.jsonl 2>> $out/bench.err
