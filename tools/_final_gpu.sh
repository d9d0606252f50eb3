set -u
export TMPDIR=/tmp
bash tools/run_profiles.sh r04 2>&1 | tail -3
bash tools/run_profiles_cfg4.sh r04 2>&1 | tail -3
bash tools/run_profiles_cfg5.sh r04 2>&1 | tail -3
(time timeout 2400 python -m pytest tests -m gpu -q --durations=5) > gpurun_out/r04/r04_gputests.log 2>&1
grep -v amdgpu gpurun_out/r04/r04_gputests.log | tail -8
