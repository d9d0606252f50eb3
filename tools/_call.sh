U3D_PROFILES_CORE=1 bash tools/run_profiles.sh r06b 2>&1 | tail -3
ls gpurun_out/r06b
