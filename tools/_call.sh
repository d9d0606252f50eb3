U3D_PROFILES_CORE=1 bash tools/run_profiles.sh r06f
