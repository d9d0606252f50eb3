python __graft_entry__.py --smoke 2>&1 | grep -E "smoke|Error|error" | cut -c1-400
U3D_PROFILES_CORE=1 bash tools/run_profiles.sh r06c 2>&1 | tail -2 | cut -c1-300
