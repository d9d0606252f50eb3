bash tools/run_profiles_cfg4.sh r06f_cfg4 --act-bf16
