cd $GRAFT_REPO_ROOT
bash tools/run_profiles.sh r06
ls gpurun_out/prof_r06 | head -50
