cd $GRAFT_REPO_ROOT; bash tools/run_profiles.sh r06; tail -n 2 gpurun_out/profile_round_r06.log
