cd $GRAFT_REPO_ROOT; O=gpurun_out/c60; mkdir -p $O
timeout 2400 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -n 3 $O/pytest_gpu.log
bash tools/run_profiles.sh r06; tail -n 1 gpurun_out/profile_round_r06.log
