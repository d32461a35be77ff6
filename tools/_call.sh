R=$GRAFT_REPO_ROOT; cd $R
bash tools/run_profiles.sh r06
timeout 300 python3 tools/bn_pool_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_r06/r06_stem_probe.txt
timeout 300 python3 tools/host_lag.py 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_r06/host_lag_now.txt
cat gpurun_out/prof_r06/host_lag_now.txt
