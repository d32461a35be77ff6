#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3b10; mkdir -p $O
timeout 900 python tools/side_stream_soak.py 150 32 > $O/side_soak.log 2>&1
timeout 900 python tools/side_stream_soak.py 250 4 >> $O/side_soak.log 2>&1
timeout 1500 python tools/soak.py 1500 all > $O/soak.log 2>&1
timeout 900 python -m pytest tests/test_gpu_steps.py tests/test_gpu_transfer.py -q -m gpu > $O/tests_steps.log 2>&1; echo "steps+transfer rc $?" >> $O/summary.txt
bash tools/profile_configs.sh r03 > $O/profile_configs.log 2>&1
cat $O/summary.txt; grep -v amdgpu $O/side_soak.log; cat $O/soak.log | grep -v amdgpu | tail -20; tail -3 $O/tests_steps.log; head -3 gpurun_out/prof_r03/r03_config*_kernel_stats.txt
