R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c18; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_rccl.py tests/test_gpu_dp.py -q -m gpu > $O/pytest_dp.log 2>&1; echo "rc=$?" >> $O/pytest_dp.log
tail -n 5 $O/pytest_dp.log; cp gpurun_out/rccl_world1.log $O/
