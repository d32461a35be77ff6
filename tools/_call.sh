set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c13; mkdir -p $O; cd $R
timeout 1800 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -n 6 $O/pytest_gpu.log
