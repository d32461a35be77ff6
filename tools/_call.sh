R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c51; mkdir -p $O; cd $R
timeout 2400 python3 -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 3 $O/pytest.log
