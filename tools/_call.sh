R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c62; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_conv.py -q -m gpu -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 3 $O/pytest.log
timeout 900 python3 tools/ab_env.py --config 2 --var DSF_X6_LIVE --values 0 1 --rounds 6 > $O/ab_live_c2.txt 2>&1; tail -n 2 $O/ab_live_c2.txt
timeout 900 python3 tools/ab_env.py --config 4 --var DSF_X6_LIVE --values 0 1 --rounds 3 --block 4 > $O/ab_live_c4.txt 2>&1; tail -n 2 $O/ab_live_c4.txt
