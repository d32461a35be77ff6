R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c67; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_bn_pool.py tests/test_gpu_fused.py -q -m gpu > $O/pytest_pool.log 2>&1; echo "rc=$?" >> $O/pytest_pool.log; tail -n 15 $O/pytest_pool.log
timeout 900 python3 tools/ab_env.py --config 2 --var DSF_C1_STATS --values 0 1 --block 10 --rounds 12 2>&1 | tail -2 | tee $O/ab_c1stats_c2.txt
timeout 900 python3 tools/ab_env.py --config 3 --var DSF_C1_STATS --values 0 1 --block 10 --rounds 8 2>&1 | tail -2 | tee $O/ab_c1stats_c3.txt
