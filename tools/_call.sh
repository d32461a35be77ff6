R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c70; mkdir -p $O; cd $R
timeout 1800 python3 -m pytest tests/test_gpu_bn_pool.py tests/test_gpu_steps.py tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_determinism.py -q -x -m gpu > $O/pytest_sel.log 2>&1; echo "rc=$?" >> $O/pytest_sel.log; tail -n 12 $O/pytest_sel.log
timeout 900 python3 tools/ab_env.py --config 3 --var DSF_C1_BN --values 0 1 --block 10 --rounds 10 2>&1 | tail -2 | tee $O/ab_c1bn_c3.txt
timeout 900 python3 tools/ab_env.py --config 4 --var DSF_C1_BN --values 0 1 --block 3 --rounds 6 2>&1 | tail -2 | tee $O/ab_c1bn_c4.txt
