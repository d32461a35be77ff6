R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c61; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_fused.py -q -m gpu -x -k "hourglass_residual" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 12 $O/pytest.log
timeout 900 python3 tools/ab_env.py --config 3 --var DSF_CONV_RESIDUAL --values 0 1 --rounds 6 > $O/ab_res_c3.txt 2>&1; tail -n 2 $O/ab_res_c3.txt
