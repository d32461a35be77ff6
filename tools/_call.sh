R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c69; mkdir -p $O; cd $R
timeout 900 python3 tools/ab_env.py --config 2 --var DSF_C1_BN --values 0 1 --block 10 --rounds 12 2>&1 | tail -2 | tee $O/ab_c1bn_c2.txt
timeout 900 python3 tools/ab_env.py --config 5 --var DSF_C1_BN --values 0 1 --block 5 --rounds 8 2>&1 | tail -2 | tee $O/ab_c1bn_c5.txt
timeout 2400 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -n 6 $O/pytest_gpu.log
timeout 300 python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
