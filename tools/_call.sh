R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c46; mkdir -p $O; cd $R
timeout 1800 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_edge.py tests/test_gpu_stress.py tests/test_gpu_determinism.py tests/test_gpu_steps.py -q -m gpu > $O/pytest_geo.log 2>&1; echo "rc=$?" >> $O/pytest_geo.log; tail -n 4 $O/pytest_geo.log
timeout 900 python3 tools/ab_env.py --config 5 --var DSF_DECODE_CL --values 1 1 --rounds 3 --block 5 > $O/c5.txt 2>&1; tail -n 2 $O/c5.txt
