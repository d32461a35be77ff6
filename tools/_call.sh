R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c81; mkdir -p $O; cd $R
timeout 1800 python3 -m pytest tests/test_gpu_bn_pool.py tests/test_gpu_steps.py tests/test_gpu_fused.py -q -m gpu 2>&1 | tail -2
