R=$GRAFT_REPO_ROOT; cd $R
timeout 1800 python3 -m pytest tests/test_gpu_bn_pool.py tests/test_gpu_bn_pair.py tests/test_gpu_fused.py tests/test_gpu_steps.py -q -m gpu 2>&1 | tail -2
