R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c87; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_steps.py tests/test_gpu_determinism.py -q -x -m gpu 2>&1 | tail -3
timeout 900 python3 tools/ab_env.py --config 2 --var DSF_EARLY_MODEL_TERMS --values 0 1 --block 10 --rounds 12 2>&1 | grep "^AB" | tee $O/ab_early.txt
