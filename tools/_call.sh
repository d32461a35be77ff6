cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/c14
timeout 900 python3 tools/x6/try_wrw_splits.py > gpurun_out/c14/wrw_splits.txt 2>&1
cat gpurun_out/c14/wrw_splits.txt
