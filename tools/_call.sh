R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c75; mkdir -p $O; cd $R
timeout 1200 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_steps.py -q -x -m gpu 2>&1 | tail -2
timeout 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
