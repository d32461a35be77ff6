R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c86; mkdir -p $O; cd $R
timeout 2400 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -n 3 $O/pytest_gpu.log
timeout 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
