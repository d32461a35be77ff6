R=$GRAFT_REPO_ROOT; cd $R
python3 tools/opt_block.py 2>&1 | grep -v amdgpu
python3 tools/host_lag.py 2>&1 | grep -v amdgpu
timeout 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
