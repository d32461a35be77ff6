#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/hg10; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/full_suite.txt; cat $O/full_suite.txt
run() { env $1 timeout 400 python bench.py --config $2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $2 $1', d['value'], d['ms_per_step'])" >> $O/bench.txt; }
run "A=1" 3; run "A=1" 3; run "A=1" 3
cat $O/bench.txt
