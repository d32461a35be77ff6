#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/hg7; mkdir -p $O; cd $R
run() { env $1 timeout 300 python bench.py --config $2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $2 $1', d['value'], d['ms_per_step'])" >> $O/bench.txt; }
for rep in 1 2; do
  run "DSF_WRW_MIN_GFLOP=1e9" 3
  run "DSF_WRW_MIN_GFLOP=1e9 DSF_BRANCHES=0" 3
  run "DSF_WRW_MIN_GFLOP=1e9 DSF_LOSS_FORK=0" 3
  run "DSF_WRW_MIN_GFLOP=2" 3
  run "DSF_WRW_MIN_GFLOP=8" 3
  run "DSF_WRW_MIN_GFLOP=32" 3
  run "DSF_WRW_MIN_GFLOP=0" 3
done
for rep in 1 2; do
  run "DSF_WRW_MIN_GFLOP=0" 2
  run "DSF_WRW_MIN_GFLOP=2" 2
  run "DSF_WRW_MIN_GFLOP=8" 2
  run "DSF_WRW_MIN_GFLOP=32" 2
done
cat $O/bench.txt
