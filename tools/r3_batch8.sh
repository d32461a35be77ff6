#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3b8; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/tests_all.log 2>&1; echo "tests rc $?" >> $O/summary.txt
timeout 600 python bench.py --config 5 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 5 fused generator', j['value'], j['ms_per_step'])" >> $O/summary.txt
bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1
cat $O/summary.txt; tail -8 $O/tests_all.log; cat gpurun_out/prof_r03/r03_kernel_categories.txt
