#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3b7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_steps.py -q -m gpu -k "pretrain_and_config4" > $O/tests_r50.log 2>&1; echo "r50 rc $?" >> $O/summary.txt
for c in 5 4 2; do for pr in 0 -1 0 -1; do
  DSF_MAIN_PRIORITY=$pr timeout 600 python bench.py --config $c --steps 15 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c main priority $pr', j['value'], j['ms_per_step'])" >> $O/prio.log 2>&1
done; done
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench.err
cat $O/summary.txt $O/prio.log; grep -n "two-stage ResNet-50\|unsplit conv\|passed\|failed" $O/tests_r50.log | tail; python -c "
import json
j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline_raster'])"
