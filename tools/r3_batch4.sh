#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3b4; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/tests_all.log 2>&1; echo "tests rc $?" >> $O/summary.txt
for c in 2 3 4 5; do
  timeout 900 python bench.py --config $c > $O/bench_config$c.json 2> $O/bench_config$c.err; echo "bench config $c rc $?" >> $O/summary.txt
done
timeout 600 python bench.py --config 3 --graph --no-cpu-baseline > $O/bench_config3_graph.json 2>> $O/bench_config3.err
cat $O/summary.txt; tail -15 $O/tests_all.log
for c in 2 3 4 5; do python -c "
import json
j=json.loads(open('$O/bench_config$c.json').read().strip().splitlines()[-1])
print($c, j['value'], j['unit'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['frac'], j.get('cpu_baseline'))"; done
