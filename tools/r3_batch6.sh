#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3b6; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/tests_all.log 2>&1; echo "tests rc $?" >> $O/summary.txt
for c in 3 4 5; do timeout 600 python tools/torch_ops_by_config.py $c > $O/torch_ops_config$c.log 2>&1; done
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench.err
cat $O/summary.txt; tail -12 $O/tests_all.log; grep -n "two-stage ResNet-50\|unsplit conv" $O/tests_all.log; head -40 $O/torch_ops_config5.log; python -c "
import json
j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline_raster'])"
