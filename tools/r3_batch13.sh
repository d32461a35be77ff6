#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3b13; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/tests_all.log 2>&1; echo "tests rc $?" >> $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/summary.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench rc $?" >> $O/summary.txt
for a in "" "--no-graph"; do timeout 600 python bench.py --config 3 --no-cpu-baseline $a 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3 [$a]', j['value'], j['ms_per_step'], j['config']['hip_graph'])" >> $O/summary.txt; done
timeout 600 python tools/conv_layers.py 2>&1 | grep -v amdgpu > $O/conv_layers.txt; grep "wrw_x6_kernel<64>\|total" $O/conv_layers.txt
cat $O/summary.txt; tail -4 $O/tests_all.log; cat $O/smoke.log | tail -2; python -c "
import json
j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['frac'], j['roofline']['traffic'], j['roofline_raster']['frac'], j['cpu_baseline'])"
