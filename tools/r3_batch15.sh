#!/bin/bash
# round 3, last verification of HEAD: whole GPU suite, smoke, default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3b15; mkdir -p $O
timeout 3000 python3 -m pytest tests -q -m gpu > $O/tests_all.log 2>&1; echo "tests rc $?"; tail -2 $O/tests_all.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench rc $?"
python3 -c "
import json
j=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline']['value'])"
