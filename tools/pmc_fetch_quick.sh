#!/bin/bash
# One FETCH_SIZE pass over the bench command -> corrected MB per launch of the split-operand convolution kernels (GPU box).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/f -o p -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > /tmp/b.log 2>&1 || { echo "pmc run failed"; tail -3 /tmp/b.log; exit 1; }
timeout 60 python3 $R/tools/pmc_summary.py /tmp/f.json /tmp/f/p_counter_collection.csv
timeout 60 python3 -c "
import json
d = json.load(open('/tmp/f.json'))
for k, v in sorted(d.items()):
    if 'igemm_x6' in k or 'wrw_x6' in k:
        print('%-45s n=%4d fetch MB (x2 corrected) %.1f' % (k[:45], v['FETCH_SIZE']['n'], v['FETCH_SIZE']['avg'] * 1024 * 2 / 1e6))
"
