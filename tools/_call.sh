R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c99; mkdir -p $O; cd $R
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python3 bench.py 2>/dev/null | tail -1 > $O/bench.json; python3 -c "import json; d=json.loads(open('$O/bench.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'], sorted(d.keys()))"
