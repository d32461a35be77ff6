R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c80; mkdir -p $O; cd $R
timeout 2400 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -n 3 $O/pytest_gpu.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python3 bench.py 2>/dev/null | tail -1 > $O/bench.json; python3 -c "import json; d=json.loads(open('$O/bench.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
