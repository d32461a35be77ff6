#!/bin/bash
O=gpurun_out/r4final; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/tests_all.txt 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python tools/perf_pfd.py 2>&1 | grep -v amdgpu > $O/perf_pfd.txt
tail -4 $O/tests_all.txt; head -c 400 $O/bench.json; echo; cat $O/perf_pfd.txt
