#!/bin/bash
O=gpurun_out/r4b10; mkdir -p $O
timeout 300 python tools/perf_pfd.py 2>&1 | grep -v amdgpu > $O/perf_pfd.txt
timeout 3000 python -m pytest tests -q -m gpu > $O/tests_all.txt 2>&1
cat $O/perf_pfd.txt; tail -8 $O/tests_all.txt
