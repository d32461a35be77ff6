#!/bin/bash
O=gpurun_out/r4b5; mkdir -p $O
timeout 600 python tools/scratch/pfd_debug.py > $O/pfd_debug.txt 2>&1
timeout 300 python tools/perf_pfd.py > $O/perf_pfd.txt 2>&1
timeout 3000 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_dp.py > $O/tests.txt 2>&1
cat $O/pfd_debug.txt; tail -5 $O/perf_pfd.txt; tail -15 $O/tests.txt
