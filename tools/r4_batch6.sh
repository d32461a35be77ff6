#!/bin/bash
O=gpurun_out/r4b6; mkdir -p $O
timeout 300 python tools/perf_pfd.py 2>&1 | grep -v amdgpu > $O/perf_pfd.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gpu_properties.py tests/test_gpu_edge.py tests/test_gpu_determinism.py -q -x > $O/tests.txt 2>&1
cat $O/perf_pfd.txt; tail -4 $O/tests.txt
