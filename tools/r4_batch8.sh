#!/bin/bash
O=gpurun_out/r4b8; mkdir -p $O
timeout 300 python tools/perf_pfd.py 2>&1 | grep -v amdgpu > $O/perf_pfd.txt
B=32 timeout 300 python tools/perf_pfd.py 2>&1 | grep -v amdgpu >> $O/perf_pfd.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gpu_properties.py tests/test_gpu_edge.py tests/test_gpu_determinism.py -q -x > $O/tests_geom.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_dp.py -q -x -s > $O/tests_dp.txt 2>&1
for c in 3 4 5; do timeout 900 python bench.py --config $c --no-cpu-baseline > $O/bench_config$c.json 2> $O/bench_config$c.err; done
cat $O/perf_pfd.txt; tail -3 $O/tests_geom.txt; tail -3 $O/tests_dp.txt; for c in 3 4 5; do head -c 330 $O/bench_config$c.json; echo; done
