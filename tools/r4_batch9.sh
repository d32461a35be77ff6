#!/bin/bash
O=gpurun_out/r4b9; mkdir -p $O
timeout 300 python tools/perf_pfd.py 2>&1 | grep -v amdgpu > $O/perf_pfd.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gpu_properties.py tests/test_gpu_edge.py tests/test_gpu_determinism.py -q -x > $O/tests_geom.txt 2>&1
bash tools/profile_configs.sh r04t "3" > $O/profile_configs.log 2>&1
cat $O/perf_pfd.txt; tail -3 $O/tests_geom.txt; cat gpurun_out/prof_r04t/r04t_config3_kernel_categories.txt | head -40; ls gpurun_out/prof_r04t
