#!/bin/bash
O=gpurun_out/r4b7; mkdir -p $O
timeout 300 python tools/perf_pfd.py 2>&1 | grep -v amdgpu > $O/perf_pfd.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gpu_properties.py tests/test_gpu_edge.py tests/test_gpu_determinism.py -q -x > $O/tests_geom.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_dp.py -q -x -s > $O/tests_dp.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_steps.py -q -x -s -k "teacher_forced or resnet50_bottleneck" > $O/tests_r50.txt 2>&1
timeout 600 python tools/platform/mano_beside_conv_x6.py 2>&1 | grep -v amdgpu > $O/mano_beside_conv_x6.txt
timeout 600 python tools/winograd_probe.py 2>&1 | grep -v amdgpu > $O/winograd_probe.txt
timeout 600 python bench.py > $O/bench_config2.json 2> $O/bench_config2.err
cat $O/perf_pfd.txt; tail -3 $O/tests_geom.txt; tail -5 $O/tests_dp.txt; grep -i "teacher-forced blocks\|R50 golden\|passed\|failed" $O/tests_r50.txt; cat $O/mano_beside_conv_x6.txt $O/winograd_probe.txt; head -c 600 $O/bench_config2.json
