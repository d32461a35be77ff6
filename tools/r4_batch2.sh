#!/bin/bash
# round 4, GPU call 2: 256-thread MANO kernels -- A/B against the round-3 kernels, the suites that touch them, step trace
O=gpurun_out/r4b2; mkdir -p $O
timeout 600 python tools/scratch/mano_ab.py > $O/mano_ab.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_properties.py tests/test_gpu_determinism.py -x -q > $O/tests.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/$O/ks -o k -- python3 $R/tools/step_only.py > /dev/null 2>&1
( python3 $R/tools/kstats.py $R/$O/ks/k_kernel_trace.csv 12; python3 $R/tools/busy.py $R/$O/ks/k_kernel_trace.csv | tail -3 ) > $R/$O/kernel_categories.txt
rm -rf $R/$O/ks
cd $R; tail -3 $O/tests.txt; cat $O/mano_ab.txt | tail -6; grep -i mano $O/kernel_categories.txt
