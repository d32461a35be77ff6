#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3b5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_determinism.py -q -m gpu > $O/tests_dp.log 2>&1; echo "tests dp rc $?" >> $O/summary.txt
for n in 1 2 3 1 2 3; do
  DSF_WRW_STREAMS=$n timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WRW_STREAMS=$n', j['value'], j['ms_per_step'])" >> $O/streams_ab.log 2>&1
done
for n in 1 2; do
  DSF_WRW_STREAMS=$n timeout 600 python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config5 WRW_STREAMS=$n', j['value'], j['ms_per_step'])" >> $O/streams_ab.log 2>&1
done
timeout 300 python tools/head_gemm.py > $O/head_gemm.log 2>&1
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_gpu_dp.py --deselect tests/test_gpu_determinism.py > $O/tests_rest.log 2>&1; echo "tests rest rc $?" >> $O/summary.txt
cat $O/summary.txt $O/streams_ab.log $O/head_gemm.log; tail -8 $O/tests_dp.log; tail -8 $O/tests_rest.log
