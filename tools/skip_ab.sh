#!/bin/bash
# same box, alternating: residual fan-in adds left to autograd (DSF_SKIP_EPILOGUE=0) vs added in the backward-data epilogue
O=gpurun_out/skip_ab; mkdir -p $O; : > $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_determinism.py tests/test_gpu_steps.py -q -x > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for round in 1 2; do
  for v in 0 1; do
    for c in 2 4 5; do
      DSF_SKIP_EPILOGUE=$v timeout 900 python bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('skip_epilogue=$v config $c', d['value'], d['ms_per_step'])" >> $O/ab.txt
    done
  done
done
cat $O/ab.txt
