#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3b3; mkdir -p $O
for v in 0 1 4; do echo "== DSF_X6_SCHED=$v" >> $O/wrw_ab.log; DSF_X6_SCHED=$v timeout 600 python tools/wrw_ab.py 32 2>&1 | grep -v amdgpu.ids >> $O/wrw_ab.log; done
for v in 0 1 2 3 4 6; do
  DSF_X6_SCHED=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null > $O/bench_sched$v.json
  python -c "
import json,sys
j=json.loads(open('$O/bench_sched$v.json').read().strip().splitlines()[-1]); print('SCHED=$v', j['value'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['frac'], ' | '.join('%s %.1f' % (k.replace('igemm_',''), c['TFLOP/s']) for k, c in j['conv_kernels'].items()))" >> $O/bench_ab.log 2>&1
done
for v in "DSF_BN_ACC=1" "DSF_BN_ACC=0"; do
  env $v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', j['value'], j['ms_per_step'])" >> $O/bench_ab.log 2>&1
done
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_determinism.py -x -q -m gpu > $O/tests_conv.log 2>&1; echo "conv rc $?" >> $O/summary.txt
DSF_X6_SCHED=4 timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu > $O/tests_conv4.log 2>&1; echo "conv sched4 rc $?" >> $O/summary.txt
DSF_X6_SCHED=3 timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu > $O/tests_conv3.log 2>&1; echo "conv sched3 rc $?" >> $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_steps.py -x -q -m gpu -k "render_forward or pretrain_and_config4" > $O/tests_b.log 2>&1; echo "tests_b rc $?" >> $O/summary.txt
for v in "" "DSF_DETERMINISTIC=1"; do
  echo "== truth [$v]" >> $O/truth.log
  env $v timeout 900 python tools/step_truth.py ResNet_stage_50 3 2 2>&1 | grep -v "Warning\|amdgpu.ids\|print(" | head -5 >> $O/truth.log
done
cat $O/summary.txt $O/wrw_ab.log $O/bench_ab.log $O/truth.log; tail -5 $O/tests_conv.log $O/tests_b.log
