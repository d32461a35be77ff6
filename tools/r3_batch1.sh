#!/bin/bash
# round-3 GPU batch 1: new parity tests, the config-4 "1.8x" investigation, stream priorities, baseline bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3b1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_transfer.py -x -q -m gpu > $O/tests_a.log 2>&1; echo "tests_a rc $?" >> $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_steps.py -x -q -m gpu -k "render_forward or pretrain_and_config4" > $O/tests_b.log 2>&1; echo "tests_b rc $?" >> $O/summary.txt
timeout 600 python tools/prio_ab.py 32 > $O/prio_ab.log 2>&1
timeout 600 python bench.py --steps 30 --warmup 8 > $O/bench.json 2> $O/bench.err
for v in "" "DSF_CONV_MATH=f32" "DSF_DETERMINISTIC=1" "DSF_FUSED_BN=0"; do
  echo "== truth [$v]" >> $O/truth.log
  env $v timeout 900 python tools/step_truth.py ResNet_stage_50 3 2 2>&1 | grep -v Warning | head -12 >> $O/truth.log
done
echo "== truth frozen" >> $O/truth.log
timeout 900 python tools/step_truth.py ResNet_stage_50 3 2 frozen 2>&1 | grep -v Warning | head -12 >> $O/truth.log
tail -3 $O/tests_a.log $O/tests_b.log; cat $O/prio_ab.log; cat $O/truth.log
