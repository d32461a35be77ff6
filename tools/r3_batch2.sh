#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3b2; mkdir -p $O
timeout 600 python tools/wrw_ab.py 32 > $O/wrw_ab.log 2>&1
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu > $O/tests_conv.log 2>&1; echo "conv rc $?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_transfer.py tests/test_gpu_fused.py -x -q -m gpu > $O/tests_a.log 2>&1; echo "tests_a rc $?" >> $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_steps.py -x -q -m gpu -k "render_forward or pretrain_and_config4" > $O/tests_b.log 2>&1; echo "tests_b rc $?" >> $O/summary.txt
for v in "DSF_X6_WRW_DIRECT=1 DSF_BN_ACC=1" "DSF_X6_WRW_DIRECT=0 DSF_BN_ACC=1" "DSF_X6_WRW_DIRECT=1 DSF_BN_ACC=0" "DSF_X6_WRW_DIRECT=0 DSF_BN_ACC=0"; do
  env $v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', j['value'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['frac'], j['roofline']['avg_launch_us'])" >> $O/bench_ab.log 2>&1
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/atomic_rounding.hip -o /tmp/atomic_rounding && timeout 120 /tmp/atomic_rounding > $O/atomic_rounding.log 2>&1
timeout 600 python tools/prio_ab.py 32 > $O/prio_ab.log 2>&1
for v in "TRUTH_DET=fwd" "TRUTH_DET=wrw" "DSF_WRW_STREAM=0"; do
  echo "== truth [$v]" >> $O/truth.log
  env $v timeout 900 python tools/step_truth.py ResNet_stage_50 3 2 2>&1 | grep -v "Warning\|amdgpu.ids\|print(" | head -4 >> $O/truth.log
done
echo "== truth frozen" >> $O/truth.log
timeout 900 python tools/step_truth.py ResNet_stage_50 3 2 frozen 2>&1 | grep -v "Warning\|amdgpu.ids\|print(" | head -4 >> $O/truth.log
cat $O/summary.txt $O/wrw_ab.log $O/bench_ab.log $O/atomic_rounding.log $O/prio_ab.log $O/truth.log; tail -5 $O/tests_conv.log $O/tests_a.log $O/tests_b.log
