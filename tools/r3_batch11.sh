#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3b11; mkdir -p $O
for v in 1 0 1 0; do
  DSF_GEN_FUSED=$v timeout 600 python bench.py --config 5 --steps 15 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config5 GEN_FUSED=$v', j['value'], j['ms_per_step'])" >> $O/gen_ab.log 2>&1
done
timeout 600 python tools/torch_ops_by_config.py 5 > $O/torch_ops_config5.log 2>&1
timeout 900 python tools/conv_layers.py > $O/r03_conv_layers.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_dp.py -q -m gpu > $O/tests_dp.log 2>&1; echo "dp rc $?" >> $O/summary.txt
cat $O/summary.txt $O/gen_ab.log; grep -n "copy_\|direct_copy" $O/torch_ops_config5.log | cut -c1-220 | head; tail -5 $O/tests_dp.log; head -30 $O/r03_conv_layers.txt
