R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c42; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_step_ops.py -q -m gpu -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 3 $O/pytest.log
python3 - <<'PY' 2>&1 | grep -v amdgpu > $O/pool_linear.txt
import torch, sys
sys.path.insert(0, ".")
from dsf_amd import ops
import bench
for (B, C, H) in ((32, 512, 8), (64, 512, 8), (192, 2048, 4)):
    lin = torch.nn.Linear(C, 62).cuda()
    x = torch.randn(B, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    us = bench.gpu_time_per_call_us(lambda: ops.pool_linear(x, lin), 50)[0]
    print("pool_linear forward B=%d %dx%dx%d: %.1f us" % (B, H, H, C, us))
PY
cat $O/pool_linear.txt
for i in 1 2 3; do timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
