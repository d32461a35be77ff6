R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c47; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_conv.py -q -m gpu -x -k "c1 or stem or one_channel or conv2d_matches" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 3 $O/pytest.log
python3 - <<'PY' 2>&1 | grep -v amdgpu
import torch, sys
sys.path.insert(0, ".")
from dsf_amd import nn_conv
import bench
x = torch.randn(32, 1, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
gy = torch.randn(32, 64, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
us = bench.gpu_time_per_call_us(lambda: nn_conv._wrw_c1(x, gy, 5, 1, 2), 30)[0]
print("stem weight gradient (conv_c1_wrw + combine), B = 32: %.1f us" % us)
PY
