R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c41; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_step_ops.py tests/test_gpu_parity.py tests/test_library_abi.py -q -m gpu -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -n 4 $O/pytest.log
python3 - <<'PY' > $O/decode_times.txt 2>&1
import torch, sys
sys.path.insert(0, ".")
from dsf_amd import ops
import bench
B, J, S = 32, 21, 64
maps = (torch.randn(B, 4 * J, S, S, device="cuda") * 0.3)
depth = torch.rand(B, 1, 128, 128, device="cuda") * 2 - 1
gj = torch.randn(B, J, 3, device="cuda")
for name, m in (("NCHW", maps.clone()), ("channels-last", maps.clone().contiguous(memory_format=torch.channels_last))):
    for cl in (True, False):
        ops.DECODE_CL[0] = cl
        us_f = bench.gpu_time_per_call_us(lambda: ops.Offset2Joint.apply(m, depth, 0.8, 30.0), 50)[0]
        mr = m.clone().requires_grad_(True)
        j = ops.Offset2Joint.apply(mr, depth, 0.8, 30.0)
        us_b = bench.gpu_time_per_call_us(lambda: torch.autograd.grad((j * gj).sum(), mr, retain_graph=True), 50, capture=False)[0]
        print("%-14s DECODE_CL=%d: forward %.1f us, backward (eager, incl. the sum) %.1f us" % (name, cl, us_f, us_b))
PY
grep -v amdgpu $O/decode_times.txt
timeout 900 python3 tools/ab_env.py --config 2 --var DSF_DECODE_CL --values 0 1 --rounds 8 > $O/ab_cl_c2.txt 2>&1; tail -n 2 $O/ab_cl_c2.txt
