#!/bin/bash
# compiler scheduling options on conv_x6.hip (objects cross-compiled in the build container: tools/x6objs/), same box: per-kernel
# replay times (bench.py's conv_kernels) and the step
O=gpurun_out/x6_flags; mkdir -p $O; : > $O/ab.txt
for v in v0 v2 v3 v4 v5 v0 v2 v3; do
  cp tools/x6objs/$v.o dsf_amd/lib/conv_x6.o; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dsf_amd/lib/libdsf_hip.so dsf_amd/lib/*.o
  timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); ck=d['conv_kernels']; print('$v', d['value'], d['ms_per_step'], 'wrw128', d['roofline']['avg_launch_us'], 'x6b128', ck['igemm_x6b_kernel<128, false, 128>']['avg_launch_us'], 'x6_64', ck['igemm_x6_kernel<128, false, 64>']['avg_launch_us'], 'x6b64_256', ck['igemm_x6b_kernel<64, false, 256>']['avg_launch_us'], 'wrw64', ck['igemm_wrw_x6_kernel<64>']['avg_launch_us'])" >> $O/ab.txt
done
echo "v0 = shipped flags; v2 + -mllvm -amdgpu-sched-strategy=max-ilp; v3 + max-memory-clause; v4 + -amdgpu-schedule-metric-bias=0; v5 + -amdgpu-schedule-relaxed-occupancy" >> $O/ab.txt
cat $O/ab.txt
