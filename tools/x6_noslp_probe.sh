#!/bin/bash
# same box, alternating: the three convolution sources built with / without the SLP vectoriser (everything else is already without)
O=gpurun_out/x6_noslp; mkdir -p $O; : > $O/ab2.txt
cd dsf_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
for f in conv conv_x6 conv_c1; do
  /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o /tmp/${f}_slp.o &
  /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize -c $f.hip -o /tmp/${f}_noslp.o &
done
wait
cd ../..
for round in 1 2 3; do
  for v in slp noslp; do
    for f in conv conv_x6 conv_c1; do cp /tmp/${f}_$v.o dsf_amd/lib/$f.o; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dsf_amd/lib/libdsf_hip.so dsf_amd/lib/*.o
    timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); ck=d['conv_kernels']; print('convs $v: config 2', d['value'], d['ms_per_step'], 'c1 fwd/wrw', ck['conv_c1_fwd_kernel<5, 1>']['avg_launch_us'], ck['conv_c1_wrw_kernel<5, 1>']['avg_launch_us'], 'fp32 path', d['fp32_mfma_path']['ms_per_step'])" >> $O/ab2.txt
  done
done
cat $O/ab2.txt
