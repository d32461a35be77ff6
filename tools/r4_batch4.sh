#!/bin/bash
O=gpurun_out/r4b4; mkdir -p $O; : > $O/summary3.txt
for m in 0 1 2; do echo "victim mode $m" >> $O/summary3.txt; timeout 300 python tools/platform/pk_fp32_beside_mfma.py tools/scratch/libpk_victim$m.so 2>&1 | grep -v amdgpu.ids >> $O/summary3.txt; done
cat $O/summary3.txt
