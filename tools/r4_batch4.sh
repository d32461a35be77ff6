#!/bin/bash
O=gpurun_out/r4b4; mkdir -p $O; : > $O/summary.txt
for load in wrw wrw_f32 fwd matmul elementwise; do
  LOAD=$load timeout 300 python tools/scratch/mano_race.py 2>&1 | grep "side-stream load" >> $O/summary.txt
done
for var in 0 1 2; do
  VAR=$var LOAD=wrw timeout 300 python tools/scratch/mano_race.py 2>&1 | grep "side-stream load" >> $O/summary.txt
done
cat $O/summary.txt
