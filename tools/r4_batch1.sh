#!/bin/bash
# round 4, GPU call 1: the two-stage ResNet-50 controls + a baseline bench line on this round's box
mkdir -p gpurun_out/r4b1
timeout 1500 python tools/r50_controls.py > gpurun_out/r4b1/controls_train.txt 2>&1
timeout 1500 python tools/r50_controls.py frozen > gpurun_out/r4b1/controls_frozen.txt 2>&1
DSF_CONV_MATH=f32 DSF_FUSED_BN=0 DSF_WRW_STREAM=0 DSF_DETERMINISTIC=1 timeout 900 python tools/step_truth.py ResNet_stage_50 3 2 > gpurun_out/r4b1/truth_all_switches.txt 2>&1
timeout 600 python bench.py > gpurun_out/r4b1/bench_default.json 2> gpurun_out/r4b1/bench_default.err
tail -3 gpurun_out/r4b1/controls_train.txt
