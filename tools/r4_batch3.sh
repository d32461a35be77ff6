#!/bin/bash
O=gpurun_out/r4b3; mkdir -p $O
timeout 600 python tools/scratch/mano_ab.py > $O/mano_ab.txt 2>&1
timeout 600 python tools/scratch/det_debug.py > $O/det_debug.txt 2>&1
DSF_WRW_STREAM=0 timeout 600 python tools/scratch/det_debug.py > $O/det_debug_onestream.txt 2>&1
grep -v Warn $O/mano_ab.txt | tail -6; tail -30 $O/det_debug.txt; tail -12 $O/det_debug_onestream.txt
