#!/bin/bash
# Round profile: default bench line, rocprofv3 kernel stats of the same command, PMC passes (separate runs).
# usage (GPU box): bash tools/profile_round.sh r01
TAG=${1:-r01}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 400 python3 $R/bench.py > $O/${TAG}_bench_default.json 2> $O/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 $R/bench.py --steps 10 --warmup 4 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/rocprof.err
cp $O/kt/k_kernel_stats.csv $O/${TAG}_kernel_stats.csv
rm -rf $O/kt
# per-step category table and GPU-busy fraction from a trace of bare training steps (no bench diagnostics in it)
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/ks -o k -- python3 $R/tools/step_only.py --config 2 --steps 12 --warmup 4 > $O/step_only.log 2>&1
ms=$(grep STEP_ONLY $O/step_only.log | sed 's/.*wall_ms \([0-9.]*\).*/\1/')
( echo "config 2, per step over the 12 timed steps of tools/step_only.py (kernels starting in the last $ms ms of the trace):"; python3 $R/tools/kstats.py $O/ks/k_kernel_trace.csv 12 $ms; echo; echo "GPU busy per step (tools/busy.py, last steps):"; python3 $R/tools/busy.py $O/ks/k_kernel_trace.csv | tail -3 ) > $O/${TAG}_kernel_categories.txt
rm -rf $O/ks
run() { name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o p -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/$name.log 2>&1
  python3 $R/tools/pmc_summary.py $O/${TAG}_pmc_$name.json $O/$name/p_counter_collection.csv && rm -rf $O/$name $O/$name.log; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU
run l2 TCC_HIT_sum TCC_MISS_sum
ls -la $O
