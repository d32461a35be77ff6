#!/bin/bash
# PMC passes over the bench command (each counter group in its own run, --kernel-trace only), aggregated per kernel
# on the box so that only small JSON summaries travel back.  usage (on the GPU box): bash scratch/pmc_collect.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o p -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/$name.log 2>&1
  python3 $R/scratch/pmc_summary.py $O/$name.json $O/$name/p_counter_collection.csv && rm -rf $O/$name; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU
run l2 TCC_HIT_sum TCC_MISS_sum
ls -la $O
